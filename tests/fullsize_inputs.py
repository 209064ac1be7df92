"""Inputs of the full-size pins (N=8192, K=21760), drawn from a generator this repository owns (numpy PCG64) so that
the build container (reference pack(), tests/golden/gen_fullsize_pins.py), the CPU tests and the GPU box all see the
same bytes.  Distribution per SURVEY 8d: W ~ N(0, 0.02^2) quantised the way the reference's GPTQ flow does it
(gptq/quant.py:85-93, gptq/gptq.py:198-217), x ~ N(0,1)."""
import numpy as np

N_FULL, K_FULL = 8192, 21760
SEEDS = {-1: 2025_0901, 128: 2025_0902}
C0_SEED = 2025_0903


def layer_inputs(group_size, N=N_FULL, K=K_FULL):
    """(W_fq fp16 [N,K], scale f32 [N,1] or [N,G], s_extra fp16 [N,1] or None) as gptq hands them to pack()"""
    rng = np.random.Generator(np.random.PCG64(SEEDS[group_size]))
    W = rng.standard_normal((N, K), dtype=np.float32)
    W *= np.float32(0.02)
    if group_size == -1:
        scale = (np.abs(W).max(axis=1, keepdims=True) / np.float32(7.0)).astype(np.float32)
        np.divide(W, scale, out=W)
        np.rint(W, out=W)
        np.clip(W, -7, 7, out=W)
        W *= scale
        return W.astype(np.float16), scale, None
    G = K // group_size
    Wg = W.reshape(N, G, group_size)
    scale = (np.float32(2.0) * np.abs(Wg).max(axis=2) / np.float32(15.0)).astype(np.float32)  # [N,G]
    np.divide(Wg, scale[:, :, None], out=Wg)
    np.rint(Wg, out=Wg)
    Wg += np.float32(8.0)
    np.clip(Wg, 0, 15, out=Wg)
    Wg -= np.float32(8.0)
    Wg *= scale[:, :, None]
    W_fq = W.astype(np.float16)
    s_extra = (np.abs(W_fq).max(axis=1, keepdims=True).astype(np.float32) / np.float32(127.0)).astype(np.float16)
    return W_fq, scale, s_extra


def c0_tokens(M=16, K=K_FULL):
    rng = np.random.Generator(np.random.PCG64(C0_SEED))
    return rng.standard_normal((M, K), dtype=np.float32).astype(np.float16)


SWEEP_SEED = 2025_0904
SWEEP_MS = (1, 16, 128, 1024, 4096)  # BASELINE configs[1] / configs[2]


def sweep_tokens(M=4096, K=K_FULL):
    """tokens of the pinned BASELINE sweep: the sweep's M values are row prefixes of this draw"""
    rng = np.random.Generator(np.random.PCG64(SWEEP_SEED))
    return rng.standard_normal((M, K), dtype=np.float32).astype(np.float16)
