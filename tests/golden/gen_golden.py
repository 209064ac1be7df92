#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own python on CPU.

Runs only in the build container (needs /root/reference).  The reference module
QQQ/gptq/qlinear/qlinear_marlin.py is loaded BY FILE PATH with a stub `QQQ._CUDA`
(its CUDA extension cannot be built here) and its two ctor guards neutralised
(qlinear_marlin.py:56-63 reject ROCm torch / query a CUDA device).  Only
`QuantLinear.pack()` and `QuantLinear.dynamic_quant()` are executed -- those are the
pieces of the hot path the reference can run without its kernel.  The outputs are
committed as DATA (inputs + expected outputs); no reference source is copied.

The int32 accumulators / fp16 outputs stored next to them come from the oracle
(oracle/qqq_ref.py) and act as regression pins for it and as parity targets for
the HIP kernel; they are labelled `oracle_*` to keep provenance explicit.

usage: python tests/golden/gen_golden.py            (writes tests/golden/qqq_golden.npz)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/QQQ/gptq/qlinear/qlinear_marlin.py"


def load_reference_module():
    stub_pkg = types.ModuleType("QQQ")
    stub_cuda = types.ModuleType("QQQ._CUDA")

    def _no_kernel(*a, **k):
        raise RuntimeError("reference CUDA kernel is not available in this container")

    stub_cuda.qqq_gemm = _no_kernel
    sys.modules["QQQ"] = stub_pkg
    sys.modules["QQQ._CUDA"] = stub_cuda
    spec = importlib.util.spec_from_file_location("ref_qlinear_marlin", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_case(rng, N, K, group_size):
    """Fake-quantised weight + scales the way gptq hands them to pack()
    (gptq/quant.py:85-93 scale conventions, gptq/gptq.py:198-217)."""
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    if group_size in (-1, K):  # group == K is handled as per-channel by pack() (qlinear_marlin.py:87-95)
        scale = (np.abs(W).max(axis=1, keepdims=True) / 7.0).astype(np.float32)  # [N,1]
        codes = np.clip(np.rint(W / scale), -7, 7)
        W_fq = (codes * scale).astype(np.float16)
        return W_fq, scale, None
    G = K // group_size
    Wg = W.reshape(N, G, group_size)
    scale = (2.0 * np.abs(Wg).max(axis=2) / 15.0).astype(np.float32)  # [N,G]
    u = np.clip(np.rint(Wg / scale[:, :, None]) + 8, 0, 15)
    W_fq = ((u - 8) * scale[:, :, None]).reshape(N, K).astype(np.float16)
    s_extra = (np.abs(W_fq.astype(np.float32)).max(axis=1, keepdims=True) / 127.0).astype(np.float16)
    return W_fq, scale, s_extra


def main():
    sys.path.insert(0, ROOT)
    from oracle import qqq_ref

    ref = load_reference_module()
    torch.version.hip = None  # qlinear_marlin.py:56-59
    torch.cuda.get_device_capability = lambda *a, **k: (8, 0)  # :60-63

    rng = np.random.Generator(np.random.PCG64(20240612))
    out = {}
    # (N, K) -> token counts M; kept small so the committed fixture stays ~1 MB
    shapes = {
        (64, 128): [1, 16],  # with group_size=128 this is the reference's "group == K" edge: per-channel
        (64, 256): [3],
        (128, 256): [1, 5, 16, 17, 33, 64],
        (256, 256): [16, 40],
        (256, 512): [1, 16],
        (128, 1024): [100, 128],  # 8 scale groups of 128, token counts past one MFMA tile (SURVEY 8c list)
    }
    names = []
    for group_size in (-1, 128):
        for (N, K), Ms in shapes.items():
            tag = f"g{group_size}_n{N}_k{K}"
            W_fq, scale, s_extra = make_case(rng, N, K, group_size)
            bias = (rng.standard_normal(N) * 0.1).astype(np.float16)
            lin = torch.nn.Linear(K, N, bias=True).half()
            lin.weight.data = torch.from_numpy(W_fq.copy())
            lin.bias.data = torch.from_numpy(bias.copy())
            ql = ref.QuantLinear(4, group_size, K, N, bias=True)
            ql.pack(
                lin,
                torch.from_numpy(scale.copy()),
                None if s_extra is None else torch.from_numpy(s_extra.copy()),
            )
            B = ql.B.numpy().copy()
            s_channel = ql.s_channel.numpy().copy()
            s_group = ql.s_group.numpy().copy()
            out[f"{tag}/W_fq"] = W_fq
            out[f"{tag}/Ms"] = np.array(Ms)
            out[f"{tag}/scale"] = scale
            if s_extra is not None:
                out[f"{tag}/s_extra"] = s_extra
            elif group_size != -1:
                s_extra = (np.abs(W_fq.astype(np.float32)).max(axis=1, keepdims=True) / 127.0).astype(np.float16)
                out[f"{tag}/s_extra"] = s_extra  # passed to pack() and ignored by it
            out[f"{tag}/bias"] = bias
            out[f"{tag}/ref_B"] = B
            out[f"{tag}/ref_s_channel"] = s_channel
            out[f"{tag}/ref_s_group"] = s_group
            for M in Ms:
                x = rng.standard_normal((M, K)).astype(np.float16)
                if M >= 5:
                    x[M // 2, :] *= np.float16(6.0)  # an outlier token
                xq, s1 = ql.dynamic_quant(torch.from_numpy(x.copy()))
                xq = xq.numpy().copy()
                s1 = s1.numpy().copy()
                D, acc = qqq_ref.qqq_gemm(xq, B, s1, s_channel, s_group, return_acc=True)
                mt = f"{tag}/m{M}"
                out[f"{mt}/x"] = x
                out[f"{mt}/ref_xq"] = xq
                out[f"{mt}/ref_s1"] = s1
                out[f"{mt}/oracle_acc"] = acc
                out[f"{mt}/oracle_D"] = D
            names.append(tag)
    out["cases"] = np.array(names)
    path = os.path.join(HERE, "qqq_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", len(out), "arrays")


if __name__ == "__main__":
    main()
