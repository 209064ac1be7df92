"""Round 6 (CPU): the opt-in load-time expansion of per-group weights (include/qqq_amd.h: qqq_expand_int8).

  * the oracle's W8 layout (oracle/qqq_ref.expand_int8) against an index-by-index construction from the documented closed form, and as
    the MFMA operand it claims to be: lane (h, c, jt) of column set q multiplying the activation fragment of its 16 k gives the GEMM;
  * the dispatcher plans the expanded weights exactly where the wide kernel runs a per-group call that has them, and argument checks
    happen on the host (no GPU)."""
import numpy as np
import pytest

from oracle import qqq_ref as R


def test_oracle_w8_layout_is_the_documented_closed_form():
    rng = np.random.default_rng(1)
    K, N = 256, 128
    codes = rng.integers(0, 16, size=(K, N)).astype(np.int8)
    s3 = (rng.random((K // 128, N), dtype=np.float32) * 30 + 0.5).astype(np.float16)
    B = R.pack_codes(codes, True)
    Wq = R.weight_operand(B, s3, True)
    W8 = R.expand_int8(B, s3).reshape(K // 64, N // 64, 4, 64, 16)
    for s in range(K // 64):
        for ng in range(N // 64):
            for q in range(4):
                hf, b = q >> 1, q & 1
                for lane in range(64):
                    h, c, jt = lane >> 4, (lane >> 2) & 3, lane & 3
                    n = 64 * ng + 16 * jt + 8 * b + 4 * hf + c
                    assert np.array_equal(W8[s, ng, q, lane], Wq[64 * s + 16 * h: 64 * s + 16 * h + 16, n]), (s, ng, q, lane)


def test_oracle_w8_as_mfma_operands_reproduces_the_gemm():
    """v_mfma_i32_16x16x64_i8 with the weights as the A operand: A-lane l holds row l % 16, k-block l // 16 (16 consecutive k); the
    B-lane (token j, k-block) holds the activations' same 16 k; D[row][j] sums over the four k-blocks.  Row r = 4 c + jt of column set
    q = 2 hf + b is column 16 jt + 8 b + 4 hf + c of the wave's 64 (the kernels' epilogue map)."""
    rng = np.random.default_rng(2)
    K, N, M = 128, 64, 16
    codes = rng.integers(0, 16, size=(K, N)).astype(np.int8)
    s3 = (rng.random((1, N), dtype=np.float32) * 15 + 0.5).astype(np.float16)
    B = R.pack_codes(codes, True)
    A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    acc = np.zeros((M, N), np.int64)
    W8 = R.expand_int8(B, s3).reshape(K // 64, N // 64, 4, 64, 16).astype(np.int64)
    for s in range(K // 64):
        for q in range(4):
            hf, b = q >> 1, q & 1
            for lane in range(64):
                h, row = lane >> 4, lane & 15
                c, jt = row >> 2, row & 3
                n = 16 * jt + 8 * b + 4 * hf + c
                x = A[:, 64 * s + 16 * h: 64 * s + 16 * h + 16].astype(np.int64)  # [M, 16]
                acc[:, n] += x @ W8[s, 0, q, lane]
    assert np.array_equal(acc.astype(np.int32), R.gemm_int32(A, R.weight_operand(B, s3, True)))


def test_plan_uses_expanded_weights_only_where_the_wide_kernel_runs():
    from qqq_amd import _lib, build

    build.build()
    for (N, K) in ((8192, 21760), (4096, 4096), (11008, 4096), (4096, 11008)):
        for M in (1, 16, 128, 512, 1024, 4096, 8192):
            base = _lib.plan(M, N, K, 128, 16)
            assert base["w8"] == 0
            pl = _lib.plan(M, N, K, 128, 16, tune=dict(w8=1))
            assert (pl["w8"] == 1) == (pl["kernel"] == 5), (N, K, M, pl)
            pc = _lib.plan(M, N, K, -1, 16, tune=dict(w8=1))
            assert (pc["w8"] == 1) == (pc["kernel"] == 5), (N, K, M, pc)
            assert _lib.plan(M, N, K, 128, 16, tune=dict(w8=-1))["w8"] == 0
    # the BASELINE per-group sweep points the expansion is for
    assert _lib.plan(4096, 8192, 21760, 128, 16, tune=dict(w8=1))["w8"] == 1
    assert _lib.plan(1024, 8192, 21760, 128, 16, tune=dict(w8=1))["w8"] == 1
    # K % 128 != 0: the wide kernel (whole 128-k stages) is not a candidate, nothing to use them in
    assert _lib.plan(4096, 4096, 4096 + 64, -1, 16, tune=dict(w8=1))["w8"] == 0


def test_expand_int8_argument_checks_need_no_gpu():
    from qqq_amd import _lib, build

    build.build()
    L = _lib.lib()
    buf = np.zeros(1 << 12, np.uint8)
    p = (buf.ctypes.data + 63) & ~63
    assert L.qqq_expand_int8(None, None, None, 0, 256, 128, 0, None) == 0          # empty: success, nothing launched
    assert L.qqq_expand_int8(None, p, p, 256, 256, 128, 0, None) == 17             # null B
    assert L.qqq_expand_int8(p, p, p, 256, 256, 64, 0, None) == 17                 # groups of 128 (or per-channel: -1) only
    assert L.qqq_expand_int8(p, None, p, 256, 256, 128, 0, None) == 17             # per-group needs the group scales
    assert L.qqq_expand_int8(p, p, p, 192, 256, 128, 0, None) == 17                # k % 128
    assert L.qqq_expand_int8(p, p, p, 256, 96, 128, 0, None) == 17                 # n % 64
    assert L.qqq_expand_int8(p + 4, p, p, 256, 256, 128, 0, None) == 17            # misaligned
    assert b"qqq_expand_int8" in L.qqq_amd_last_error()
