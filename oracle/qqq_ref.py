"""CPU ORACLE for the QQQ W4A8 GEMM hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (qqq_amd/) never imports it and never falls
back to it: the product path fails loudly when the HIP library is missing.

This is a numpy restatement of the arithmetic of the reference
(/root/reference, HandH1998/QQQ @ 2025-08-24).  Each function cites the
reference file:line it follows.  Nothing here is copied from the reference;
the closed forms were re-derived from the kernel's index arithmetic and are
pinned (tests/test_oracle_golden.py) against vectors produced by importing the
reference's own `QuantLinear.pack()` / `dynamic_quant()` in the build container
(tests/golden/gen_golden.py).

Parity status ("what pins this oracle"):
  * packed layout (B, s_channel, s_group)  -- PINNED by reference pack() output.
  * dynamic_quant                          -- PINNED by reference dynamic_quant()
                                              output (CPU torch semantics; the
                                              CUDA/ROCm torch semantics differ in
                                              one documented place, see below).
  * int32 accumulators + fp16 epilogue     -- restated from csrc/qqq_gemm.cu (the
                                              CUDA kernel cannot be built or run anywhere
                                              in this project: no nvcc, PTX inline asm; the
                                              reference ships NO tests / golden vectors for
                                              it).  PINNED two ways since round 3:
                                              (1) tests/marlin_model.py follows Marlin<>
                                              thread by thread (its own index expressions,
                                              ldmatrix / mma.m16n8k16 fragment layouts,
                                              reduce and write-out) on reference-packed
                                              operands and reproduces this oracle's
                                              accumulators and outputs bit for bit
                                              (tests/test_marlin_model_cpu.py);
                                              (2) cross-checked against the fake-quant float
                                              path the reference defines (D ~= (xq*s1) @
                                              W_fq.T) with an absolute tolerance.  What is
                                              taken from the PTX ISA rather than from a run:
                                              the semantics of the instructions themselves.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------------
# layout: permutations used by QuantLinear.pack  (qlinear_marlin.py:147-176)
# ----------------------------------------------------------------------------------------------


def s_channel_stored_index(n: np.ndarray) -> np.ndarray:
    """Stored position of logical column n's per-channel scale.

    pack() applies `_scale_perm_single` inside every block of 32 columns
    (qlinear_marlin.py:173-175, :216-219, :222-226):
        stored[32*blk + 8*i + jj] = orig[32*blk + 2*i + (0,1,8,9,16,17,24,25)[jj]]
    Inverting: for n%32 = 2*i + 8*q + e  ->  stored = 32*blk + 8*i + 2*q + e.
    """
    n = np.asarray(n)
    w = n % 32
    i = (w % 8) // 2
    e = w % 2
    q = w // 8
    return (n // 32) * 32 + 8 * i + 2 * q + e


def s_group_stored_index(n: np.ndarray) -> np.ndarray:
    """Stored position of logical column n's per-group scale (within one group row).

    pack() applies `_scale_perm` inside every block of 64 columns
    (qlinear_marlin.py:170-172, :215):  stored[64*blk + 8*i + j] = orig[64*blk + i + 8*j].
    Inverting: stored = 64*blk + 8*(n%8) + (n%64)//8.
    """
    n = np.asarray(n)
    return (n // 64) * 64 + 8 * (n % 8) + (n % 64) // 8


def _nibble_maps(grouped: bool):
    """For nibble p (bits 4p..4p+3) of a packed word: (b, r) = (column half, k offset).

    Derived from how the kernel consumes a word (csrc/qqq_gemm.cu:146-151 + :540 per-channel;
    :167-210 + :536 per-group) and equal to pack()'s `interleave`
    (qlinear_marlin.py:164-168).
    """
    p = np.arange(8)
    if grouped:
        b = (p & 3) >> 1
        r = 2 * (p & 1) + (p >> 2)
    else:
        b = 1 - (p & 1)
        r = p >> 1
    return b, r


def word_coords(K: int, N: int, grouped: bool):
    """(k, n) coordinates of every nibble of the packed tensor B[K/16, 2N].

    Closed form of the Marlin/QQQ layout (qlinear_marlin.py:147-168, :228-248; consumed by
    csrc/qqq_gemm.cu:351-358, :379-383, :394-407, :523):
      word B[kt][wi], wi = 128*ng + 4*lane + j   (ng<N/64, lane<32, j<4) holds
      k = 16*kt + 4*(lane%4) + r,  n = 64*ng + 16*j + lane//4 + 8*b.
    Returns k, n arrays of shape [K/16, 2N, 8].
    """
    assert K % 16 == 0 and N % 64 == 0
    kt = np.arange(K // 16)[:, None, None]
    wi = np.arange(2 * N)[None, :, None]
    ng = wi // 128
    lane = (wi % 128) // 4
    j = wi % 4
    b, r = _nibble_maps(grouped)
    b = b[None, None, :]
    r = r[None, None, :]
    k = 16 * kt + 4 * (lane % 4) + r
    n = 64 * ng + 16 * j + lane // 4 + 8 * b
    k = np.broadcast_to(k, (K // 16, 2 * N, 8))
    n = np.broadcast_to(n, (K // 16, 2 * N, 8))
    return k, n


def pack_codes(codes: np.ndarray, grouped: bool) -> np.ndarray:
    """codes[K,N] -> B int32 [K/16, 2N].

    per-channel: codes are signed int4 in [-8,7] stored as two's-complement nibbles
    (qlinear_marlin.py:246-247); per-group: unsigned u in [0,15] (:242-244).
    """
    K, N = codes.shape
    k, n = word_coords(K, N, grouped)
    nib = (codes[k, n].astype(np.int64) & 0xF).astype(np.uint32)
    shifts = (4 * np.arange(8, dtype=np.uint32))[None, None, :]
    words = np.bitwise_or.reduce(nib << shifts, axis=2).astype(np.uint32)
    return words.view(np.int32)


def unpack_codes(B: np.ndarray, grouped: bool) -> np.ndarray:
    """B int32 [K/16, 2N] -> codes[K,N] (int8; signed int4 per-channel, unsigned u per-group)."""
    KT, W = B.shape
    K, N = KT * 16, W // 2
    k, n = word_coords(K, N, grouped)
    words = B.view(np.uint32)[:, :, None]
    shifts = (4 * np.arange(8, dtype=np.uint32))[None, None, :]
    nib = ((words >> shifts) & 0xF).astype(np.int8)
    if not grouped:
        nib = np.where(nib >= 8, nib - 16, nib).astype(np.int8)
    codes = np.empty((K, N), dtype=np.int8)
    codes[k, n] = nib
    return codes


# ----------------------------------------------------------------------------------------------
# weight operand as the kernel's tensor cores see it
# ----------------------------------------------------------------------------------------------


def dequant_per_group_faithful(u: np.ndarray, s3: np.ndarray) -> np.ndarray:
    """Bit-faithful restatement of `dequant_per_group` (csrc/qqq_gemm.cu:167-210).

    The kernel turns nibble u into fp16 (u-8) exactly (0x6400 magic, SUB/MUL/ADD constants,
    :169-187), then does ONE fp16 FMA (u-8)*s3 + 1152 (:194-201, MAGIC_NUM 0x6480), keeps the
    LOW BYTE of each fp16 result (prmt 0x6420, :206-207) and flips bit 7 (:208).
    (u-8)*s3 + 1152 is exact in float64, so float64 -> float16 (RNE) is the single rounding.
    For (u-8)*s3 in [-128, 127.5) this equals rint((u-8)*s3); outside, the byte wraps exactly
    like the kernel's does.
    """
    with np.errstate(over="ignore", invalid="ignore"):
        x = (u.astype(np.float64) - 8.0) * s3.astype(np.float64) + 1152.0
        h = x.astype(np.float16)
    low = (h.view(np.uint16) & 0xFF).astype(np.uint8) ^ np.uint8(0x80)
    return low.view(np.int8)


def weight_operand(B, s3, grouped: bool) -> np.ndarray:
    """int8 Wq[K,N] that multiplies the int8 activations inside the reference kernel.

    per-channel: Wq = 16*w4 (the kernel masks nibbles into the HIGH half of each byte,
    csrc/qqq_gemm.cu:146-151, :540; pack() pre-divides s_channel by 16,
    qlinear_marlin.py:221-226).
    per-group: Wq = dequant_per_group(u, s3[k//g, n]) with s3 in STORED (permuted) order.
    """
    codes = unpack_codes(np.ascontiguousarray(B), grouped)
    K, N = codes.shape
    if not grouped:
        return (codes.astype(np.int16) * 16).astype(np.int8)
    s3 = np.asarray(s3)
    G = s3.shape[0]
    gs = K // G
    sidx = s_group_stored_index(np.arange(N))
    s_log = s3[:, sidx]  # logical [G, N]
    s_full = np.repeat(s_log, gs, axis=0)  # [K, N]
    return dequant_per_group_faithful(codes, s_full)


def expand_int8(B, s3=None) -> np.ndarray:
    """What qqq_expand_int8 (include/qqq_amd.h; round 6, SURVEY 8 f-3's opt-in load-time re-layout) must produce for a
    layer: the int8 operand `weight_operand(B, s3, grouped)` -- per-group `dequant_per_group` (csrc/qqq_gemm.cu:167-210)
    applied once per weight, per-channel (s3 None / empty) 16 * w4 -- laid out in the wide kernel's MFMA operand order
        W8[k // 64][n // 64][2*hf + b][lane = 16*h + 4*c + jt][i] = Wq[64*(k // 64) + 16*h + i, 64*(n // 64) + 16*jt + 8*b + 4*hf + c]
    (h, c, jt in 0..3; hf, b in 0..1; i in 0..15).  The reference has no counterpart of the LAYOUT (it re-quantises inside its
    main loop, :527-537); the VALUES are its in-loop operand.  Returns int8 [K * N]."""
    grouped = s3 is not None and np.asarray(s3).size != 0  # (per-channel layers: the operand is 16 * w4, csrc/qqq_gemm.cu:146-151, :540)
    Wq = weight_operand(B, s3, grouped)  # [K, N]
    K, N = Wq.shape
    assert K % 64 == 0 and N % 64 == 0
    # axes of the reshape: k -> (s, h, i); n -> (ng, jt, b, hf, c)   [n % 64 = 16 jt + 8 b + 4 hf + c]
    W = Wq.reshape(K // 64, 4, 16, N // 64, 4, 2, 2, 4)  # [s, h, i, ng, jt, b, hf, c]
    W = W.transpose(0, 3, 6, 5, 1, 7, 4, 2)             # [s, ng, hf, b, h, c, jt, i]
    return np.ascontiguousarray(W).reshape(-1)


# ----------------------------------------------------------------------------------------------
# the GEMM
# ----------------------------------------------------------------------------------------------


def gemm_int32(A: np.ndarray, Wq: np.ndarray) -> np.ndarray:
    """acc[m,n] = sum_k A[m,k]*Wq[k,n] in int32 (mma ... s32.s8.s8.s32, csrc/qqq_gemm.cu:106-117).

    Evaluated through float64 BLAS: every partial sum is an integer < 2^53, so it is exact.
    `satfinite` never triggers for K*128*128 < 2^31 (K < 131072).
    """
    acc = A.astype(np.float64) @ Wq.astype(np.float64)
    assert np.abs(acc).max(initial=0) < 2**31
    return acc.astype(np.int64).astype(np.int32)


def epilogue(acc: np.ndarray, s1: np.ndarray, s2_stored: np.ndarray) -> np.ndarray:
    """D = fp16_rne( (fp32(acc) * s2[n]) * s1[m] )  (csrc/qqq_gemm.cu:695-700, :129-143).

    fp32(acc) is cvt.rn.f32.s32; the two multiplies are separate fp32 RN multiplies in this
    order; the final conversion is cvt.rn.f16.f32.  s2 arrives in stored (permuted) order.
    """
    M, N = acc.shape
    s2 = np.asarray(s2_stored, dtype=np.float32).reshape(-1)[s_channel_stored_index(np.arange(N))]
    s1 = np.asarray(s1, dtype=np.float32).reshape(M, 1)
    t = acc.astype(np.float32) * s2[None, :]
    t = t.astype(np.float32) * s1
    with np.errstate(over="ignore"):
        return t.astype(np.float32).astype(np.float16)


def qqq_gemm(A, B, s1, s2, s3=None, return_acc=False):
    """Full restatement of one `qqq_gemm` call (csrc/qqq_gemm.cu:1048-1106 -> :950-1046 -> :240-820)."""
    grouped = s3 is not None and np.asarray(s3).size != 0
    Wq = weight_operand(B, s3, grouped)
    acc = gemm_int32(np.asarray(A), Wq)
    D = epilogue(acc, s1, s2)
    return (D, acc) if return_acc else D


# ----------------------------------------------------------------------------------------------
# activation quantisation
# ----------------------------------------------------------------------------------------------


def dynamic_quant(x: np.ndarray, scalar_div: str = "div"):
    """Per-token int8 quantisation (qlinear_marlin.py:265-268).

        quant_scale = x.abs().max(-1, keepdim)[0].div(127.0).to(float32)
        xq = (x / quant_scale).round().clamp(-128, 127).to(int8)

    x is fp16.  `.div(127.0)` stays in fp16: torch evaluates it in fp32 and rounds to fp16.
    On CPU torch that is  fp16(fp32(amax) / 127.0f)           -> scalar_div="div";
    on CUDA/ROCm torch a division by a python scalar is lowered to a multiplication by the
    fp32 reciprocal,  fp16(fp32(amax) * (1.0f/127.0f))       -> scalar_div="recip"
    (this is what the reference computes on its real platform, and what the fused HIP
    kernel reproduces).  The two differ by one fp16 ulp for a small fraction of amax values.
    `x / quant_scale` promotes to fp32 (true division); round() is half-to-even.
    An all-zero row gives 0/0 = NaN -> int8 undefined in the reference; not exercised.
    """
    x = np.asarray(x, dtype=np.float16)
    amax = np.abs(x).max(axis=-1, keepdims=True).astype(np.float32)
    if scalar_div == "div":
        s = (amax / np.float32(127.0)).astype(np.float32)
    elif scalar_div == "recip":
        s = (amax * (np.float32(1.0) / np.float32(127.0))).astype(np.float32)
    else:
        raise ValueError(scalar_div)
    s = s.astype(np.float16).astype(np.float32)
    q = x.astype(np.float32) / s
    q = np.clip(np.rint(q.astype(np.float32)), -128, 127).astype(np.int8)
    return q, s


# ----------------------------------------------------------------------------------------------
# the offline side: fake-quant weights -> packed (pins the layout against reference pack())
# ----------------------------------------------------------------------------------------------


def _torch_div(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """tensor / tensor with torch's type promotion: evaluated in fp32, result dtype =
    promote(a, b) (half/half -> half, half/float -> float)."""
    rt = np.result_type(a.dtype, b.dtype)
    return (a.astype(np.float32) / b.astype(np.float32)).astype(rt)


def pack_from_fakequant(W_fq, scales, s_extra=None, group_size=-1):
    """Restatement of QuantLinear.pack (qlinear_marlin.py:181-262) on numpy arrays.

    W_fq: fp16 [N,K] fake-quantised weight; scales: [N, K/g] (per-group) or [N,1] (per-channel),
    fp32 in the real pipeline (gptq/quant.py:85-93 on W.float()), fp16 also accepted;
    s_extra: [N,1] second-level scales (per-group only; gptq/gptq.py:204-216).
    Returns (B int32 [K/16,2N], s_channel f32 [1,N], s_group f16 [K/g,N] or empty).
    """
    W = np.asarray(W_fq, dtype=np.float16)
    N, K = W.shape
    grouped = group_size not in (-1, K)
    sc = np.asarray(scales)
    if not grouped:
        s = sc.reshape(N, 1)
        # w = round(w / s), clamp(-7, 7)                                  (:202, :207)
        q = _torch_div(W, s)
        codes = np.clip(np.rint(q.astype(np.float32)), -7, 7).astype(np.int8).T  # [K,N]
        # s / 2**(8-bits) in s's dtype (exact: power of two), permuted, -> fp32  (:221-226)
        s_ch = (s.reshape(-1) / s.dtype.type(16.0)).astype(np.float32)
        s_ch_stored = np.empty(N, dtype=np.float32)
        s_ch_stored[s_channel_stored_index(np.arange(N))] = s_ch
        return pack_codes(codes, False), s_ch_stored.reshape(1, N), np.zeros((0,), np.float16)
    G = K // group_size
    s = sc.reshape(N, G)
    s_full = np.repeat(s, group_size, axis=1)  # [N,K]
    q = _torch_div(W, s_full)
    u = np.clip(np.rint(q.astype(np.float32)) + 8, 0, 15).astype(np.int8).T  # [K,N]  (:204-205)
    se = np.asarray(s_extra).reshape(1, N).astype(np.float32)  # (:209)
    # s_group = half( s / s_extra(fp32) )                                 (:210)
    s_g = _torch_div(np.ascontiguousarray(s.T), se).astype(np.float16)  # logical [G,N]
    s_g_stored = np.empty_like(s_g)
    s_g_stored[:, s_group_stored_index(np.arange(N))] = s_g
    s_ch_stored = np.empty(N, dtype=np.float32)
    s_ch_stored[s_channel_stored_index(np.arange(N))] = se.reshape(-1)
    return pack_codes(u, True), s_ch_stored.reshape(1, N), s_g_stored
