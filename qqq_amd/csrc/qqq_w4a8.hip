// qqq_w4a8.hip -- MI355X (gfx950 / CDNA4) W4A8 GEMM kernels behind QQQ's `qqq_gemm` operator.
//
// Not a port of the reference CUDA kernel (csrc/qqq_gemm.cu): the only things shared with it are
// the DATA contracts -- the Marlin/QQQ packed int4 layout produced by QuantLinear.pack()
// (qlinear_marlin.py:147-262) and the arithmetic
//     D[m,n] = fp16_rn( (f32_rn(sum_k A[m,k] * Wq[k,n]) * s2[n]) * s1[m] )
// (csrc/qqq_gemm.cu:106-117, :146-151, :167-210, :695-700).  Everything else is designed for
// wave64 / MFMA / LDS:
//
//  * Packed layout, closed form: word B[kt][128*ng + 4*(4*c + kq) + jt] holds, for k-tile kt
//    (16 k) and 64-column group ng, the 4 consecutive k = 16*kt + 4*kq + r (r = 0..3) of the two
//    columns n = 64*ng + 16*jt + c + 8*b (b = 0,1).  Hence for a fixed (kt, ng, c) the 64
//    contiguous bytes [64*c, 64*c+64) of the 512-byte block are ALL 16 k of the 8 columns
//    {16*jt + 8*b + c}.  One lane that loads those 64 bytes owns complete MFMA operands
//    (16 int8 along k) for 8 columns: the unpack `q & 0xF0F0F0F0`, `(q << 4) & 0xF0F0F0F0`
//    doubles as the register transpose, no cross-lane traffic, every bit used exactly once.
//  * MFMA operand roles are swapped w.r.t. the textbook: the WEIGHTS are the MFMA "A" operand
//    (tile row i <-> weight column n), the ACTIVATIONS the "B" operand (tile column j <-> token m),
//    so that each lane ends up with 4 CONSECUTIVE n of one token: 8-byte fp16 stores, 16-byte
//    int32 partial-sum stores.  The integer dot products do not care about the k order inside
//    an operand as long as both operands use the same order ("k-slot freedom"), and the
//    packed order [kq][r] IS natural k order, so the activation operand is 16 contiguous bytes.
//  * "column" kernel (decode, m <= 16): HBM-bound.  32 weight columns x all of K per workgroup, so a wide layer
//    fills the chip without split-K (one launch per call); packed words re-distributed between lanes with DPP.
//  * "stream" kernel (m <= 128): HBM-bound.  v_mfma_i32_16x16x64_i8; a lane (i = 8*g + c, h)
//    loads its 64 weight bytes of k-tile 4*s + h straight from HBM into VGPRs (no LDS: the
//    weights are used once), a wave eats 128 columns x 64 k = 4 KiB per step, waves of a
//    workgroup split K and reduce through LDS, workgroups split K through int32 slabs in the
//    reduce buffer C (int32 addition is associative: bit-exact for every split).
//  * "tiled" kernel (m > 128): MFMA-bound.  v_mfma_i32_32x32x32_i8; BMx256 tiles, BK = 128,
//    activations and RAW packed weights staged in LDS by LDS-DMA (XOR-swizzled 16-byte chunks so that
//    every fragment read is bank-conflict free), continuous fragment pipeline, XCD-aware tile order,
//    in-launch split-K through tile-sized slots of C.
//  Host side: make_plan() picks family / tile / split from a small measured cost model; the C-ABI entry points
//  are at the end of the file.
//
// The accumulators are the reference's: per-channel weights enter as 16*w4 (high nibble of each
// byte) and pack() has pre-divided s_channel by 16; per-group weights are re-quantised to int8
// with ONE packed-fp16 FMA exactly like dequant_per_group (csrc/qqq_gemm.cu:167-210).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/qqq_amd.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define QQQ_NIB_MASK 0xF0F0F0F0u

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------

// stored position of logical column n's per-channel scale (inverse of _scale_perm_single,
// qlinear_marlin.py:173-175):  n%32 = 2*i + 8*q + e  ->  32*(n/32) + 8*i + 2*q + e
__device__ __forceinline__ int s2_stored_index(int n) {
  const int w = n & 31;
  return (n & ~31) + 8 * ((w & 7) >> 1) + 2 * (w >> 3) + (w & 1);
}

// One lane's 4 consecutive outputs (n % 4 == 0): the fused dequant epilogue
// (csrc/qqq_gemm.cu:695-700): two separate fp32 RN multiplies, then RN to fp16.
__device__ __forceinline__ h4 epilogue_vals4(const int v0, const int v1, const int v2, const int v3,
                                             const int n, const float a_s,
                                             const float* __restrict__ s2) {
  const int i0 = s2_stored_index(n);  // n%4==0: (n, n+1) -> (i0, i0+1); (n+2, n+3) -> (i0+8, i0+9)
  const float2 sa = *reinterpret_cast<const float2*>(s2 + i0);
  const float2 sb = *reinterpret_cast<const float2*>(s2 + i0 + 8);
  h4 o;
  o[0] = (_Float16)__fmul_rn(__fmul_rn((float)v0, sa.x), a_s);
  o[1] = (_Float16)__fmul_rn(__fmul_rn((float)v1, sa.y), a_s);
  o[2] = (_Float16)__fmul_rn(__fmul_rn((float)v2, sb.x), a_s);
  o[3] = (_Float16)__fmul_rn(__fmul_rn((float)v3, sb.y), a_s);
  return o;
}

// ... stored straight from the lane; `bias` (may be null) is added in fp16 AFTER the fp16 round, exactly
// like the reference's separate `D + self.bias` (qlinear_marlin.py:287).
__device__ __forceinline__ void epilogue_store4(const int v0, const int v1, const int v2,
                                                const int v3, const int m, const int n,
                                                const int N, const float a_s,
                                                const float* __restrict__ s2,
                                                _Float16* __restrict__ D,
                                                int32_t* __restrict__ acc_out,
                                                const _Float16* __restrict__ bias = nullptr) {
  h4 o = epilogue_vals4(v0, v1, v2, v3, n, a_s, s2);
  if (bias) o = o + *reinterpret_cast<const h4*>(bias + n);
  *reinterpret_cast<h4*>(D + (size_t)m * N + n) = o;
  if (acc_out) {
    v4i a = {v0, v1, v2, v3};
    *reinterpret_cast<v4i*>(acc_out + (size_t)m * N + n) = a;
  }
}

// per-group int4 -> int8 re-quantisation of 4 weights (nibbles p0,p4,p1,p5 of q):
// u -> fp16(u-8) exactly, ONE fp16 FMA (u-8)*s + 1152, low byte, ^0x80
// (bit-identical to dequant_per_group, csrc/qqq_gemm.cu:167-210).
__device__ __forceinline__ unsigned dequant_group4(const unsigned q, const h2 s) {
  const unsigned t0 = (q & 0x000f000fu) | 0x64006400u;  // {1024+p0, 1024+p4}
  const unsigned t1 = (q & 0x00f000f0u) | 0x64006400u;  // {1024+16*p1, 1024+16*p5}
  const h2 c_sub = {(_Float16)-1032.0f, (_Float16)-1032.0f};
  const h2 c_mul = {(_Float16)0.0625f, (_Float16)0.0625f};
  const h2 c_add = {(_Float16)-72.0f, (_Float16)-72.0f};
  const h2 c_mag = {(_Float16)1152.0f, (_Float16)1152.0f};
  h2 a = __builtin_bit_cast(h2, t0) + c_sub;                                  // exact
  h2 b = __builtin_elementwise_fma(__builtin_bit_cast(h2, t1), c_mul, c_add);  // exact
  a = __builtin_elementwise_fma(a, s, c_mag);
  b = __builtin_elementwise_fma(b, s, c_mag);
  // bytes: [a.lo, a.hi, b.lo, b.hi] low bytes  (v_perm pool: src1 = bytes 0-3, src0 = bytes 4-7)
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a),
                               0x06040200u) ^
         0x80808080u;
}

// LDS-DMA of 16 bytes per lane: LDS destination = lds_dst (wave-uniform byte address) + 16*lane, the
// global source is per lane.  Issued through inline asm on purpose: hipcc cannot prove that the LDS
// image being filled (stage buf^1) does not alias the ds_reads of the stage being consumed, and would
// drain it with s_waitcnt vmcnt(0) before the first ds_read -- serialising load and compute.  Being asm,
// these loads are invisible to the compiler's wait-count bookkeeping: the kernel waits for them itself
// (one explicit vmcnt(0) in front of the stage barrier).  M0 is saved/restored around the DMA.
__device__ __forceinline__ void glds16(const void* gsrc, const unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

template <bool GROUPED>
__device__ __forceinline__ void unpack_pair(const unsigned q, const h2 s_b0, const h2 s_b1,
                                            int& w_b0, int& w_b1) {
  if constexpr (GROUPED) {
    w_b0 = (int)dequant_group4(q, s_b0);
    w_b1 = (int)dequant_group4(q >> 8, s_b1);
  } else {
    w_b0 = (int)(q & QQQ_NIB_MASK);         // odd nibbles  -> 16*w4 of column n      (b = 0)
    w_b1 = (int)((q << 4) & QQQ_NIB_MASK);  // even nibbles -> 16*w4 of column n + 8  (b = 1)
  }
}

// ------------------------------------------------------------------------------------------
// "stream" kernel: small m, weights HBM -> VGPR -> MFMA 16x16x64
// ------------------------------------------------------------------------------------------
//
// grid = (ceil(N/128) strips, ksplit, ceil(M / (16*MT)));  block = WAVES * 64.
// MFMA A operand lane l = (i = l & 15, h = l >> 4): i = 8*g + c  <->  weight columns
//     n = 128*strip + 64*g + 16*jt + 8*b + c   for the 8 MFMAs (jt, b) of a step,
//     k = 64*s + 16*h + [0,16)   (k-tile 4*s + h).
// MFMA B operand lane l = (j = l & 15, h): token m = mbase + 16*mt + j, same 16 k.
// MFMA D: lane l holds column j = l & 15 (token) and rows i = 4*(l >> 4) + r, r = 0..3, i.e.
//     g = l >> 5, c = 4*((l >> 4) & 1) + r  ->  4 consecutive n.

template <int MT>
struct StreamStep {
  v4u w[4];   // w[kq][jt]
  v4i x[MT];  // activation operands
  h8 sc;      // per-group scales [2*jt + b]
};

template <int MT, bool GROUPED, int WAVES, int PF>
__global__ __launch_bounds__(WAVES * 64) void qqq_stream_kernel(
    const int8_t* __restrict__ A, const unsigned char* __restrict__ B, int32_t* __restrict__ C,
    _Float16* __restrict__ D, const float* __restrict__ s1, const float* __restrict__ s2,
    const _Float16* __restrict__ s3, int32_t* __restrict__ acc_out, int* __restrict__ tickets,
    const _Float16* __restrict__ bias, const int M, const int N, const int K, const int ksplit,
    const int fused) {
  constexpr int NQ = MT * 8;  // MFMA output tiles per wave
  __shared__ int red[NQ * 4 * 64 + 64];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = blockIdx.x;
  const int sp = blockIdx.y;
  const int mbase = blockIdx.z * (16 * MT);

  if constexpr (WAVES > 1) {
    for (int i = tid; i < NQ * 4 * 64; i += WAVES * 64) red[i] = 0;
    __syncthreads();
  }

  const int j = lane & 15;
  const int h = lane >> 4;
  const int g = j >> 3;
  const int c = j & 7;
  const int ngroups = N >> 6;
  int ng = strip * 2 + g;
  if (ng >= ngroups) ng = ngroups - 1;  // clamp (N % 128 == 64): loads stay legal, output dropped
  const size_t rowbytes = (size_t)N * 8;
  const unsigned char* bptr = B + (size_t)h * rowbytes + (size_t)ng * 512 + c * 64;

  const int8_t* xptr[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = mbase + 16 * mt + j;
    if (row >= M) row = M - 1;
    xptr[mt] = A + (size_t)row * K + 16 * h;
  }
  const _Float16* sptr = GROUPED ? (s3 + (size_t)ng * 64 + c * 8) : nullptr;

  const int KS = K >> 6;  // 64-k steps
  const int ks_begin = (int)(((long long)KS * sp) / ksplit);
  const int ks_end = (int)(((long long)KS * (sp + 1)) / ksplit);

  v4i acc[MT][4][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      acc[mt][jt][0] = (v4i){0, 0, 0, 0};
      acc[mt][jt][1] = (v4i){0, 0, 0, 0};
    }

  auto load_step = [&](const int s, StreamStep<MT>& r) {
    const unsigned char* p = bptr + (size_t)(4 * s) * rowbytes;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
#ifdef QQQ_STREAM_NT
      r.w[kq] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p + 16 * kq));  // streamed once
#else
      r.w[kq] = *reinterpret_cast<const v4u*>(p + 16 * kq);
#endif
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) r.x[mt] = *reinterpret_cast<const v4i*>(xptr[mt] + 64 * s);
    if constexpr (GROUPED) r.sc = *reinterpret_cast<const h8*>(sptr + (size_t)(s >> 1) * N);
  };

  auto compute_step = [&](const StreamStep<MT>& r) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      v4i a0, a1;
      h2 sb0 = {(_Float16)0, (_Float16)0}, sb1 = sb0;
      if constexpr (GROUPED) {
        sb0 = (h2){r.sc[2 * jt], r.sc[2 * jt]};
        sb1 = (h2){r.sc[2 * jt + 1], r.sc[2 * jt + 1]};
      }
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        int w0, w1;
        unpack_pair<GROUPED>(r.w[kq][jt], sb0, sb1, w0, w1);
        a0[kq] = w0;
        a1[kq] = w1;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][jt][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, r.x[mt], acc[mt][jt][0], 0, 0, 0);
        acc[mt][jt][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, r.x[mt], acc[mt][jt][1], 0, 0, 0);
      }
    }
  };

  // software pipeline: PF steps (4 KiB of weights each) in flight per wave.  The steady-state loop is
  // branch-free so that hipcc can place COUNTED s_waitcnt vmcnt(N) (loads of the younger ring slots
  // stay in flight while the oldest slot is consumed); the ragged tail takes the checked path.
  StreamStep<MT> ring[PF];
  int s = ks_begin + wave;
  if (s + (2 * PF - 1) * WAVES < ks_end) {
    // unconditional prologue + branch-free loop: the wait counters are exact on every path
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      load_step(s + p * WAVES, ring[p]);
      __builtin_amdgcn_sched_barrier(0);  // ring order == issue order, so vmcnt(N) can be counted
    }
    for (; s + (2 * PF - 1) * WAVES < ks_end; s += PF * WAVES) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        compute_step(ring[p]);
        __builtin_amdgcn_sched_barrier(0);  // keep the refill right behind its consumer (hipcc would
        load_step(s + (p + PF) * WAVES, ring[p]);  // otherwise sink all loads to the loop end)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (s + p * WAVES < ks_end) load_step(s + p * WAVES, ring[p]);
  }
  for (; s < ks_end; s += PF * WAVES) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int sc = s + p * WAVES;
      if (sc < ks_end) {
        compute_step(ring[p]);
        const int sn = sc + PF * WAVES;
        if (sn < ks_end) load_step(sn, ring[p]);
      }
    }
  }

  // ---- reduce the waves of this workgroup through LDS (int adds: order-independent) ----
  if constexpr (WAVES > 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __hip_atomic_fetch_add(&red[(((mt * 4 + jt) * 2 + b) * 4 + r) * 64 + lane],
                                   acc[mt][jt][b][r], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            red[(((mt * 4 + jt) * 2 + b) * 4 + r) * 64 + lane] = acc[mt][jt][b][r];
    __syncthreads();
  }

  // ---- write out: item = (q = (mt, jt, b), lane) -> 4 consecutive n of one token ----
  auto item_coords = [&](const int it, int& m, int& n) {
    const int q = it >> 6, ln = it & 63;
    const int mt = q >> 3, jt = (q >> 1) & 3, b = q & 1;
    const int qd = ln >> 4;
    m = mbase + 16 * mt + (ln & 15);
    n = strip * 128 + 64 * (qd >> 1) + 16 * jt + 8 * b + 4 * (qd & 1);
  };

  if (ksplit == 1) {
    for (int it = tid; it < NQ * 64; it += WAVES * 64) {
      int m, n;
      item_coords(it, m, n);
      if (m < M && n < N) {
        const int q = it >> 6, ln = it & 63;
        const int* rp = &red[(q * 4) * 64 + ln];
        epilogue_store4(rp[0], rp[64], rp[128], rp[192], m, n, N, s1[m], s2, D, acc_out, bias);
      }
    }
    return;
  }

  // split-K: partial sums -> slab sp of C  (C[(sp*M + m)*N + n])
  for (int it = tid; it < NQ * 64; it += WAVES * 64) {
    int m, n;
    item_coords(it, m, n);
    if (m < M && n < N) {
      const int q = it >> 6, ln = it & 63;
      const int* rp = &red[(q * 4) * 64 + ln];
      v4i v = {rp[0], rp[64], rp[128], rp[192]};
      int32_t* dst = C + ((size_t)sp * M + m) * N + n;
      if (fused == 2) {
        // write-through (sc0 sc1) slab store: reaches memory without a later L2 write-back, so the
        // publish below needs no agent-scope release fence (MI355X hand-off recipe R1)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
      } else {
        *reinterpret_cast<v4i*>(dst) = v;
      }
    }
  }
  if (!fused) return;  // a separate reduce launch finishes the job

  // in-launch reduction by the last-arriving workgroup of this (strip, m-block) tile:
  // (fused == 1) plain stores -> agent-scope release -> ticket, or (fused == 2) write-through stores ->
  // drained -> ticket; then one agent-scope acquire in the last arriver (placement independent).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* flag = &red[NQ * 4 * 64];
  if (tid == 0) {
    if (fused == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int* tk = tickets + (blockIdx.z * gridDim.x + strip);
    const int t = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == ksplit - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // workspace zero on return
    }
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  for (int it = tid; it < NQ * 64; it += WAVES * 64) {
    int m, n;
    item_coords(it, m, n);
    if (m < M && n < N) {
      v4i sum = {0, 0, 0, 0};
      for (int p = 0; p < ksplit; ++p)
        sum += *reinterpret_cast<const v4i*>(C + ((size_t)p * M + m) * N + n);
      epilogue_store4(sum[0], sum[1], sum[2], sum[3], m, n, N, s1[m], s2, D, acc_out, bias);
    }
  }
}

// ------------------------------------------------------------------------------------------
// "column" kernel: decode (m <= 16).  One workgroup = 32 weight columns over the WHOLE K range, so the
// N/32 workgroups of a wide layer fill the chip without split-K: no slabs, no reduce launch (worth ~4 us of
// a ~21 us call at N=8192, K=21760).
// ------------------------------------------------------------------------------------------
//
// grid = (N/32, ksplit, ceil(M / (16*MT)));  block = WAVES * 64.  Workgroup x -> column group ng = x >> 1,
// half = x & 1: the 32 columns n = 64*ng + 16*jt + 8*b + 4*half + c', c' in [0,4) -- chunks c = 4*half + c'
// of the packed layout, 256 contiguous bytes of every 16-k row of B.  The waves split the 64-k steps of the
// K range round-robin and are summed through LDS at the end (as in the stream kernel).
// Weights and activations come in with plain 16-byte loads and the packed words are re-distributed between
// lanes in registers:
//     load lane L = 16*h + 4*c' + kq holds the 4 words (jt = 0..3) of piece (k-tile 4*s + h, chunk c', kq);
//     MFMA lane l = 16*h + 4*c' + jt needs the word jt of the pieces kq = 0..3 of the same (h, c'),
// i.e. a 4x4 transpose between the 4 registers and the 4 lanes of every quad: two butterfly stages of DPP
// quad_perm moves.  The MFMA row of lane l is then i = 4*c' + jt (any bijection onto the 16 rows will do);
// D lane l holds rows 4*(l >> 4) + r, i.e. c' = l >> 4 (of the OUTPUT lane) and jt = r.
// Each workgroup re-reads all m x K activation bytes (from L2), which at m = 16 equals its weight bytes:
// the kernel pays off for m <= 8 everywhere and up to m = 16 while m*K stays small (host heuristic).
// (An LDS-DMA ring variant -- global_load_lds into wave-private slots, no VGPR staging -- measured the same
// at m = 1 and slower at m = 16: the DMA path moves ~34 B/clk/CU against 64 B/clk for plain loads.)
template <int MT, bool GROUPED, int WAVES, int PF>
__global__ __launch_bounds__(WAVES * 64) void qqq_column_kernel(
    const int8_t* __restrict__ A, const unsigned char* __restrict__ B, int32_t* __restrict__ C,
    _Float16* __restrict__ D, const float* __restrict__ s1, const float* __restrict__ s2,
    const _Float16* __restrict__ s3, int32_t* __restrict__ acc_out, const _Float16* __restrict__ bias,
    const int M, const int N, const int K, const int ksplit) {
  constexpr int NQ = MT * 2;
  __shared__ int red[NQ * 4 * 64];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ng = blockIdx.x >> 1, half = blockIdx.x & 1;
  const int sp = blockIdx.y;
  const int mbase = blockIdx.z * (16 * MT);
  if constexpr (WAVES > 1) {
    for (int i = tid; i < NQ * 4 * 64; i += WAVES * 64) red[i] = 0;
    __syncthreads();
  }
  const size_t rowbytes = (size_t)N * 8;
  const int h = lane >> 4, cq = (lane >> 2) & 3, q4 = lane & 3;  // q4: kq as a load lane, jt as an MFMA lane
  const unsigned char* bptr = B + (size_t)h * rowbytes + (size_t)ng * 512 + (4 * half + cq) * 64 + q4 * 16;
  const int8_t* xptr[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = mbase + 16 * mt + (lane & 15);
    if (row >= M) row = M - 1;
    xptr[mt] = A + (size_t)row * K + 16 * h;
  }
  const _Float16* sptr = GROUPED ? (s3 + (size_t)ng * 64 + (4 * half + cq) * 8 + 2 * q4) : nullptr;

  const int KS = K >> 6;
  const int ks_begin = (int)(((long long)KS * sp) / ksplit);
  const int ks_end = (int)(((long long)KS * (sp + 1)) / ksplit);

  v4i acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt][0] = acc[mt][1] = (v4i){0, 0, 0, 0};

  struct Step {
    v4u w;
    v4i x[MT];
    h2 sc;
  };
  auto load_step = [&](const int s, Step& r) {
    r.w = *reinterpret_cast<const v4u*>(bptr + (size_t)(4 * s) * rowbytes);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) r.x[mt] = *reinterpret_cast<const v4i*>(xptr[mt] + 64 * s);
    if constexpr (GROUPED) r.sc = *reinterpret_cast<const h2*>(sptr + (size_t)(s >> 1) * N);
  };
  const bool odd = lane & 1, hi = lane & 2;
  auto compute_step = [&](const Step& r) {
    // 4x4 transpose over (register e, quad lane q): out[e](lane q) = in[q](lane e)
    unsigned z[4], y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // stage 1: exchange across lane bit 0 / register bit 0
      const unsigned t = (unsigned)__builtin_amdgcn_mov_dpp((int)r.w[e ^ 1], 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
      z[e] = (odd == (bool)(e & 1)) ? r.w[e] : t;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // stage 2: across lane bit 1 / register bit 1
      const unsigned t = (unsigned)__builtin_amdgcn_mov_dpp((int)z[e ^ 2], 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
      y[e] = (hi == (bool)(e & 2)) ? z[e] : t;
    }
    h2 sb0 = {(_Float16)0, (_Float16)0}, sb1 = sb0;
    if constexpr (GROUPED) {
      sb0 = (h2){r.sc[0], r.sc[0]};
      sb1 = (h2){r.sc[1], r.sc[1]};
    }
    v4i a0, a1;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      int w0, w1;
      unpack_pair<GROUPED>(y[kq], sb0, sb1, w0, w1);
      a0[kq] = w0;
      a1[kq] = w1;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[mt][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, r.x[mt], acc[mt][0], 0, 0, 0);
      acc[mt][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, r.x[mt], acc[mt][1], 0, 0, 0);
    }
  };

  // software pipeline as in the stream kernel: PF steps in flight per wave, branch-free steady state
  Step ring[PF];
  int s = ks_begin + wave;
  if (s + (2 * PF - 1) * WAVES < ks_end) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      load_step(s + p * WAVES, ring[p]);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (; s + (2 * PF - 1) * WAVES < ks_end; s += PF * WAVES) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        compute_step(ring[p]);
        __builtin_amdgcn_sched_barrier(0);
        load_step(s + (p + PF) * WAVES, ring[p]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (s + p * WAVES < ks_end) load_step(s + p * WAVES, ring[p]);
  }
  for (; s < ks_end; s += PF * WAVES) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int sc = s + p * WAVES;
      if (sc < ks_end) {
        compute_step(ring[p]);
        const int sn = sc + PF * WAVES;
        if (sn < ks_end) load_step(sn, ring[p]);
      }
    }
  }

  if constexpr (WAVES > 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __hip_atomic_fetch_add(&red[((mt * 2 + b) * 4 + r) * 64 + lane], acc[mt][b][r], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((mt * 2 + b) * 4 + r) * 64 + lane] = acc[mt][b][r];
  }
  __syncthreads();

  // write out: MFMA row i = 4*c' + jt, D lane ln holds rows 4*(ln >> 4) + r  ->  c' = ln >> 4, jt = r.
  // item = (q = (mt, b), jt, token): gathers c' = 0..3 (4 consecutive n) from the 4 lanes token + 16*c'.
  for (int it = tid; it < NQ * 64; it += WAVES * 64) {
    const int q = it >> 6, jt = (it >> 4) & 3, tok = it & 15;
    const int mt = q >> 1, b = q & 1;
    const int m = mbase + 16 * mt + tok;
    const int n = 64 * ng + 16 * jt + 8 * b + 4 * half;
    if (m < M) {
      const int* rp = &red[(q * 4 + jt) * 64 + tok];
      if (ksplit == 1) {
        epilogue_store4(rp[0], rp[16], rp[32], rp[48], m, n, N, s1[m], s2, D, acc_out, bias);
      } else {
        v4i v = {rp[0], rp[16], rp[32], rp[48]};
        *reinterpret_cast<v4i*>(C + ((size_t)sp * M + m) * N + n) = v;
      }
    }
  }
}

// Separate reduce + dequant launch for split-K partial sums: one thread = 4 consecutive n.
// Latency-bound (a few MB): all slab loads of a thread are issued before the first add.
__global__ __launch_bounds__(64) void qqq_reduce_kernel(const int32_t* __restrict__ C,
                                                        _Float16* __restrict__ D,
                                                        const float* __restrict__ s1,
                                                        const float* __restrict__ s2,
                                                        int32_t* __restrict__ acc_out,
                                                        const _Float16* __restrict__ bias, const int M,
                                                        const int N, const int ksplit) {
  const int nq = N >> 2;
  const long long total = (long long)M * nq;
  const size_t slab = (size_t)M * N;
  for (long long it = (long long)blockIdx.x * 64 + threadIdx.x; it < total;
       it += (long long)gridDim.x * 64) {
    const int m = (int)(it / nq);
    const int n = (int)(it % nq) * 4;
    const int32_t* p0 = C + (size_t)m * N + n;
    const float a_s = s1[m];
    v4i sum = {0, 0, 0, 0};
    int p = 0;
    for (; p + 4 <= ksplit; p += 4) {
      const v4i v0 = *reinterpret_cast<const v4i*>(p0 + (size_t)(p + 0) * slab);
      const v4i v1 = *reinterpret_cast<const v4i*>(p0 + (size_t)(p + 1) * slab);
      const v4i v2 = *reinterpret_cast<const v4i*>(p0 + (size_t)(p + 2) * slab);
      const v4i v3 = *reinterpret_cast<const v4i*>(p0 + (size_t)(p + 3) * slab);
      sum += (v0 + v1) + (v2 + v3);
    }
    for (; p < ksplit; ++p) sum += *reinterpret_cast<const v4i*>(p0 + (size_t)p * slab);
    epilogue_store4(sum[0], sum[1], sum[2], sum[3], m, n, N, a_s, s2, D, acc_out, bias);
  }
}

// ------------------------------------------------------------------------------------------
// "tiled" kernel: large m, LDS-staged, MFMA 32x32x32
// ------------------------------------------------------------------------------------------
//
// Workgroup tile = BM tokens x 256 weight columns (4 column groups ng0..ng0+3), BK = 128.
// MFMA A operand lane l = (i = l & 31, h = l >> 5): i = 8*g + c <-> columns
//     n = n0 + 64*g + 16*jt + 8*b + c,  k = 32*t + 16*h + [0,16)  (k-tile 2*t + h of the stage).
// MFMA B operand lane l = (j = l & 31, h): token m0 + 32*mtile + j, same k.
// MFMA D lane l: column j = l & 31 (token), rows i = (r & 3) + 8*(r >> 2) + 4*(l >> 5)
//     -> g = r >> 2, c = 4*h + (r & 3): 4 consecutive n per (jt, b, g).
// Waves: WM x WN; wave (wm, wn) owns tokens [32*MTW*wm, +32*MTW) and jt in [JW*wn, +JW).
//
// LDS stage: W region 8 k-tiles x 2048 B (raw packed words; 16-byte chunk (c, kq) of block
// (kt, g) is stored at chunk position 4*c + (kq ^ g): the four 16-lane groups of a
// ds_read_b128 then hit 16 distinct bank quads), X region BM rows x 128 B (chunk position
// p ^ ((row >> 1) & 7)).  With global_load_lds the LDS image is lane-linear, so the swizzle is
// applied to the per-lane SOURCE address; the register-staged variant writes the same image.

template <int BM, int MTW, int JW, int NB, bool GROUPED, int NS>
__global__ __launch_bounds__((BM / (32 * MTW)) * (4 / JW) * (2 / NB) * 64) void qqq_tiled_kernel(
    const int8_t* __restrict__ A, const unsigned char* __restrict__ B, int32_t* __restrict__ C,
    _Float16* __restrict__ D, const float* __restrict__ s1, const float* __restrict__ s2,
    const _Float16* __restrict__ s3, int32_t* __restrict__ acc_out,
    const _Float16* __restrict__ bias, const int M, const int N, const int K, const int ksplit,
    const int tiles_m, const int tiles_n, int* __restrict__ tickets, const int nslots, const int PW) {
  // wave tile: MTW m-tiles of 32 tokens x JW column tiles (jt) x NB column halves (b).  NB == 1: the two
  // b halves of a packed word go to two different waves (per-group mode: every weight is re-quantised
  // by exactly one wave of the workgroup).
  static_assert(NB == 1 || NB == 2, "NB");
  static_assert(JW == 1 || JW == 2 || JW == 4, "JW");
  constexpr int WM = BM / (32 * MTW);
  constexpr int WN = (4 / JW) * (2 / NB);
  constexpr int NT = WM * WN * 64;
  constexpr int W_BYTES = 8 * 2048;
  constexpr int X_BYTES = BM * 128;
  // NS == 0: register-staged, 2 LDS buffers (simple reference variant)
  // NS >= 2: LDS-DMA (global_load_lds) ring of NS stages, NS-1 stages of loads in flight
  // NS == 5: LDS-DMA ring of 3 stages driven by the staggered two-group ("ping-pong") schedule
  constexpr bool GLDS = NS > 0;
  // NS == 6: LDS-DMA ring of 3 stages, ONE barrier per 128-k block placed two k-steps before the stage
  //          switch, fragment pipeline (W reads 2 steps ahead, unpack 1 step ahead) running across it
  constexpr bool PINGPONG = (NS == 5);
  constexpr bool CONTPIPE = (NS == 6);
  constexpr int NSTAGE = (PINGPONG || CONTPIPE) ? 3 : NS;
  constexpr int SC_BYTES = (GLDS && GROUPED) ? WM * WN * 512 : 0;  // per-wave slot of group scales
  constexpr int STAGE = W_BYTES + X_BYTES + SC_BYTES;
  constexpr int W_CHUNKS = W_BYTES / 16;  // 1024
  constexpr int X_CHUNKS = X_BYTES / 16;
  constexpr int WPT = W_CHUNKS / NT;  // chunks per thread
  constexpr int XPT = X_CHUNKS / NT;
  static_assert(W_CHUNKS % NT == 0 && X_CHUNKS % NT == 0, "tile/threads mismatch");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = (wave % WN) / (2 / NB);    // which jt set
  const int bsel = (wave % WN) % (2 / NB);  // which b half (NB == 1 only; 0 otherwise)

  // ---- XCD-aware tile order: block b runs on XCD b % 8; give each XCD a contiguous run of the
  // panel-major tile sequence (panels of PW strips x all m-tiles) so co-resident workgroups of one
  // XCD share weight strips / activation rows in that XCD's L2.  Speed only, never correctness.
  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  int lin;
  {
    const int q = ntiles >> 3, rr = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    lin = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  int tile_m, tile_n;
  {
    const int full = (tiles_n / PW) * PW * tiles_m;
    if (lin < full) {
      const int panel = lin / (PW * tiles_m), within = lin % (PW * tiles_m);
      tile_m = within / PW;
      tile_n = panel * PW + within % PW;
    } else {
      const int rem = lin - full, pw = tiles_n % PW;
      tile_m = rem / pw;
      tile_n = (tiles_n / PW) * PW + rem % pw;
    }
  }
  const int sp = blockIdx.y;
  const int m0 = tile_m * BM;
  const int ng0 = tile_n * 4;
  const int ngroups = N >> 6;

  const int NKB = K >> 7;  // 128-k blocks
  const int kb_begin = (int)(((long long)NKB * sp) / ksplit);
  const int kb_end = (int)(((long long)NKB * (sp + 1)) / ksplit);
  const size_t rowbytes = (size_t)N * 8;

  // ---- per-thread staging sources (offsets relative to the stage's first k-tile / k byte) ----
  unsigned wsrc[WPT], xsrc[XPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int cw = tid + i * NT;  // LDS chunk index inside the W region
    const int ktl = cw >> 7, gl = (cw >> 5) & 3, pos = cw & 31;
    const int cc = pos >> 2, kq = (pos & 3) ^ gl;
    int ngx = ng0 + gl;
    if (ngx >= ngroups) ngx = ngroups - 1;
    wsrc[i] = (unsigned)(ktl * rowbytes + (size_t)ngx * 512 + (4 * cc + kq) * 16);
  }
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int cx = tid + i * NT;
    const int row = cx >> 3, pos = cx & 7;
    const int chunk = pos ^ ((row >> 1) & 7);
    int grow = m0 + row;
    if (grow >= M) grow = M - 1;
    xsrc[i] = (unsigned)((size_t)(grow - m0) * K + chunk * 16);
  }
  const unsigned char* Abase = reinterpret_cast<const unsigned char*>(A) + (size_t)m0 * K;

  // ---- per-lane LDS read offsets ----
  const int li = lane & 31, h = lane >> 5;
  const int g = li >> 3, c = li & 7;
  // Which jt of the 16-byte chunk this lane consumes.  A lane reads 8 of the 16 bytes (JW <= 2); lanes
  // c < 4 take one half, lanes c >= 4 the OTHER half: the 32 lanes of a ds_read_b64 group then cover
  // all 64 banks exactly once (a fixed half would be a 2-way conflict).  This is only another
  // row<->column permutation of the MFMA tile (a sibling wave takes the complementary halves); the
  // epilogue and the group-scale addresses use the same lane-dependent jt.
  //   JW == 2: half = wn ^ (c >> 2), jt = 2*half + jj
  //   JW == 1: half = (wn >> 1) ^ (c >> 2), jt = 2*half + (wn & 1)   (8 bytes read, one dword used)
  const int half = (JW == 2) ? (wn ^ (c >> 2)) : (JW == 1) ? ((wn >> 1) ^ (c >> 2)) : 0;
  const int esel = (JW == 1) ? (wn & 1) : 0;
  const int jt0 = (JW == 4) ? 0 : 2 * half + esel;  // first jt of this lane
  unsigned wrd[4];  // + t*4096
#pragma unroll
  for (int kq = 0; kq < 4; ++kq)
    wrd[kq] = h * 2048 + g * 512 + (4 * c + (kq ^ g)) * 16 + ((JW == 4) ? 0 : half * 8);
  unsigned xrd[4];  // per k-step t; + mt*32*128
#pragma unroll
  for (int t = 0; t < 4; ++t)
    xrd[t] = W_BYTES + (wm * MTW * 32 + li) * 128 + (((2 * t + h) ^ ((li >> 1) & 7)) * 16);

  const _Float16* sptr = nullptr;  // register path (NS == 0)
  unsigned scsrc = 0;              // LDS-DMA path: byte offset of this lane's 16 B inside the group row
  unsigned scrd = 0;               // LDS byte offset (inside the stage) of this lane's 2*JW scales
  if constexpr (GROUPED) {
    int ngx = ng0 + g;
    if (ngx >= ngroups) ngx = ngroups - 1;
    sptr = s3 + (size_t)ngx * 64 + c * 8 + 2 * jt0;
    int ngl = ng0 + ((lane & 31) >> 3);
    if (ngl >= ngroups) ngl = ngroups - 1;
    scsrc = (unsigned)((ngl * 64 + (lane & 7) * 8) * 2);
    scrd = W_BYTES + X_BYTES + wave * 512 + (g * 64 + c * 8 + 2 * jt0) * 2;
  }

  v16i acc[MTW][JW][NB];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int jj = 0; jj < JW; ++jj)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][jj][b][r] = 0;

  v4u wreg[WPT], xreg[XPT];  // register staging (unused with GLDS)

  // part / nparts: issue only the part-th of nparts contiguous groups of this thread's DMA instructions
  // (compile-time constants at every call site); the default issues the whole stage.
  auto issue_loads = [&](const int kb, const int buf, const int part = 0, const int nparts = 1) {
    const unsigned char* wb = B + (size_t)(kb * 8) * rowbytes;
    const unsigned char* xb = Abase + (size_t)kb * 128;
#if defined(QQQ_ABLATE) && (QQQ_ABLATE & 2)  // ablation: no global->LDS traffic
    if (kb >= 0) return;
#endif
    if constexpr (GLDS) {
      const unsigned st = lds_base + buf * STAGE + wave * 1024;
      const int lo = (WPT + XPT) * part / nparts, hi = (WPT + XPT) * (part + 1) / nparts;
#pragma unroll
      for (int i = 0; i < WPT; ++i)
        if (i >= lo && i < hi) glds16(wb + wsrc[i], st + i * (NT * 16));
#pragma unroll
      for (int i = 0; i < XPT; ++i)
        if (WPT + i >= lo && WPT + i < hi) glds16(xb + xsrc[i], st + W_BYTES + i * (NT * 16));
      if (part != nparts - 1) return;
      if constexpr (GROUPED) {
        // this wave's private copy of the tile's 256 group scales (512 B): lanes 0..31, 16 B each
        const unsigned scdst =
            __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE + W_BYTES + X_BYTES + wave * 512);
        if (lane < 32)  // exec-masked DMA: only 32 x 16 B are written
          glds16(reinterpret_cast<const unsigned char*>(s3 + (size_t)kb * N) + scsrc, scdst);
      }
    } else {
#pragma unroll
      for (int i = 0; i < WPT; ++i) wreg[i] = *reinterpret_cast<const v4u*>(wb + wsrc[i]);
#pragma unroll
      for (int i = 0; i < XPT; ++i) xreg[i] = *reinterpret_cast<const v4u*>(xb + xsrc[i]);
    }
  };
  auto commit_loads = [&](const int buf) {  // register-staged variant: write the LDS image
    if constexpr (!GLDS) {
      unsigned char* st = smem + buf * STAGE;
#pragma unroll
      for (int i = 0; i < WPT; ++i) *reinterpret_cast<v4u*>(st + (tid + i * NT) * 16) = wreg[i];
#pragma unroll
      for (int i = 0; i < XPT; ++i)
        *reinterpret_cast<v4u*>(st + W_BYTES + (tid + i * NT) * 16) = xreg[i];
    }
  };

  // per-group scales of this lane's (jt in [JW*wn, +JW), b) columns: 2*JW consecutive fp16
  typedef _Float16 hsc __attribute__((ext_vector_type(2 * JW)));
  hsc sc_cur = {}, sc_nxt = {};
  if constexpr (GROUPED)
    if (kb_begin < kb_end) sc_cur = *reinterpret_cast<const hsc*>(sptr + (size_t)kb_begin * N);

  // One k-step (32 k) of fragments: raw packed weight words [kq][jj] (only this wave's jt are read
  // from LDS) + the activation operands of this wave's m-tiles.
  struct Frag {
    unsigned wq[4][JW];
    v4i xop[MTW];
  };
  auto read_frag = [&](const unsigned char* st, const int t, Frag& f) {
#if defined(QQQ_ABLATE) && (QQQ_ABLATE & 1)  // ablation: no LDS fragment reads
    if (t >= 0) {
#pragma unroll
      for (int kq = 0; kq < 4; ++kq)
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) asm volatile("" : "+v"(f.wq[kq][jj]));
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) asm volatile("" : "+v"(f.xop[mt]));
      return;
    }
#endif
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      const unsigned char* p = st + wrd[kq] + t * 4096;
      if constexpr (JW == 4) {
        const v4u v = *reinterpret_cast<const v4u*>(p);
        f.wq[kq][0] = v[0]; f.wq[kq][1] = v[1]; f.wq[kq][2] = v[2]; f.wq[kq][3] = v[3];
      } else if constexpr (JW == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        f.wq[kq][0] = v.x; f.wq[kq][1] = v.y;
      } else {
        const uint2 v = *reinterpret_cast<const uint2*>(p);  // conflict-free b64, one dword used
        f.wq[kq][0] = esel ? v.y : v.x;
      }
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
      f.xop[mt] = *reinterpret_cast<const v4i*>(st + xrd[t] + mt * (32 * 128));
  };
  // unpacked MFMA weight operands of one k-step
  struct Ops {
    v4i a[JW][NB];
  };
  auto unpack_frag = [&](const Frag& f, Ops& o) {
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      h2 sb0 = {(_Float16)0, (_Float16)0}, sb1 = sb0;
      if constexpr (GROUPED) {
        sb0 = (h2){sc_cur[2 * jj], sc_cur[2 * jj]};
        sb1 = (h2){sc_cur[2 * jj + 1], sc_cur[2 * jj + 1]};
      }
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        const unsigned q = f.wq[kq][jj];
#if defined(QQQ_ABLATE) && (QQQ_ABLATE & 4)  // ablation: no unpack VALU
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) o.a[jj][bi][kq] = (int)q;
#else
        if constexpr (NB == 2) {
          int w0, w1;
          unpack_pair<GROUPED>(q, sb0, sb1, w0, w1);
          o.a[jj][0][kq] = w0;
          o.a[jj][1][kq] = w1;
        } else if constexpr (GROUPED) {  // one b half per wave (bsel is wave-uniform)
          o.a[jj][0][kq] = (int)dequant_group4(q >> (8 * bsel), bsel ? sb1 : sb0);
        } else {
          o.a[jj][0][kq] = (int)((q << (4 * bsel)) & QQQ_NIB_MASK);
        }
#endif
      }
    }
  };
  auto mfma_ops = [&](const Ops& o, const Frag& f) {
#pragma unroll
    for (int jj = 0; jj < JW; ++jj)
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
          acc[mt][jj][bi] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.a[jj][bi], f.xop[mt], acc[mt][jj][bi], 0, 0, 0);
        }
  };
  auto mma_frag = [&](const Frag& f) {
    Ops o;
    unpack_frag(f, o);
    mfma_ops(o, f);
  };
  // software pipeline inside a stage: the LDS reads of k-step t+1 are issued before the MFMAs of
  // k-step t, so their latency hides under the matrix pipe instead of stalling the (in-order) wave
  auto compute_stage = [&](const int buf) {
    const unsigned char* st = smem + buf * STAGE;
    Frag f0 = {}, f1 = {};
    read_frag(st, 0, f0);
    read_frag(st, 1, f1);
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(f0);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(st, 2, f0);
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(f1);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(st, 3, f1);
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(f0);
    mma_frag(f1);
  };

  if constexpr (!GLDS) {
    // ---- register-staged variant: 2 LDS buffers, one barrier per 128-k block ----
    if (kb_begin < kb_end) {
      issue_loads(kb_begin, 0);
      commit_loads(0);
      __syncthreads();
    }
    for (int kb = kb_begin; kb < kb_end; ++kb) {
      const int buf = (kb - kb_begin) & 1;
      const bool more = (kb + 1 < kb_end);
      if (more) {
        issue_loads(kb + 1, buf ^ 1);
        if constexpr (GROUPED) sc_nxt = *reinterpret_cast<const hsc*>(sptr + (size_t)(kb + 1) * N);
      }
      compute_stage(buf);
      if (more) commit_loads(buf ^ 1);
      if constexpr (GROUPED) sc_cur = sc_nxt;
      __syncthreads();
    }
  } else {
    // ---- LDS-DMA ring: NS stages, NS-1 stages of loads in flight, one barrier per 128-k block.
    // The DMA loads are inline asm (invisible to hipcc's counters): we count them ourselves.  Every
    // thread issues exactly NL wave-instructions per stage, in stage order, so "stage j has landed"
    // == "at most NL * (number of younger stages issued) loads outstanding".
    constexpr int NL = WPT + XPT + (GROUPED ? 1 : 0);
    static_assert(NL * (NSTAGE - 2) <= 63, "vmcnt is a 6-bit counter");
    auto wait_younger = [&](const int younger) {  // wave-uniform
      if (younger <= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (younger == 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
      } else if (younger == 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL > 63 ? 63 : 2 * NL) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NL > 63 ? 63 : 3 * NL) : "memory");
      }
    };
#ifdef QQQ_TRACE
    int tr_n = 0;
    auto stamp = [&](int tag) {
      if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && C && tr_n < 400) {  // C is idle when ksplit == 1
        const unsigned long long tm = __builtin_readcyclecounter();
        C[(wave * 400 + tr_n) * 4 + 0] = tag;
        C[(wave * 400 + tr_n) * 4 + 1] = (int)(tm & 0xffffffffu);
        C[(wave * 400 + tr_n) * 4 + 2] = (int)(tm >> 32);
        ++tr_n;
      }
    };
#define QQQ_STAMP(x) stamp(x)
#else
#define QQQ_STAMP(x)
#endif
    const int nkb = kb_end - kb_begin;
    if constexpr (CONTPIPE) {
      // ---- continuous fragment pipeline over a 3-stage DMA ring.
      // step u = 4*i + t:   ds_read W(u+2), X(u+1)  |  unpack W(u+1) -> ops  |  8 MFMA(u)
      // The only barrier of block i sits at the start of its step t=2: by then every wave has waited for
      // its DMA of stage i+1 (issued one block earlier), so from t=2 on stage i+1 may be read; stage i+2
      // is issued right behind that barrier into the buffer of stage i-1 (last read at (i-1, t=1)).
      // Reads past the last step are redirected to the last stage (harmless), never branched around, so
      // each step is one straight-line scheduling region with an enforced MFMA/VALU/DS interleave.
      unsigned wraw[2][4][JW];
      v4i xfr[2][MTW];
      Ops ops2[2];
      auto read_w = [&](const unsigned char* st, const int t, unsigned (&w)[4][JW]) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          const unsigned char* p = st + wrd[kq] + t * 4096;
          if constexpr (JW == 4) {
            const v4u v = *reinterpret_cast<const v4u*>(p);
            w[kq][0] = v[0]; w[kq][1] = v[1]; w[kq][2] = v[2]; w[kq][3] = v[3];
          } else if constexpr (JW == 2) {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
            w[kq][0] = v.x; w[kq][1] = v.y;
          } else {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
            w[kq][0] = esel ? v.y : v.x;
          }
        }
      };
      auto read_x = [&](const unsigned char* st, const int t, v4i (&x)[MTW]) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) x[mt] = *reinterpret_cast<const v4i*>(st + xrd[t] + mt * (32 * 128));
      };
      auto unpack_w = [&](const unsigned (&w)[4][JW], Ops& o) {
        Frag f;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
#pragma unroll
          for (int jj = 0; jj < JW; ++jj) f.wq[kq][jj] = w[kq][jj];
        unpack_frag(f, o);
      };
      auto mfma_x = [&](const Ops& o, const v4i (&x)[MTW]) {
#pragma unroll
        for (int jj = 0; jj < JW; ++jj)
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
              acc[mt][jj][bi] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.a[jj][bi], x[mt], acc[mt][jj][bi], 0, 0, 0);
      };
      if (nkb > 0) {
        issue_loads(kb_begin, 0);
        if (nkb > 1) issue_loads(kb_begin + 1, 1);
        wait_younger(nkb > 1 ? 1 : 0);
        __syncthreads();
        if constexpr (GROUPED) sc_cur = *reinterpret_cast<const hsc*>(smem + scrd);
        read_w(smem, 0, wraw[0]);
        read_x(smem, 0, xfr[0]);
        read_w(smem, 1, wraw[1]);
        unpack_w(wraw[0], ops2[0]);
      }
      for (int i = 0; i < nkb; ++i) {
        const unsigned char* st = smem + (i % 3) * STAGE;
        const bool has_next = (i + 1 < nkb);
        const unsigned char* stn = has_next ? smem + ((i + 1) % 3) * STAGE : st;  // redirect past-the-end reads
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          QQQ_STAMP(0 + t);
          if (t == 2) {
            if (has_next) {
              wait_younger(0);  // this wave's DMA of stage i+1 (the only one in flight) has landed
              QQQ_STAMP(10);
              __syncthreads();  // ... and everybody else's
              QQQ_STAMP(11);
            }
            QQQ_STAMP(12);
          }
          {
            // DMA issue of a stage spread over the 3 k-steps behind the barrier (2 of the wave's 6 instructions
            // each): a burst of all 8 waves x 6 KB right behind the barrier overruns the address pipe (~30
            // cycles per 1 KB instruction, measured with QQQ_TRACE), and an in-order wave stuck in VMEM issue
            // cannot issue its MFMAs either.  -3 % cycles, 1-3 % time on real data.
            constexpr int SPREAD = 3;
            const int part = (t + 2) & 3;  // t=2 -> 0, t=3 -> 1, t=0 -> 2, t=1 -> 3
            const int stg = (t >= 2) ? i + 2 : i + 1;
            if (part < SPREAD && stg < nkb && stg >= 2) issue_loads(kb_begin + stg, stg % 3, part, SPREAD);
          }
          __builtin_amdgcn_sched_barrier(0);
          // fragment reads: W two steps ahead, X one step ahead
          read_w((t + 2 < 4) ? st : stn, (t + 2) & 3, wraw[t & 1]);
          read_x((t + 1 < 4) ? st : stn, (t + 1) & 3, xfr[(t + 1) & 1]);
          if constexpr (GROUPED)
            if (t == 3) sc_cur = *reinterpret_cast<const hsc*>(stn + scrd);  // scales of the block being unpacked
          __builtin_amdgcn_sched_barrier(0);
          QQQ_STAMP(20 + t);
          // hand-interleaved issue order: one MFMA of step u, then the unpack of one packed word of step
          // u+1 (hipcc otherwise issues the 8 MFMAs back to back and leaves the VALU work uncovered)
          constexpr int NM = MTW * JW * NB;  // MFMAs per step
          constexpr int NWD = 4 * JW;        // packed words per step
#pragma unroll
          for (int q = 0; q < (NM > NWD ? NM : NWD); ++q) {
            if (q < NM) {
              const int jj = q / (MTW * NB), mt = (q / NB) % MTW, bi = q % NB;
              acc[mt][jj][bi] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops2[t & 1].a[jj][bi], xfr[t & 1][mt],
                                                                      acc[mt][jj][bi], 0, 0, 0);
            }
            if (q < NWD) {
              const int kq = q & 3, jj = q >> 2;
              const unsigned wq_ = wraw[(t + 1) & 1][kq][jj];
              h2 sb0 = {(_Float16)0, (_Float16)0}, sb1 = sb0;
              if constexpr (GROUPED) {
                sb0 = (h2){sc_cur[2 * jj], sc_cur[2 * jj]};
                sb1 = (h2){sc_cur[2 * jj + 1], sc_cur[2 * jj + 1]};
              }
              // the empty asm pins the unpack HERE (LLVM would otherwise sink it to its use, one step later)
              if constexpr (NB == 2) {
                int w0, w1;
                unpack_pair<GROUPED>(wq_, sb0, sb1, w0, w1);
                asm volatile("" : "+v"(w0), "+v"(w1));
                ops2[(t + 1) & 1].a[jj][0][kq] = w0;
                ops2[(t + 1) & 1].a[jj][1][kq] = w1;
              } else if constexpr (GROUPED) {
                int w0 = (int)dequant_group4(wq_ >> (8 * bsel), bsel ? sb1 : sb0);
                asm volatile("" : "+v"(w0));
                ops2[(t + 1) & 1].a[jj][0][kq] = w0;
              } else {
                int w0 = (int)((wq_ << (4 * bsel)) & QQQ_NIB_MASK);
                asm volatile("" : "+v"(w0));
                ops2[(t + 1) & 1].a[jj][0][kq] = w0;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    } else if constexpr (!PINGPONG) {
      int issued = 0;  // stages issued so far (relative index)
      for (; issued < NSTAGE - 1 && issued < nkb; ++issued) issue_loads(kb_begin + issued, issued % NSTAGE);
      if (nkb > 0) {
        wait_younger(issued - 1);
        __syncthreads();
      }
      for (int i = 0; i < nkb; ++i) {
        // refill the buffer consumed in iteration i-1 (every wave has passed that iteration's barrier)
        if (issued < nkb) {
          issue_loads(kb_begin + issued, issued % NSTAGE);
          ++issued;
        }
        if constexpr (GROUPED)
          sc_cur = *reinterpret_cast<const hsc*>(smem + (i % NSTAGE) * STAGE + scrd);
        compute_stage(i % NSTAGE);
        if (i + 1 < nkb) {
          wait_younger(issued - (i + 2));  // stage i+1 must have landed; stages i+2.. may stay in flight
          __syncthreads();
        }
      }
    } else {
      // ---- staggered two-group schedule ("ping-pong").  Every k-step is cut in a LOAD phase (LDS
      // fragment reads for the NEXT k-step, int4 unpack of the current one, DMA issue / landing waits)
      // and an MFMA phase (8 matrix instructions), each closed by a workgroup barrier.  The second half
      // of the waves runs one phase behind the first, so on every SIMD one wave's MFMA burst overlaps
      // its partner's LOAD phase instead of both stalling on LDS at the same time.
      //   DMA of stage j is issued in LOAD(j-2, t=0) into buffer j % 3 (last read two barriers ago);
      //   its landing is awaited by every wave in LOAD(j-1, t=2); first read is in LOAD(j-1, t=3).
      const bool late = wave >= (WM * WN) / 2;  // wave-uniform
      auto phase_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);  // register-only MFMAs would otherwise drift across it
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      Frag fr[2] = {};
      Ops ops;
      if (nkb > 0) {
        issue_loads(kb_begin, 0);
        if (nkb > 1) issue_loads(kb_begin + 1, 1);
        wait_younger(nkb > 1 ? 1 : 0);
        __syncthreads();
        read_frag(smem, 0, fr[0]);
        if (late) phase_barrier();
      }
      for (int i = 0; i < nkb; ++i) {
        const unsigned char* st = smem + (i % 3) * STAGE;
        const unsigned char* stn = smem + ((i + 1) % 3) * STAGE;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // ---------------- LOAD phase ----------------
          QQQ_STAMP(0 + t);
          if (t == 0) {
            if (i + 2 < nkb) issue_loads(kb_begin + i + 2, (i + 2) % 3);
            if constexpr (GROUPED) sc_cur = *reinterpret_cast<const hsc*>(st + scrd);
          }
          if (t == 2 && i + 1 < nkb) wait_younger(i + 2 < nkb ? 1 : 0);
          if (t < 3) {
            read_frag(st, t + 1, fr[(t + 1) & 1]);
          } else if (i + 1 < nkb) {
            read_frag(stn, 0, fr[0]);
          }
          QQQ_STAMP(10 + t);
          unpack_frag(fr[t & 1], ops);
          QQQ_STAMP(20 + t);
          phase_barrier();
          QQQ_STAMP(30 + t);
          // ---------------- MFMA phase ----------------
#ifndef QQQ_NO_SETPRIO
          __builtin_amdgcn_s_setprio(1);  // the partner wave on this SIMD is in its LOAD phase: MFMA issue first
#endif
          mfma_ops(ops, fr[t & 1]);
#ifndef QQQ_NO_SETPRIO
          __builtin_amdgcn_s_setprio(0);
#endif
          QQQ_STAMP(40 + t);
          phase_barrier();
          QQQ_STAMP(50 + t);
        }
      }
      if (nkb > 0 && !late) phase_barrier();  // both groups execute the same number of barriers
    }
  }

  // ---- in-launch split-K (nslots > 0): the K slices of one tile meet in `nslots` tile-sized int32 slots of C.
  // Workgroups take a ticket in ARRIVAL order.  Arrival t < ksplit-1 deposits its partial tile in slot
  // t % nslots -- adding what the slot already holds when it is the slot's (t / nslots + 1)-th deposit --
  // and leaves; the last arrival adds every slot to its accumulators and runs the normal epilogue.
  // Whoever is waited for has already arrived (is past its main loop), so the spins are short and cannot
  // deadlock whatever the dispatch order.  Slot images are lane-linear (16 B per lane, full lines); they
  // are written through (sc0 sc1) and read behind an agent-scope acquire, because the K slices of a tile
  // run on different XCDs (= different, mutually non-coherent L2s).  int32 adds commute: bit-exact.
  // tickets[(1 + nslots) * tile + {0: arrivals, 1 + s: deposits completed in slot s}], all zero again on exit.
  bool finish = (ksplit == 1);
  if (ksplit > 1 && nslots > 0) {
    constexpr int NQ4 = MTW * JW * NB * 4;
    __shared__ int xch;
    int* tk = tickets + (size_t)lin * (1 + nslots);
    if (tid == 0) xch = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int t = xch;
    const unsigned voff = lane * 16;  // lane offset inside a 1 KiB wave-row of a slot image
    auto slot_base = [&](const int s_) {  // wave-uniform: lets the accesses use the SGPR-base + VGPR-offset form
      return reinterpret_cast<const unsigned char*>(C + ((size_t)s_ * ntiles + lin) * ((size_t)BM * 256) +
                                                    (size_t)wave * NQ4 * 256);
    };
    auto wait_done = [&](const int s_, const int want) {
      if (tid == 0)
        while (__hip_atomic_load(tk + 1 + s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want)
          __builtin_amdgcn_s_sleep(4);
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };
    auto add_slot = [&](const int s_) {
      const unsigned char* p = slot_base(s_);
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) {
#pragma unroll
          for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const v4i v = *reinterpret_cast<const v4i*>(p + (((mt * JW + jj) * NB + bi) * 4 + gq) * 1024 + (size_t)voff);
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[mt][jj][bi][4 * gq + r] += v[r];
            }
          __builtin_amdgcn_sched_barrier(0);  // at most 4*NB loads in flight: no spill next to 128 accumulators
        }
    };
    // one loop for both roles (a depositor folds at most its own slot, the last arrival every used slot)
    const bool last = (t == ksplit - 1);
    const int slot = last ? 0 : t % nslots, gen = last ? 0 : t / nslots;
    const int used = (ksplit - 1 < nslots) ? ksplit - 1 : nslots;
    const int nfold = last ? used : (gen > 0 ? 1 : 0);
    for (int i = 0; i < nfold; ++i) {
      const int s_ = slot + i;
      wait_done(s_, last ? (ksplit - 1 - s_ + nslots - 1) / nslots : gen);
      add_slot(s_);
    }
    if (!last) {
      // scalar base + one 32-bit lane offset: the 4*MTW*JW*NB stores share a single address VGPR
      const unsigned char* sbase = slot_base(slot);
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int jj = 0; jj < JW; ++jj)
#pragma unroll
          for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const v4i v = {acc[mt][jj][bi][4 * gq + 0], acc[mt][jj][bi][4 * gq + 1], acc[mt][jj][bi][4 * gq + 2],
                             acc[mt][jj][bi][4 * gq + 3]};
              const unsigned char* sb = sbase + (((mt * JW + jj) * NB + bi) * 4 + gq) * 1024;
              asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(voff), "v"(v), "s"(sb) : "memory");
            }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // every wave's part of the deposit has reached memory
      if (tid == 0) __hip_atomic_store(tk + 1 + slot, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid <= used) __hip_atomic_store(tk + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // workspace zero on return
    finish = true;
  }

  // ---- epilogue ----
  int mrow[MTW];
  float a_s[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    mrow[mt] = m0 + (wm * MTW + mt) * 32 + li;
    a_s[mt] = (mrow[mt] < M && finish) ? s1[mrow[mt]] : 0.f;
  }
  const int n_lane = ng0 * 64 + 4 * h;  // + 64*g' + 16*jt + 8*b  (g' = r >> 2), + (r & 3)
  if (finish) {
    // fp16 tile -> LDS (row-major, 16-byte chunks XOR-swizzled by the row so that the 8-byte writes of a
    // lane column and the 16-byte reads of a row are both conflict-light) -> full 128-byte-line stores.
    // Straight-from-register stores would be 8 bytes per lane scattered over 32 rows (measured: 2.6x write
    // amplification at the fabric, store-issue bound tail).
    __syncthreads();  // every wave is done reading the operand ring
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      // D-tile rows of this lane have c = 4*h + (r & 3): c >> 2 == h, so the lane's jt is uniform over r
      const int jt = (JW == 4) ? jj : (JW == 2) ? 2 * (wn ^ h) + jj : 2 * ((wn >> 1) ^ h) + (wn & 1);
#pragma unroll
      for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int nl = 4 * h + 64 * gq + 16 * jt + 8 * ((NB == 2) ? bi : bsel);  // column inside the tile
          const int n = ng0 * 64 + nl;
          const int ncl = (n < N) ? n : 0;  // scales of a dropped column: any legal address
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            const int ml = (wm * MTW + mt) * 32 + li;
            const int v0 = acc[mt][jj][bi][4 * gq + 0], v1 = acc[mt][jj][bi][4 * gq + 1];
            const int v2 = acc[mt][jj][bi][4 * gq + 2], v3 = acc[mt][jj][bi][4 * gq + 3];
            const h4 o = epilogue_vals4(v0, v1, v2, v3, ncl, a_s[mt], s2);
            *reinterpret_cast<h4*>(smem + ml * 512 + (((nl >> 3) ^ (ml & 31)) << 4) + ((nl & 7) << 1)) = o;
            if (acc_out && n < N && mrow[mt] < M) {
              v4i a = {v0, v1, v2, v3};
              *reinterpret_cast<v4i*>(acc_out + (size_t)mrow[mt] * N + n) = a;
            }
          }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < BM * 32; idx += NT) {
      const int ml = idx >> 5, j = idx & 31;
      const int m = m0 + ml, n = ng0 * 64 + j * 8;
      if (m < M && n < N) {
        h8 v = *reinterpret_cast<const h8*>(smem + ml * 512 + ((j ^ (ml & 31)) << 4));
        if (bias) v = v + *reinterpret_cast<const h8*>(bias + n);  // fp16 add after the fp16 round
        *reinterpret_cast<h8*>(D + (size_t)m * N + n) = v;
      }
    }
    return;
  }
  // split-K: int32 partial sums straight from the accumulators into slab sp of C
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    const int jt = (JW == 4) ? jj : (JW == 2) ? 2 * (wn ^ h) + jj : 2 * ((wn >> 1) ^ h) + (wn & 1);
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = n_lane + 64 * gq + 16 * jt + 8 * ((NB == 2) ? bi : bsel);
        if (n >= N) continue;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const int m = mrow[mt];
          if (m >= M) continue;
          v4i v = {acc[mt][jj][bi][4 * gq + 0], acc[mt][jj][bi][4 * gq + 1], acc[mt][jj][bi][4 * gq + 2],
                   acc[mt][jj][bi][4 * gq + 3]};
          *reinterpret_cast<v4i*>(C + ((size_t)sp * M + m) * N + n) = v;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------
// fused per-token dynamic int8 quantisation (QuantLinear.dynamic_quant, qlinear_marlin.py:265-268)
// one workgroup per token row; the row is kept in registers between the two passes.
// ------------------------------------------------------------------------------------------
template <int VPT>  // 16-byte vectors (8 halfs) per thread; covers K <= 256*8*VPT
__global__ __launch_bounds__(256) void qqq_dynamic_quant_kernel(const _Float16* __restrict__ x,
                                                                int8_t* __restrict__ xq,
                                                                float* __restrict__ s1, const int K) {
  __shared__ float wmax[4];
  const int row = blockIdx.x;
  const int tid = threadIdx.x;
  const int nvec = K >> 3;
  const h8* xr = reinterpret_cast<const h8*>(x + (size_t)row * K);
  h8 v[VPT];
  // |x| max in packed fp16 (max is exact in any precision): clear the sign bits, v_pk_max_f16
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  h2 amax2 = {(_Float16)0, (_Float16)0};
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int idx = tid + i * 256;
    if (idx < nvec) {
      v[i] = xr[idx];
      const u4v bits = __builtin_bit_cast(u4v, v[i]);
      const h2 m0 = __builtin_elementwise_max(__builtin_bit_cast(h2, bits.x & 0x7fff7fffu), __builtin_bit_cast(h2, bits.y & 0x7fff7fffu));
      const h2 m1 = __builtin_elementwise_max(__builtin_bit_cast(h2, bits.z & 0x7fff7fffu), __builtin_bit_cast(h2, bits.w & 0x7fff7fffu));
      amax2 = __builtin_elementwise_max(amax2, __builtin_elementwise_max(m0, m1));
    }
  }
  float amax = fmaxf((float)amax2[0], (float)amax2[1]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if ((tid & 63) == 0) wmax[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  // torch on GPU lowers `.div(127.0)` to a multiply by the fp32 reciprocal; result kept in fp16
  const float scale = (float)(_Float16)__fmul_rn(amax, 1.0f / 127.0f);
  if (tid == 0) s1[row] = scale;
  // x / scale must be the correctly rounded fp32 quotient before rint() (torch semantics).  An IEEE division costs
  // ~12 VALU ops per element and made this kernel compute-bound; rint(x * (1/scale)) equals rint(x / scale) unless
  // the product lies within ~1e-4 of a half-integer (|q| <= 128 and the reciprocal-multiply is good to a few ulp),
  // so only vectors with such an element (a few per thousand) take the exact division.
  const float rinv = (scale > 0.f) ? __frcp_rn(scale) : 0.f;  // all-zero row: reference gives NaN -> int8 (UB); we emit 0
  int2* qr = reinterpret_cast<int2*>(xq + (size_t)row * K);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int idx = tid + i * 256;
    if (idx < nvec) {
      float q[8];
      bool near_tie = false;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float p = (float)v[i][e] * rinv;
        q[e] = rintf(p);
        near_tie |= fabsf(p - q[e]) > 0.4995f;
      }
      if (near_tie) {
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = (scale > 0.f) ? rintf(__fdiv_rn((float)v[i][e], scale)) : 0.f;
      }
      unsigned lo = 0, hi = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned byte = (unsigned)((int)fminf(fmaxf(q[e], -128.f), 127.f)) & 0xFFu;
        if (e < 4)
          lo |= byte << (8 * e);
        else
          hi |= byte << (8 * (e - 4));
      }
      qr[idx] = make_int2((int)lo, (int)hi);
    }
  }
}

__global__ __launch_bounds__(256) void qqq_add_bias_kernel(_Float16* __restrict__ D,
                                                           const _Float16* __restrict__ bias,
                                                           const long long total_vec, const int nvec) {
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total_vec;
       it += (long long)gridDim.x * 256) {
    h8 d = reinterpret_cast<h8*>(D)[it];
    const h8 b = reinterpret_cast<const h8*>(bias)[it % nvec];
    d = d + b;  // fp16 add, RN -- same as torch's half + half
    reinterpret_cast<h8*>(D)[it] = d;
  }
}

// ------------------------------------------------------------------------------------------
// hardware probes (tests/test_gpu_probe.py)
// ------------------------------------------------------------------------------------------
__global__ void qqq_probe_mfma16_kernel(const v4i* a, const v4i* b, v4i* out) {
  const int l = threadIdx.x;
  v4i acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[l], b[l], acc, 0, 0, 0);
  out[l] = acc;
}
__global__ void qqq_probe_mfma32_kernel(const v4i* a, const v4i* b, v16i* out) {
  const int l = threadIdx.x;
  v16i acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0;
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[l], b[l], acc, 0, 0, 0);
  out[l] = acc;
}
// Read-bandwidth probe: every workgroup (512 threads, UNR independent 16-byte loads in flight per thread) streams
// `bytes_per_wg` bytes starting at src + wg_stride * blockIdx.x, `reps` times, and folds them into one word.
// wg_stride == 0: all workgroups read the same L2-resident window (per-CU L2 -> L1 fill rate);
// wg_stride == bytes_per_wg: disjoint windows (HBM / Infinity-Cache streaming rate).
template <int UNR>
__global__ __launch_bounds__(512) void qqq_probe_fill_kernel(const v4u* __restrict__ src, const size_t wg_stride,
                                                             const size_t bytes_per_wg, const int reps,
                                                             unsigned* __restrict__ sink) {
  const v4u* p = reinterpret_cast<const v4u*>(reinterpret_cast<const unsigned char*>(src) + wg_stride * blockIdx.x);
  const size_t nvec = bytes_per_wg / 16;
  v4u acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i + (UNR - 1) * 512 < nvec; i += UNR * 512) {
      v4u v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = p[i + u * 512];
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc ^= v[u];
    }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;  // keep the loads alive
}

__global__ void qqq_probe_glds_kernel(const v4u* src, const int* perm, v4u* dst) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2048];
  const int l = threadIdx.x;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  // same helper the tiled kernel uses; destination deliberately not at the start of the array
  glds16(src + perm[l], lds_base + 1024);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  dst[l] = reinterpret_cast<const v4u*>(lds + 1024)[l];
}

// ------------------------------------------------------------------------------------------
// host side: validation (mirrors the reference's), dispatch, C-ABI
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail_hip(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return QQQ_ERR_HIP;
}

struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev && dev >= 0) {
      changed = (hipSetDevice(dev) == hipSuccess);
    }
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

// reference: is_valid_config (csrc/qqq_gemm.cu:867-897)
static bool ref_valid_config(int thread_k, int thread_n, int num_threads, int n, int k) {
  if (thread_k == -1 || thread_n == -1 || num_threads == -1) return false;
  if (k % thread_k != 0 || n % thread_n != 0) return false;
  if (thread_k != 128 && thread_k != 64) return false;
  if (thread_n < 64 || thread_k < 64) return false;
  if (num_threads < 128) return false;
  return true;
}

// reference: determine_thread_config + the CALL_IF table (csrc/qqq_gemm.cu:847-865, :899-945,
// :1033-1036).  Returns 0 / ERR_PROB_SHAPE / ERR_KERN_SHAPE exactly when the reference does.
static int ref_shape_check(int m, int n, int k, int groupsize, int thread_k, int thread_n) {
  struct Cfg { int tk, tn, nt; };
  static const Cfg small_cfgs[4] = {{128, 128, 256}, {128, 64, 128}, {64, 256, 256}, {64, 128, 128}};
  static const Cfg large_cfgs[4] = {{64, 256, 256}, {128, 128, 256}, {64, 128, 128}, {128, 64, 128}};
  Cfg cfg = {-1, -1, -1};
  if (thread_k != -1 && thread_n != -1) {
    cfg = {thread_k, thread_n, 256};
  } else {
    const Cfg* list = (m <= 16) ? small_cfgs : large_cfgs;
    for (int i = 0; i < 4; ++i)
      if (ref_valid_config(list[i].tk, list[i].tn, list[i].nt, n, k)) {
        cfg = list[i];
        break;
      }
  }
  const int group_blocks = (groupsize == -1) ? -1 : groupsize / 16;
  if (!ref_valid_config(cfg.tk, cfg.tn, cfg.nt, n, k) || (group_blocks != -1 && group_blocks != 0 && k % group_blocks != 0) ||
      group_blocks == 0)
    return QQQ_ERR_PROB_SHAPE;
  if (m == 0 || n == 0 || k == 0) return QQQ_OK;
  const int nb = cfg.tn / 16, kb = cfg.tk / 16;
  const bool known = (nb == 8 && kb == 8 && cfg.nt == 256) || (nb == 16 && kb == 4 && cfg.nt == 256) ||
                     (nb == 8 && kb == 4 && cfg.nt == 128) || (nb == 4 && kb == 8 && cfg.nt == 128);
  if (!known || (group_blocks != -1 && group_blocks != 8)) return QQQ_ERR_KERN_SHAPE;
  return QQQ_OK;
}

struct LaunchArgs {
  const int8_t* A;
  const unsigned char* B;
  int32_t* C;
  _Float16* D;
  const float* s1;
  const float* s2;
  const _Float16* s3;
  int32_t* acc_out;
  int* tickets;
  const _Float16* bias;
  int M, N, K;
  hipStream_t stream;
};

template <int MT, bool GROUPED, int WAVES, int PF>
static hipError_t launch_stream_t(const LaunchArgs& a, int ksplit, int fused) {
  dim3 grid((a.N + 127) / 128, ksplit, (a.M + 16 * MT - 1) / (16 * MT));
  hipLaunchKernelGGL((qqq_stream_kernel<MT, GROUPED, WAVES, PF>), grid, dim3(WAVES * 64), 0, a.stream,
                     a.A, a.B, a.C, a.D, a.s1, a.s2, a.s3, a.acc_out, a.tickets, a.bias, a.M, a.N, a.K,
                     ksplit, fused);
  return hipGetLastError();
}

// prefetch depth PF (ring slots of 4 KiB weights per wave): deeper for the small-m bodies
template <bool GROUPED, int WAVES>
static hipError_t launch_stream_mt(const LaunchArgs& a, int mt, int pf, int ksplit, int fused) {
  if constexpr (WAVES == 16) {
    // 1024-thread blocks cap VGPRs at 128: only the MT=1 / PF=3 body fits (host never asks otherwise)
    return launch_stream_t<1, GROUPED, 16, 3>(a, ksplit, fused);
  } else {
    switch (mt) {
      case 1:
        if (pf >= 7) return launch_stream_t<1, GROUPED, WAVES, 7>(a, ksplit, fused);
        if (pf >= 5) return launch_stream_t<1, GROUPED, WAVES, 5>(a, ksplit, fused);
        return launch_stream_t<1, GROUPED, WAVES, 3>(a, ksplit, fused);
      case 2:
        if (pf >= 5) return launch_stream_t<2, GROUPED, WAVES, 5>(a, ksplit, fused);
        return launch_stream_t<2, GROUPED, WAVES, 3>(a, ksplit, fused);
      case 3:
        return launch_stream_t<3, GROUPED, WAVES, 2>(a, ksplit, fused);
      default:
        return launch_stream_t<4, GROUPED, WAVES, 2>(a, ksplit, fused);
    }
  }
}

static hipError_t launch_stream(const LaunchArgs& a, bool grouped, int mt, int waves, int pf, int ksplit,
                                int fused) {
  if (grouped) {
    if (waves == 4) return launch_stream_mt<true, 4>(a, mt, pf, ksplit, fused);
    if (waves == 16) return launch_stream_mt<true, 16>(a, mt, pf, ksplit, fused);
    return launch_stream_mt<true, 8>(a, mt, pf, ksplit, fused);
  }
  if (waves == 4) return launch_stream_mt<false, 4>(a, mt, pf, ksplit, fused);
  if (waves == 16) return launch_stream_mt<false, 16>(a, mt, pf, ksplit, fused);
  return launch_stream_mt<false, 8>(a, mt, pf, ksplit, fused);
}

template <int MT, bool GROUPED, int WAVES, int PF>
static hipError_t launch_column_t(const LaunchArgs& a, int ksplit) {
  dim3 grid(a.N / 32, ksplit, (a.M + 16 * MT - 1) / (16 * MT));
  hipLaunchKernelGGL((qqq_column_kernel<MT, GROUPED, WAVES, PF>), grid, dim3(WAVES * 64), 0, a.stream, a.A,
                     a.B, a.C, a.D, a.s1, a.s2, a.s3, a.acc_out, a.bias, a.M, a.N, a.K, ksplit);
  return hipGetLastError();
}

template <bool GROUPED>
static hipError_t launch_column_g(const LaunchArgs& a, int mt, int pf, int ksplit) {
  if (mt >= 2) {
    if (pf <= 4) return launch_column_t<2, GROUPED, 8, 4>(a, ksplit);
    return launch_column_t<2, GROUPED, 8, 8>(a, ksplit);
  }
  if (pf <= 2) return launch_column_t<1, GROUPED, 8, 2>(a, ksplit);
  if (pf <= 3) return launch_column_t<1, GROUPED, 8, 3>(a, ksplit);
  if (pf <= 4) return launch_column_t<1, GROUPED, 8, 4>(a, ksplit);
  if (pf <= 6) return launch_column_t<1, GROUPED, 8, 6>(a, ksplit);
  if (pf <= 8) return launch_column_t<1, GROUPED, 8, 8>(a, ksplit);
  return launch_column_t<1, GROUPED, 8, 12>(a, ksplit);
}

static hipError_t launch_column(const LaunchArgs& a, bool grouped, int mt, int pf, int ksplit) {
  return grouped ? launch_column_g<true>(a, mt, pf, ksplit) : launch_column_g<false>(a, mt, pf, ksplit);
}

template <int BM, int MTW, int JW, int NB, bool GROUPED, int NS>
static hipError_t launch_tiled_t(const LaunchArgs& a, int ksplit, int nslots, int pw) {
  constexpr int WAVES = (BM / (32 * MTW)) * (4 / JW) * (2 / NB);
  constexpr int NT = WAVES * 64;
  constexpr int STAGE = 8 * 2048 + BM * 128 + ((NS > 0 && GROUPED) ? WAVES * 512 : 0);
  constexpr int RING = ((NS == 5 || NS == 6) ? 3 : (NS > 0 ? NS : 2)) * STAGE;
  constexpr int LDS = RING > BM * 512 ? RING : BM * 512;  // the epilogue stages the fp16 tile (BM x 512 B)
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static bool attr_set[64] = {};  // per instantiation, per device
  auto kern = qqq_tiled_kernel<BM, MTW, JW, NB, GROUPED, NS>;
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur < 0 || cur >= 64 || !attr_set[cur]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    if (cur >= 0 && cur < 64) attr_set[cur] = true;
  }
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + 255) / 256;
  dim3 grid(tiles_m * tiles_n, ksplit, 1);
  hipLaunchKernelGGL(kern, grid, dim3(NT), LDS, a.stream, a.A, a.B, a.C, a.D, a.s1, a.s2, a.s3,
                     a.acc_out, a.bias, a.M, a.N, a.K, ksplit, tiles_m, tiles_n, a.tickets, nslots, pw);
  return hipGetLastError();
}

// stages: 0 = register-staged; 2..4 = LDS-DMA ring depth (clamped to what fits in 160 KiB of LDS)
template <bool GROUPED>
static hipError_t launch_tiled_bm(const LaunchArgs& a, int bm, int stages, int ksplit, int nslots, int pw) {
  switch (bm) {
    case 64:
      if (stages == 0) return launch_tiled_t<64, 1, 2, 2, GROUPED, 0>(a, ksplit, nslots, pw);
      if (stages == 2) return launch_tiled_t<64, 1, 2, 2, GROUPED, 2>(a, ksplit, nslots, pw);
      if (stages == 3) return launch_tiled_t<64, 1, 2, 2, GROUPED, 3>(a, ksplit, nslots, pw);
      return launch_tiled_t<64, 1, 2, 2, GROUPED, 4>(a, ksplit, nslots, pw);
    case 128:
      if (stages == 0) return launch_tiled_t<128, 2, 2, 2, GROUPED, 0>(a, ksplit, nslots, pw);
      if (stages == 2) return launch_tiled_t<128, 2, 2, 2, GROUPED, 2>(a, ksplit, nslots, pw);
      if (stages == 3) return launch_tiled_t<128, 2, 2, 2, GROUPED, 3>(a, ksplit, nslots, pw);
      return launch_tiled_t<128, 2, 2, 2, GROUPED, 4>(a, ksplit, nslots, pw);
    case 130:  // 128-row tile, 8 waves, each wave owns ONE (jt, b) column set over all 128 rows
      if (stages == 0) return launch_tiled_t<128, 4, 1, 1, GROUPED, 0>(a, ksplit, nslots, pw);
      if (stages == 2) return launch_tiled_t<128, 4, 1, 1, GROUPED, 2>(a, ksplit, nslots, pw);
      if (stages == 3) return launch_tiled_t<128, 4, 1, 1, GROUPED, 3>(a, ksplit, nslots, pw);
      if (stages == 5) return launch_tiled_t<128, 4, 1, 1, GROUPED, 5>(a, ksplit, nslots, pw);
      return launch_tiled_t<128, 4, 1, 1, GROUPED, 4>(a, ksplit, nslots, pw);
    case 131:  // 128-row tile, 8 waves as 2 (m) x 4 (jt): wave = 64 rows x one jt, both b
      if (stages == 0) return launch_tiled_t<128, 2, 1, 2, GROUPED, 0>(a, ksplit, nslots, pw);
      if (stages == 2) return launch_tiled_t<128, 2, 1, 2, GROUPED, 2>(a, ksplit, nslots, pw);
      if (stages == 3) return launch_tiled_t<128, 2, 1, 2, GROUPED, 3>(a, ksplit, nslots, pw);
      if (stages == 5) return launch_tiled_t<128, 2, 1, 2, GROUPED, 5>(a, ksplit, nslots, pw);
      return launch_tiled_t<128, 2, 1, 2, GROUPED, 4>(a, ksplit, nslots, pw);
    case 258:  // 256-row tile, 8 waves, each wave owns ONE (jt, b) column set over all 256 rows
      if (stages == 0) return launch_tiled_t<256, 8, 1, 1, GROUPED, 0>(a, ksplit, nslots, pw);
      if (stages == 2) return launch_tiled_t<256, 8, 1, 1, GROUPED, 2>(a, ksplit, nslots, pw);
      if (stages == 5) return launch_tiled_t<256, 8, 1, 1, GROUPED, 5>(a, ksplit, nslots, pw);
      return launch_tiled_t<256, 8, 1, 1, GROUPED, 3>(a, ksplit, nslots, pw);
    case 259:  // 256-row tile, 8 waves as 2 (m) x 4 (jt): wave = 128 rows x one jt, both b
      if (stages == 0) return launch_tiled_t<256, 4, 1, 2, GROUPED, 0>(a, ksplit, nslots, pw);
      if (stages == 2) return launch_tiled_t<256, 4, 1, 2, GROUPED, 2>(a, ksplit, nslots, pw);
      if (stages == 5) return launch_tiled_t<256, 4, 1, 2, GROUPED, 5>(a, ksplit, nslots, pw);
      return launch_tiled_t<256, 4, 1, 2, GROUPED, 3>(a, ksplit, nslots, pw);
    default:
      if (stages == 0) return launch_tiled_t<256, 2, 2, 2, GROUPED, 0>(a, ksplit, nslots, pw);
      if (stages == 2) return launch_tiled_t<256, 2, 2, 2, GROUPED, 2>(a, ksplit, nslots, pw);
      if (stages == 5) return launch_tiled_t<256, 2, 2, 2, GROUPED, 5>(a, ksplit, nslots, pw);
      if (stages == 6) return launch_tiled_t<256, 2, 2, 2, GROUPED, 6>(a, ksplit, nslots, pw);
      return launch_tiled_t<256, 2, 2, 2, GROUPED, 3>(a, ksplit, nslots, pw);
  }
}

static hipError_t launch_tiled(const LaunchArgs& a, bool grouped, int bm, int stages, int ksplit, int nslots, int pw) {
  return grouped ? launch_tiled_bm<true>(a, bm, stages, ksplit, nslots, pw)
                 : launch_tiled_bm<false>(a, bm, stages, ksplit, nslots, pw);
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// The dispatch decision of one call, as plain data (pure host logic: also exported as qqq_w4a8_plan so
// that it can be inspected and tested without a GPU).
struct Plan {
  int kernel;  // 1 stream, 2 tiled, 3 column
  int ksplit;
  int fused;   // stream: 1 / 3 in-launch, 2 separate reduce.  tiled: 1 in-launch slots, 2 slabs + reduce
  int mt, waves, pf;      // stream
  int bm, stages, nslots, pw; // tiled
};

static Plan make_plan(const int M, const int N, const int K, const bool grouped, const int max_par,
                      const bool have_C, const bool have_ws, qqq_tune_t t) {
  Plan pl;
  memset(&pl, 0, sizeof(pl));
  const long long cap_rows = (long long)(max_par > 0 ? max_par : 0) * 64;  // rows of C we may use
  const bool have_scratch = have_C && cap_rows > 0;
  const void* workspace = have_ws ? reinterpret_cast<const void*>(1) : nullptr;

  // ---- kernel choice (measured on MI355X, profiles/) ----
  // decode (m <= 16): the "column" kernel (32-column workgroups over all of K, no split-K, no reduce launch);
  // m <= 128: the HBM-bound "stream" kernel (128-column strips x K slices); above, LDS tiles.
  int kernel = t.kernel;
  const bool column_ok = (N % 64) == 0 && (K % 64) == 0;
  if (kernel == 0) {
    // measured (profiles/r01_tune_decode.txt): every column workgroup re-reads the m x K activations, so beyond
    // m = 8 it only wins while that stays small; very wide layers at m > 8 are better off with strips
    const bool column = column_ok && M <= 16 && N / 32 >= 64 &&
                        (M <= 8 || ((long long)M * K <= (grouped ? 524288 : 262144) && N / 32 <= 512));
    if (column) kernel = 3;
    else kernel = (M <= 128 || (K % 128) != 0) ? 1 : 2;
  }
  if (kernel == 3 && !column_ok) kernel = 1;
  if (kernel == 2 && (K % 128) != 0) kernel = 1;
  pl.kernel = kernel;
  int ksplit = 1;

  if (kernel == 3) {
    const int mt = (t.mt >= 1 && t.mt <= 2) ? t.mt : (M <= 16 ? 1 : 2);
    const int KS = K / 64;
    ksplit = t.ksplit > 0 ? t.ksplit : 1;  // a second launch costs more than idle CUs save (measured)
    ksplit = clampi(ksplit, 1, KS);
    if (!have_scratch) ksplit = 1;
    if (ksplit > 1 && (long long)ksplit * M > cap_rows) ksplit = (int)(cap_rows / M);
    if (ksplit < 1) ksplit = 1;
    pl.mt = mt;
    pl.waves = 8;
    pl.pf = t.pf > 0 ? t.pf : 3;
    pl.ksplit = ksplit;
    pl.fused = 2;
    return pl;
  }

  if (kernel == 1) {
    // rows are processed in m-blocks of 16*MT (grid.z); every m-block re-reads the weights, so this
    // kernel is meant for m <= 64 (one m-block) but stays correct for any m.
    const int mt = (t.mt >= 1 && t.mt <= 4) ? t.mt : clampi((M + 15) / 16, 1, 4);
    const int mblocks = (M + 16 * mt - 1) / (16 * mt);
    const int strips = (N + 127) / 128;
    const int KS = K / 64;
    int waves = t.waves ? t.waves : (mt == 1 ? 4 : 8);
    if (waves != 4 && waves != 8 && waves != 16) waves = 8;
    if (waves == 16 && mt > 1) waves = 8;  // 1024-thread blocks cap VGPRs at 128: only the MT=1 body fits
    ksplit = t.ksplit;
    if (ksplit <= 0) {
      // one workgroup per CU: (strips x m-blocks x K-slices) ~ 256, at least 2 steps per wave
      const long long base = (long long)strips * mblocks;
      ksplit = (int)((256 + base / 2) / base);
      ksplit = clampi(ksplit, 1, KS / (2 * waves) > 0 ? KS / (2 * waves) : 1);
    }
    ksplit = clampi(ksplit, 1, KS);
    if (!have_scratch) ksplit = 1;
    if (ksplit > 1 && (long long)ksplit * M > cap_rows) ksplit = (int)(cap_rows / M);
    if (ksplit < 1) ksplit = 1;
    int fused = t.fused;
    if (fused == 0) fused = 2;  // separate reduce launch measured ~2 us faster than the in-launch ticket path
    // tickets: one int per (m-block, strip); the reference guarantees n/128*max_par ints
    if ((fused == 1 || fused == 3) && (workspace == nullptr || (long long)mblocks * strips > (long long)(N / 128) * max_par))
      fused = 2;
    pl.mt = mt;
    pl.waves = waves;
    pl.pf = t.pf > 0 ? t.pf : (mt <= 2 ? 3 : 2);
    pl.ksplit = ksplit;
    pl.fused = fused;
    return pl;
  }

  // ---- tiled ----
  // Split-K comes in two forms.  In-launch (default when it fits): the K slices of a tile meet in tile-sized
  // int32 slots of C, tickets in `workspace`, the last arrival runs the epilogue -- needs one slot per tile
  // (<= max_par*64 rows of C) whatever ksplit is.  Slabs + separate reduce launch: ksplit full [m, n] slabs.
  const long long strips = (N + 255) / 256;
  const long long cap_ints = cap_rows * (long long)N;
  const long long cap_tickets = workspace ? (long long)(N / 128) * (max_par > 0 ? max_par : 0) : 0;
  auto slot_count = [&](int rows, int ks) -> int {
    if (!have_scratch || ks < 2 || t.fused == 2) return 0;
    const long long tl = (long long)((M + rows - 1) / rows) * strips;
    long long S = cap_ints / (tl * rows * 256);
    if (S > ks - 1) S = ks - 1;
    while (S > 0 && tl * (1 + S) > cap_tickets) --S;
    return (int)S;
  };
  int bm = t.bm;
  if (bm != 64 && bm != 128 && bm != 256 && bm != 258 && bm != 259 && bm != 130 && bm != 131) {
    // Pick the tile height (and its K split) by a small cost model in microseconds:
    //   rounds(tiles x ksplit over 256 CUs) x tile_time(rows, K / ksplit, rate(shape)) + split-K cost,
    // per-shape rates (TOPS at large m) measured on MI355X (profiles/).  Bigger tiles are more efficient per
    // MFMA but quantise worse over the CUs; split-K fills idle CUs at the price of int32 partial-sum traffic.
    // Wave shapes per mode: per-channel keeps 64x128 wave tiles (least LDS traffic); per-group uses the
    // column-owner shapes (258 / 130): every weight re-quantised once per workgroup.
    // {several workgroups co-resident per CU, a single one} -- small tiles lose efficiency when alone on a CU
    const double rate256 = grouped ? 1800.0 : 2300.0;
    const double rate128[2] = {grouped ? 1360.0 : 2050.0, grouped ? 1320.0 : 1650.0};
    const double rate64[2] = {grouped ? 800.0 : 1560.0, grouped ? 650.0 : 1170.0};
    int best_ks = 1;
    double best = 1e30;
    auto consider = [&](int rows, const double* rates, int code) {
      const long long tl = (long long)((M + rows - 1) / rows) * strips;
      const int ks_max = (tl < 192 && have_scratch) ? clampi((int)((256 + tl - 1) / tl), 1, (K / 128) / 4 > 0 ? (K / 128) / 4 : 1) : 1;
      for (int ks = 1; ks <= ks_max; ++ks) {
        const int S = slot_count(rows, ks);
        if (ks > 1 && S == 0 && (long long)ks * M > cap_rows) break;
        const double rate = (rows == 256) ? rates[0] : rates[(tl * ks <= 256) ? 1 : 0];
        const double tile_us = (double)rows * ((double)K / ks) * 131072.0 / (rate * 1e6) + 6.0;  // + prologue/epilogue
        double us = (double)((tl * ks + 255) / 256) * tile_us;
        if (ks > 1 && S > 0)  // every deposit is written once and read once (~4.2 TB/s chip-wide) + serial hops
          us += 3.0 + 2.0 * (ks - 1) * (double)tl * rows * 1024.0 / 4.2e6 + 3.0 * (double)((ks - 1 + S - 1) / S);
        else if (ks > 1) us += 5.0 + (double)ks * M * N * 8.0 / 3.0e6;  // slabs written + read at ~3 TB/s, + launch
        if (us < best) {
          best = us;
          bm = code;
          best_ks = ks;
        }
      }
    };
    consider(256, &rate256, grouped ? 258 : 256);
    consider(128, rate128, grouped ? 130 : 131);
    consider(64, rate64, 64);
    if (t.ksplit <= 0) t.ksplit = best_ks;
  }
  // glds: 2 = register-staged, 1 = LDS-DMA ring with `stages` buffers; auto: the DMA ring pays at the
  // 8-wave 256-row tile, register staging is faster for the 4-wave tiles (measured)
  int stages;
  if (t.glds == 2) stages = 0;
  else if (t.glds == 1) stages = ((t.stages >= 2 && t.stages <= 4) || (t.stages == 5 && bm != 64 && bm != 128) || (t.stages == 6 && bm == 256)) ? t.stages : (bm >= 256 ? 3 : 4);
  else stages = (bm == 256) ? 6 : (bm == 258) ? 2 : (bm == 130) ? 4 : 0;  // measured best per shape (profiles/r01_tune_sweep_*.txt)
  if (bm >= 256 && stages == 4) stages = 3;
  const int bm_rows = (bm >= 256) ? 256 : (bm >= 128 ? 128 : bm);
  const long long tiles = (long long)((M + bm_rows - 1) / bm_rows) * strips;
  ksplit = t.ksplit;
  if (ksplit <= 0) {
    ksplit = tiles >= 192 ? 1 : (int)((256 + tiles - 1) / tiles);
    ksplit = clampi(ksplit, 1, (K / 128) / 4 > 0 ? (K / 128) / 4 : 1);
  }
  ksplit = clampi(ksplit, 1, K / 128);
  if (!have_scratch) ksplit = 1;
  const int nslots = slot_count(bm_rows, ksplit);
  if (nslots == 0 && ksplit > 1 && (long long)ksplit * M > cap_rows) ksplit = (int)(cap_rows / M);
  if (ksplit < 1) ksplit = 1;
  pl.bm = bm;
  pl.stages = stages;
  // Tile order: an XCD runs 32 workgroups at a time = (32 / PW) m-tiles x PW weight strips.  Per 128-k block a
  // strip costs 16 KB of L2 fill and an m-tile rows/2 KB, so 4 x 8 is the cheapest split for 256- and 128-row
  // tiles (measured M=4096: L2 miss traffic 892 -> 714 MB per launch, profiles/r01_hbm_traffic.txt), 8 x 4 for 64 rows.
  pl.pw = (t.pw == 4 || t.pw == 8 || t.pw == 16 || t.pw == 32) ? t.pw : (bm == 64 ? 4 : 8);
  pl.ksplit = ksplit;
  pl.nslots = ksplit > 1 ? nslots : 0;
  pl.fused = pl.nslots > 0 ? 1 : 2;
  return pl;
}

extern "C" int qqq_w4a8_plan(int prob_m, int prob_n, int prob_k, int groupsize, int max_par, int have_scratch,
                             int have_workspace, const qqq_tune_t* tune, qqq_tune_t* plan_out) {
  g_err[0] = 0;
  if (!plan_out || prob_m <= 0 || prob_n <= 0 || prob_k <= 0) {
    snprintf(g_err, sizeof(g_err), "qqq_w4a8_plan: bad argument");
    return QQQ_ERR_ARG;
  }
  qqq_tune_t t;
  memset(&t, 0, sizeof(t));
  if (tune) t = *tune;
  const Plan pl = make_plan(prob_m, prob_n, prob_k, groupsize != -1, max_par, have_scratch != 0, have_workspace != 0, t);
  memset(plan_out, 0, sizeof(*plan_out));
  plan_out->kernel = pl.kernel;
  plan_out->ksplit = pl.ksplit;
  plan_out->fused = pl.fused;
  plan_out->waves = pl.waves;
  plan_out->pf = pl.pf;
  plan_out->mt = pl.mt;
  plan_out->bm = pl.bm;
  plan_out->stages = pl.stages;
  plan_out->glds = pl.kernel == 2 ? (pl.stages == 0 ? 2 : 1) : 0;
  plan_out->nslots = pl.nslots;
  plan_out->pw = pl.pw;
  return QQQ_OK;
}

extern "C" int qqq_w4a8_gemm_ex(const void* A, const void* B, void* C, void* D, const void* s1,
                                const void* s2, const void* s3, int prob_m, int prob_n, int prob_k,
                                void* workspace, int groupsize, int dev, void* stream, int thread_k,
                                int thread_n, int sms, int max_par, const qqq_tune_t* tune,
                                int32_t* acc_out, const void* bias) {
  (void)sms;
  g_err[0] = 0;
  const int rc = ref_shape_check(prob_m, prob_n, prob_k, groupsize, thread_k, thread_n);
  if (rc != QQQ_OK) return rc;
  if (prob_m == 0 || prob_n == 0 || prob_k == 0) return QQQ_OK;  // reference :1002-1003
  if (!A || !B || !D || !s1 || !s2 || (groupsize != -1 && !s3)) {
    snprintf(g_err, sizeof(g_err), "null pointer argument");
    return QQQ_ERR_ARG;
  }
  const bool grouped = groupsize != -1;
  qqq_tune_t t;
  memset(&t, 0, sizeof(t));
  if (tune) t = *tune;
  const int M = prob_m, N = prob_n, K = prob_k;
  const Plan pl = make_plan(M, N, K, grouped, max_par, C != nullptr, workspace != nullptr, t);

  LaunchArgs a;
  a.A = static_cast<const int8_t*>(A);
  a.B = static_cast<const unsigned char*>(B);
  a.C = static_cast<int32_t*>(C);
  a.D = static_cast<_Float16*>(D);
  a.s1 = static_cast<const float*>(s1);
  a.s2 = static_cast<const float*>(s2);
  a.s3 = static_cast<const _Float16*>(s3);
  a.acc_out = acc_out;
  a.bias = static_cast<const _Float16*>(bias);
  a.tickets = static_cast<int*>(workspace);
  a.M = M;
  a.N = N;
  a.K = K;
  a.stream = static_cast<hipStream_t>(stream);

  DeviceGuard guard(dev);
  hipError_t e = hipSuccess;
  bool reduce_launch;
  if (pl.kernel == 3) {
    e = launch_column(a, grouped, pl.mt, pl.pf, pl.ksplit);
    if (e != hipSuccess) return fail_hip(e, "qqq_column_kernel launch");
    reduce_launch = pl.ksplit > 1;
  } else if (pl.kernel == 1) {
    // kernel arg: 0 = slabs only (separate reduce launch), 1 = in-launch + release fence, 2 = in-launch + write-through
    e = launch_stream(a, grouped, pl.mt, pl.waves, pl.pf, pl.ksplit, pl.fused == 1 ? 1 : (pl.fused == 3 ? 2 : 0));
    if (e != hipSuccess) return fail_hip(e, "qqq_stream_kernel launch");
    reduce_launch = pl.ksplit > 1 && pl.fused != 1 && pl.fused != 3;
  } else {
    e = launch_tiled(a, grouped, pl.bm, pl.stages, pl.ksplit, pl.nslots, pl.pw);
    if (e != hipSuccess) return fail_hip(e, "qqq_tiled_kernel launch");
    reduce_launch = pl.ksplit > 1 && pl.nslots == 0;
  }
  if (reduce_launch) {
    const long long items = (long long)M * (N / 4);
    const int blocks = (int)((items + 63) / 64 > 8192 ? 8192 : (items + 63) / 64);
    hipLaunchKernelGGL(qqq_reduce_kernel, dim3(blocks), dim3(64), 0, a.stream, a.C, a.D, a.s1, a.s2,
                       a.acc_out, a.bias, M, N, pl.ksplit);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "qqq_reduce_kernel launch");
  }
  return QQQ_OK;
}

extern "C" int qqq_w4a8_gemm(const void* A, const void* B, void* C, void* D, const void* s1,
                             const void* s2, const void* s3, int prob_m, int prob_n, int prob_k,
                             void* workspace, int groupsize, int dev, void* stream, int thread_k,
                             int thread_n, int sms, int max_par) {
  return qqq_w4a8_gemm_ex(A, B, C, D, s1, s2, s3, prob_m, prob_n, prob_k, workspace, groupsize, dev,
                          stream, thread_k, thread_n, sms, max_par, nullptr, nullptr, nullptr);
}

extern "C" int qqq_dynamic_quant(const void* x, void* xq, void* s1, int m, int k, int dev,
                                 void* stream) {
  g_err[0] = 0;
  if (m == 0 || k == 0) return QQQ_OK;
  if (!x || !xq || !s1 || (k % 8) != 0) {
    snprintf(g_err, sizeof(g_err), "qqq_dynamic_quant: bad argument (k must be a multiple of 8)");
    return QQQ_ERR_ARG;
  }
  DeviceGuard guard(dev);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const _Float16* xp = static_cast<const _Float16*>(x);
  int8_t* qp = static_cast<int8_t*>(xq);
  float* sp = static_cast<float*>(s1);
  const int nvec = k / 8;
  const int vpt = (nvec + 255) / 256;
  if (vpt <= 2)
    hipLaunchKernelGGL(qqq_dynamic_quant_kernel<2>, dim3(m), dim3(256), 0, st, xp, qp, sp, k);
  else if (vpt <= 4)
    hipLaunchKernelGGL(qqq_dynamic_quant_kernel<4>, dim3(m), dim3(256), 0, st, xp, qp, sp, k);
  else if (vpt <= 8)
    hipLaunchKernelGGL(qqq_dynamic_quant_kernel<8>, dim3(m), dim3(256), 0, st, xp, qp, sp, k);
  else if (vpt <= 16)
    hipLaunchKernelGGL(qqq_dynamic_quant_kernel<16>, dim3(m), dim3(256), 0, st, xp, qp, sp, k);
  else if (vpt <= 32)
    hipLaunchKernelGGL(qqq_dynamic_quant_kernel<32>, dim3(m), dim3(256), 0, st, xp, qp, sp, k);
  else {
    snprintf(g_err, sizeof(g_err), "qqq_dynamic_quant: k=%d too large (max 65536)", k);
    return QQQ_ERR_ARG;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "qqq_dynamic_quant_kernel launch");
  return QQQ_OK;
}

extern "C" int qqq_add_bias(void* D, const void* bias, int m, int n, int dev, void* stream) {
  g_err[0] = 0;
  if (m == 0 || n == 0) return QQQ_OK;
  if (!D || !bias || (n % 8) != 0) {
    snprintf(g_err, sizeof(g_err), "qqq_add_bias: bad argument");
    return QQQ_ERR_ARG;
  }
  DeviceGuard guard(dev);
  const long long total = (long long)m * (n / 8);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(qqq_add_bias_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<_Float16*>(D), static_cast<const _Float16*>(bias), total, n / 8);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "qqq_add_bias_kernel launch");
  return QQQ_OK;
}

extern "C" int qqq_probe_mfma(int kind, const void* a, const void* b, void* out, int dev,
                              void* stream) {
  DeviceGuard guard(dev);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (kind == 16)
    hipLaunchKernelGGL(qqq_probe_mfma16_kernel, dim3(1), dim3(64), 0, st, static_cast<const v4i*>(a),
                       static_cast<const v4i*>(b), static_cast<v4i*>(out));
  else if (kind == 32)
    hipLaunchKernelGGL(qqq_probe_mfma32_kernel, dim3(1), dim3(64), 0, st, static_cast<const v4i*>(a),
                       static_cast<const v4i*>(b), static_cast<v16i*>(out));
  else
    return QQQ_ERR_ARG;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "probe launch");
  return QQQ_OK;
}

extern "C" int qqq_probe_fill(const void* src, size_t wg_stride, size_t bytes_per_wg, int nwg, int reps, int unroll,
                              void* sink, int dev, void* stream, float* ms_out) {
  DeviceGuard guard(dev);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail_hip(hipGetLastError(), "event");
  auto launch = [&]() {
    if (unroll >= 8)
      hipLaunchKernelGGL(qqq_probe_fill_kernel<8>, dim3(nwg), dim3(512), 0, st, static_cast<const v4u*>(src), wg_stride,
                         bytes_per_wg, reps, static_cast<unsigned*>(sink));
    else
      hipLaunchKernelGGL(qqq_probe_fill_kernel<2>, dim3(nwg), dim3(512), 0, st, static_cast<const v4u*>(src), wg_stride,
                         bytes_per_wg, reps, static_cast<unsigned*>(sink));
  };
  launch();  // warm-up
  (void)hipEventRecord(e0, st);
  launch();
  (void)hipEventRecord(e1, st);
  hipError_t e = hipStreamSynchronize(st);
  if (e == hipSuccess) e = hipEventElapsedTime(ms_out, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) return fail_hip(e, "qqq_probe_fill");
  return QQQ_OK;
}

extern "C" int qqq_probe_glds(const void* src, const void* perm, void* dst, int dev, void* stream) {
  DeviceGuard guard(dev);
  hipLaunchKernelGGL(qqq_probe_glds_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                     static_cast<const v4u*>(src), static_cast<const int*>(perm), static_cast<v4u*>(dst));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "probe launch");
  return QQQ_OK;
}

extern "C" int qqq_bench_gemm(const void* A, const void* const* Bs, int nB, void* C, void* D,
                              const void* s1, const void* s2, const void* s3, int prob_m, int prob_n,
                              int prob_k, void* workspace, int groupsize, int dev, void* stream,
                              int max_par, const qqq_tune_t* tune, int iters, float* ms_each) {
  g_err[0] = 0;
  if (iters <= 0 || nB <= 0 || !Bs || !ms_each) return QQQ_ERR_ARG;
  DeviceGuard guard(dev);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t* ev = new hipEvent_t[2 * iters];
  int made = 0, rc = QQQ_OK;
  for (; made < 2 * iters; ++made)
    if (hipEventCreate(&ev[made]) != hipSuccess) {
      rc = fail_hip(hipGetLastError(), "hipEventCreate");
      break;
    }
  if (rc == QQQ_OK) {
    for (int i = 0; i < iters && rc == QQQ_OK; ++i) {
      (void)hipEventRecord(ev[2 * i], st);
      rc = qqq_w4a8_gemm_ex(A, Bs[i % nB], C, D, s1, s2, s3, prob_m, prob_n, prob_k, workspace, groupsize,
                            dev, stream, -1, -1, -1, max_par, tune, nullptr, nullptr);
      (void)hipEventRecord(ev[2 * i + 1], st);
    }
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess && rc == QQQ_OK) rc = fail_hip(e, "hipStreamSynchronize");
    if (rc == QQQ_OK)
      for (int i = 0; i < iters; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) ms = -1.f;
        ms_each[i] = ms;
      }
  }
  for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  return rc;
}

extern "C" int qqq_amd_abi_version(void) { return QQQ_AMD_ABI_VERSION; }
extern "C" const char* qqq_amd_last_error(void) { return g_err; }
