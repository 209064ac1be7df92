// qqq_common.hip.h -- vector types and the device helpers shared by every kernel (scale index maps, fused epilogue, int4 unpack, LDS-DMA)
// Part of the single translation unit qqq_w4a8.hip (see its header comment for the design).
#ifndef QQQ_AMD_QQQ_COMMON_HIP_H_
#define QQQ_AMD_QQQ_COMMON_HIP_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/qqq_amd.h"

// Weight loads: each byte of B is read once per launch, so its lines need not stay in the vector L1 / L2 behind the load.  Marking them non-temporal (`nt`)
// pays where a wave instruction takes whole 128-byte lines ONCE -- the column kernel (decode: 17.9 -> 16.3 us on the BASELINE layer, 10.4 -> 8.9 on
// 4096 x 11008) and the panel kernel's 64-token m-blocks (-5.5 % at 64 tokens) -- and costs where a lane comes back to its line with further 16-byte loads
// (stream kernel: +11 % at 16 tokens) or the loop is bound elsewhere (panel kernel at 128 tokens +1.7 %, wide kernel level).  profiles/r05_nt_weight_loads.txt;
// a bare HBM stream of 16-byte loads runs 14 % faster with `nt` (tools/l2_fill_bench.hip).  Bits: 1 column, 2 stream, 4 panel (MT <= 4), 8 wide, 16 panel (MT = 8);
// the default is what measured faster, other values are measurement builds.
#ifndef QQQ_W_NT
#define QQQ_W_NT 5
#endif
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define QQQ_NIB_MASK 0xF0F0F0F0u
// In-launch split-K: polls (each behind an s_sleep, ~0.1-0.3 us) the last arrival of a tile spends waiting for the deposits
// of slices that have, by construction, already arrived -- microseconds in practice.  Beyond the limit (seconds) the kernel
// traps: the launch fails loudly instead of folding whatever is in the slots.
#ifndef QQQ_SPIN_LIMIT
#define QQQ_SPIN_LIMIT (1 << 24)
#endif

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------

// In-launch split-K hand-off, the two ends of it (tiled / panel / wide kernels).  As shipped: deposits are written through
// (`sc0 sc1`), `s_waitcnt vmcnt(0)` + barrier, then a RELAXED agent-scope add on the completion count; the last arrival polls
// that count (relaxed), barrier, and reads the deposits with agent-scope `sc1` loads -- no fences: an agent-scope acquire is a
// `buffer_inv sc1` over the whole L2 (~3 us on the finisher's critical path, and it evicts every other workgroup's operand
// lines on the XCD).  The formal variants are RUNTIME switches (tune.fused bits 2 / 3 -> `hflags`), so that the same library
// runs both ways and the stress tests cover both (tests/test_gpu_parity.py); `-DQQQ_HANDOFF_ACQUIRE_FENCE` still forces bit 0.
__device__ __forceinline__ bool qqq_formal_acquire(const int hflags) {
#ifdef QQQ_HANDOFF_ACQUIRE_FENCE
  return true;
#else
  return (hflags & 1) != 0;
#endif
}
__device__ __forceinline__ void qqq_publish_add(int* counter, const int hflags) {  // a depositor's "my deposit is complete"
  if (hflags & 2) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

// the same with the two scale pairs already in registers (loaded ahead of a wait)
__device__ __forceinline__ h4 epilogue_vals4(const int v0, const int v1, const int v2, const int v3, const float a_s,
                                             const float2 sa, const float2 sb) {
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
// (The 0x6400 exponent pattern is handed to the compiler in a VGPR: v_and_or_b32 is VOP3, which on gfx9 takes no
//  literal and one scalar operand -- with both constants as literals hipcc emits v_and + v_or, 4 extra VALU per packed
//  word in a loop that is VALU-bound in the per-group mode.)
__device__ __forceinline__ unsigned qqq_fp16_1024x2() {
  unsigned m;
  asm("v_mov_b32 %0, 0x64006400" : "=v"(m));  // pure: hoisted out of the loops
  return m;
}
__device__ __forceinline__ unsigned dequant_group4(const unsigned q, const h2 s) {
  const unsigned magic = qqq_fp16_1024x2();
  const unsigned t0 = (q & 0x000f000fu) | magic;  // {1024+p0, 1024+p4}
  const unsigned t1 = (q & 0x00f000f0u) | magic;  // {1024+16*p1, 1024+16*p5}
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

// 4x4 transpose over (register e, lane q of a quad): y[e](lane q) = w[q](lane e) -- what turns the 16-byte pieces a
// lane loads from the packed layout (4 words jt = 0..3 of one (k-tile, chunk, kq)) into the MFMA lane's 4 words
// kq = 0..3 of its jt.  Two butterfly stages, each ONE VOP2 v_cndmask_b32 per register with a DPP quad_perm on the
// not-taken operand (8 VALU; written as plain C++ -- __builtin_amdgcn_mov_dpp + select -- hipcc emits v_mov_dpp +
// VOP3 v_cndmask, 16 VALU, because its lane-parity conditions live in SGPR pairs, not VCC).  The lane masks are
// constants: even lanes 0x5555.., lanes with bit 1 clear 0x3333...  The two s_mov in front of the first DPP read also
// cover the 2 wait states a DPP source needs behind a VALU write; the second stage reads z's written >= 2 slots earlier.
__device__ __forceinline__ void quad_transpose4(const v4u w, unsigned (&y)[4]) {
  unsigned z0, z1, z2, z3;
  asm("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\t"
      "v_cndmask_b32_dpp %4, %9, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"    // z0 = even ? w0 : w1'
      "v_cndmask_b32_dpp %6, %11, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  // z2 = even ? w2 : w3'
      "s_mov_b32 vcc_lo, 0xaaaaaaaa\n\ts_mov_b32 vcc_hi, 0xaaaaaaaa\n\t"
      "v_cndmask_b32_dpp %5, %8, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"    // z1 = odd ? w1 : w0'
      "v_cndmask_b32_dpp %7, %10, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  // z3 = odd ? w3 : w2'
      "s_mov_b32 vcc_lo, 0x33333333\n\ts_mov_b32 vcc_hi, 0x33333333\n\t"
      "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"    // y0 = lo ? z0 : z2''
      "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"    // y1 = lo ? z1 : z3''
      "s_mov_b32 vcc_lo, 0xcccccccc\n\ts_mov_b32 vcc_hi, 0xcccccccc\n\t"
      "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"    // y2 = hi ? z2 : z0''
      "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"          // y3 = hi ? z3 : z1''
      : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]), "=&v"(z0), "=&v"(z1), "=&v"(z2), "=&v"(z3)
      : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3])
      : "vcc");
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


// 16-byte load with agent-scope coherence (`sc1`: served from the coherence point, never from a stale line of this XCD's L2)
// through a raw buffer descriptor over a wave-uniform base -- the compiler tracks it like any load (an inline-asm
// `global_load_dwordx4 ... sc1` would not be), and it stands in for an agent-scope acquire fence, which on this part is a
// `buffer_inv sc1` over the whole L2 (~3 us, and it throws out the operand lines of every other workgroup on the XCD).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t agent_view(const void* base_uniform) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base_uniform), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ v4i load16_agent(__amdgpu_buffer_rsrc_t view, const unsigned byte_offset) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = __builtin_amdgcn_raw_buffer_load_b128(view, byte_offset, 0, /*sc1*/ 16);
  return (v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w};
}

#endif  // QQQ_AMD_QQQ_COMMON_HIP_H_
