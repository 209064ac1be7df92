// qqq_small.hip.h -- small kernels: split-K reduce, fused dynamic int8 quantisation, int4 packer / unpacker
// Part of the single translation unit qqq_w4a8.hip (see its header comment for the design).
#ifndef QQQ_AMD_QQQ_SMALL_HIP_H_
#define QQQ_AMD_QQQ_SMALL_HIP_H_

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
// fused per-token dynamic int8 quantisation (QuantLinear.dynamic_quant, qlinear_marlin.py:265-268)
// one workgroup per token row; the row is kept in registers between the two passes.
// ------------------------------------------------------------------------------------------
template <int VPT, int NT = 256>  // 16-byte vectors (8 halfs) per thread; covers K <= NT*8*VPT.  NT = 1024: few rows (decode) --
__global__ __launch_bounds__(NT) void qqq_dynamic_quant_kernel(const _Float16* __restrict__ x,  // the row's latency chain is 4x shorter
                                                               int8_t* __restrict__ xq,
                                                               float* __restrict__ s1, const int K) {
  __shared__ float wmax[NT / 64];
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
    const int idx = tid + i * NT;
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
  amax = wmax[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) amax = fmaxf(amax, wmax[w]);
  // torch on GPU lowers `.div(127.0)` to a multiply by the fp32 reciprocal; result kept in fp16
  const float scale = (float)(_Float16)__fmul_rn(amax, 1.0f / 127.0f);
  if (tid == 0) s1[row] = scale;
  // x / scale must be the correctly rounded fp32 quotient before rint() (torch semantics).  An IEEE division costs
  // ~12 VALU ops per element and made this kernel compute-bound; rint(x * (1/scale)) equals rint(x / scale) unless
  // the product lies within ~1e-4 of a half-integer (|q| <= 128 and the reciprocal-multiply is good to a few ulp),
  // so only elements for which some lane of the wave is that close take the exact division.
  const float rinv = (scale > 0.f) ? __frcp_rn(scale) : 0.f;  // all-zero row: reference gives NaN -> int8 (UB); we emit 0
  int2* qr = reinterpret_cast<int2*>(xq + (size_t)row * K);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int idx = tid + i * NT;
    if (idx < nvec) {
      float q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xe = (float)v[i][e];
        const float p = xe * rinv;
        q[e] = rintf(p);
        // the exact division only where some lane of the wave needs it for THIS element (~6 % of the wave-elements;
        // decided per 8-element vector it was ~40 % of the vectors, each paying 8 divisions)
        const bool near_tie = fabsf(p - q[e]) > 0.4995f;
        if (__builtin_amdgcn_ballot_w64(near_tie) != 0) {
          const float qd = (scale > 0.f) ? rintf(__fdiv_rn(xe, scale)) : 0.f;
          q[e] = near_tie ? qd : q[e];
        }
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


// ------------------------------------------------------------------------------------------
// int4 packer / unpacker for the Marlin/QQQ layout (replaces the python loops of
// QuantLinear.pack, qlinear_marlin.py:228-248).  Closed form (qlinear_marlin.py:147-176):
//   word B[kt][128*ng + 16*c + 4*kq + jt] holds k = 16*kt + 4*kq + r, n = 64*ng + 16*jt + 8*b + c;
//   nibble p of the word is (b, r) = (1-(p&1), p>>1) per-channel, ((p&3)>>1, 2*(p&1) + (p>>2)) per-group.
// One workgroup = one (k-tile, 64-column group): 16 x 64 codes <-> 128 words, both sides coalesced.
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ void qqq_nibble_coords(const int p, const bool grouped, int& b, int& r) {
  if (grouped) {
    b = (p & 3) >> 1;
    r = 2 * (p & 1) + (p >> 2);
  } else {
    b = 1 - (p & 1);
    r = p >> 1;
  }
}

__global__ __launch_bounds__(128) void qqq_pack_int4_kernel(const int8_t* __restrict__ codes,
                                                            unsigned* __restrict__ B, const int N,
                                                            const int grouped) {
  __shared__ __attribute__((aligned(8))) int8_t tile[16][64];
  const int tid = threadIdx.x, ng = blockIdx.x, kt = blockIdx.y;
  *reinterpret_cast<uint2*>(&tile[tid >> 3][(tid & 7) * 8]) = *reinterpret_cast<const uint2*>(
      codes + (size_t)(16 * kt + (tid >> 3)) * N + 64 * ng + (tid & 7) * 8);
  __syncthreads();
  const int c = tid >> 4, kq = (tid >> 2) & 3, jt = tid & 3;
  unsigned w = 0;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    int b, r;
    qqq_nibble_coords(p, grouped != 0, b, r);
    w |= ((unsigned)tile[4 * kq + r][16 * jt + 8 * b + c] & 0xFu) << (4 * p);
  }
  B[(size_t)kt * (2 * (size_t)N) + 128 * ng + tid] = w;
}

__global__ __launch_bounds__(128) void qqq_unpack_int4_kernel(const unsigned* __restrict__ B,
                                                              int8_t* __restrict__ codes, const int N,
                                                              const int grouped) {
  __shared__ __attribute__((aligned(8))) int8_t tile[16][64];
  const int tid = threadIdx.x, ng = blockIdx.x, kt = blockIdx.y;
  const unsigned w = B[(size_t)kt * (2 * (size_t)N) + 128 * ng + tid];
  const int c = tid >> 4, kq = (tid >> 2) & 3, jt = tid & 3;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    int b, r;
    qqq_nibble_coords(p, grouped != 0, b, r);
    const int u = (int)((w >> (4 * p)) & 0xFu);
    tile[4 * kq + r][16 * jt + 8 * b + c] = (int8_t)((grouped || u < 8) ? u : u - 16);
  }
  __syncthreads();
  *reinterpret_cast<uint2*>(codes + (size_t)(16 * kt + (tid >> 3)) * N + 64 * ng + (tid & 7) * 8) =
      *reinterpret_cast<const uint2*>(&tile[tid >> 3][(tid & 7) * 8]);
}


// ------------------------------------------------------------------------------------------
// Load-time expansion of per-group weights to int8 (round 6; SURVEY 8 f-3: "an optional CDNA-friendly re-layout at load time as an
// opt-in variant, default off").  A per-group weight is a pure function of (B, s_group): the re-quantisation the reference does inside
// its main loop (dequant_per_group, csrc/qqq_gemm.cu:167-210 -- here dequant_group4, the same bit tricks, wrap region included) is run
// ONCE per weight, and the int8 result is stored in the wide kernel's MFMA operand order, so that its loop needs neither the
// re-quantiser nor the quad transpose nor the group scales:
//   W8[step s = k / 64][ng = n / 64][q = 2 hf + b][lane = 16 h + 4 c + jt][16 bytes i]  =  w8(k = 64 s + 16 h + i, n = 64 ng + 16 jt + 8 b + 4 hf + c)
// One workgroup = one (step, 64-column group): 4 k-tiles x 128 packed words in, 4 KiB out; thread (h, c8 = 4 hf + c, jt) reads its four
// words kq = 0..3 (k = 16 (4 s + h) + 4 kq + r) and the scale pair of its two columns b = 0, 1 (stored order: 64 ng + 8 c8 + 2 jt + b).
// ------------------------------------------------------------------------------------------
// Per-channel layers (GROUPED = false) expand the same way: the operand the kernels form with `q & 0xF0F0F0F0` / `(q << 4) & 0xF0F0F0F0` (16 w4,
// csrc/qqq_gemm.cu:146-151, :540) stored as int8 -- no scales involved.
template <bool GROUPED>
__global__ __launch_bounds__(128) void qqq_expand_int8_kernel(const unsigned* __restrict__ B, const _Float16* __restrict__ s3,
                                                              v4u* __restrict__ W8, const int N) {
  const int tid = threadIdx.x, ng = blockIdx.x, s = blockIdx.y;
  const int h = tid >> 5, c8 = (tid >> 2) & 7, jt = tid & 3;
  const unsigned* src = B + (size_t)(4 * s + h) * (2 * (size_t)N) + 128 * ng + 16 * c8 + jt;
  h2 sb0 = {(_Float16)0, (_Float16)0}, sb1 = sb0;
  if constexpr (GROUPED) {
    const h2 sc = *reinterpret_cast<const h2*>(s3 + (size_t)(s >> 1) * N + 64 * ng + 8 * c8 + 2 * jt);
    sb0 = (h2){sc[0], sc[0]};
    sb1 = (h2){sc[1], sc[1]};
  }
  v4u w0, w1;
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) {
    int a, b;
    unpack_pair<GROUPED>(src[4 * kq], sb0, sb1, a, b);
    w0[kq] = (unsigned)a;
    w1[kq] = (unsigned)b;
  }
  const int hf = c8 >> 2, lane = 16 * h + 4 * (c8 & 3) + jt;
  v4u* dst = W8 + (((size_t)s * (N >> 6) + ng) * 4 + 2 * hf) * 64 + lane;
  dst[0] = w0;
  dst[64] = w1;
}


#endif  // QQQ_AMD_QQQ_SMALL_HIP_H_
