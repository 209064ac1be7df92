// qqq_small.hip.h -- small kernels: split-K reduce, fused dynamic int8 quantisation, bias add, hardware probes
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


#endif  // QQQ_AMD_QQQ_SMALL_HIP_H_
