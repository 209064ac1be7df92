// qqq_dev.hip -- test / tuning companion of the operator library (libqqq_amd_dev.so; include/qqq_amd_dev.h).
//
// Nothing in here is part of the product: hardware probes for tests/test_gpu_probe.py (MFMA lane maps, LDS-DMA
// destination semantics, the per-group re-quantiser on raw operands), a read-bandwidth probe, and the event-timed
// call loop bench.py / tools use.  The GEMM itself is NOT compiled into this library: qqq_dev_bench_gemm calls the
// operator library's qqq_w4a8_gemm_ex through a function pointer handed in by the caller, so what is timed is the
// shipped kernel.
#include "qqq_common.hip.h"
#include "../../include/qqq_amd_dev.h"

typedef unsigned int v2u __attribute__((ext_vector_type(2)));

static thread_local char g_dev_err[256] = "";

static int dev_fail(hipError_t e, const char* what) {
  snprintf(g_dev_err, sizeof(g_dev_err), "%s: %s", what, hipGetErrorString(e));
  return QQQ_ERR_HIP;
}

struct DevGuard {
  int prev = -1;
  bool changed = false;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev && dev >= 0) changed = (hipSetDevice(dev) == hipSuccess);
  }
  ~DevGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

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

// the kernels' per-group re-quantiser on raw operands: out[2i] / out[2i+1] = the b = 0 / b = 1 int8 quadruples
// unpack_pair<true> makes of word q[i] with scales {s0[i], s1[i]} (what every GEMM kernel feeds its MFMAs)
__global__ __launch_bounds__(256) void qqq_probe_dequant_kernel(const unsigned* __restrict__ q,
                                                                const unsigned short* __restrict__ s0,
                                                                const unsigned short* __restrict__ s1,
                                                                unsigned* __restrict__ out, const int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const _Float16 a = __builtin_bit_cast(_Float16, s0[i]), b = __builtin_bit_cast(_Float16, s1[i]);
  int w0, w1;
  unpack_pair<true>(q[i], (h2){a, a}, (h2){b, b}, w0, w1);
  out[2 * i] = (unsigned)w0;
  out[2 * i + 1] = (unsigned)w1;
}

// Sustained matrix-pipe rate under the power limit (tools/mfma_ceiling.py): 8 waves per workgroup, each running the
// tiled kernel's k-step shape -- 8 x v_mfma_i32_32x32x32_i8 on a 4 (weights) x 2 (tokens) fragment grid -- `iters`
// times on operands taken from `ops` (random bytes or zeros: the cycle count is the same, the clock is not).
//   MODE 0: operands stay in registers (two sets, alternated)         -> what the matrix pipe alone sustains
//   MODE 1: + the k-step's LDS fragment reads at the kernel's volume (4 KiB per wave and k-step), used as operands
//   MODE 2: + the per-channel unpack (shift / and / and per packed word) on the weight words read from LDS
template <int MODE>
__global__ __launch_bounds__(512) void qqq_probe_mfma_rate_kernel(const v4i* __restrict__ ops, const int iters,
                                                                 int* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) unsigned char img[16 * 4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const v4i* p = ops + ((size_t)blockIdx.x * 512 + tid) * 12;
  v4i a[2][4], b[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[s][i] = p[s * 6 + i];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[s][i] = p[s * 6 + 4 + i];
  }
  for (int i = tid; i < 16 * 256; i += 512) reinterpret_cast<v4i*>(img)[i] = ops[(size_t)blockIdx.x * 4096 + i];
  __syncthreads();
  v16i acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned char* st = img + (it & 15) * 4096;
    if constexpr (MODE == 0) {  // two k-steps per trip, one per operand set: no selects, no copies in the loop
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0][q & 3], b[0][q >> 2], acc[q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1][q & 3], b[1][q >> 2], acc[q], 0, 0, 0);
      ++it;
    } else if constexpr (MODE == 1) {
      const v4i x0 = *reinterpret_cast<const v4i*>(st + lane * 16), x1 = *reinterpret_cast<const v4i*>(st + 1024 + lane * 16);
      const v4i w0 = *reinterpret_cast<const v4i*>(st + 2048 + lane * 16), w1 = *reinterpret_cast<const v4i*>(st + 3072 + lane * 16);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = q & 3;
        acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(i == 0 ? w0 : i == 1 ? w1 : a[0][i], (q >> 2) ? x1 : x0, acc[q], 0, 0, 0);
      }
    } else {
      const v4i x0 = *reinterpret_cast<const v4i*>(st + lane * 16), x1 = *reinterpret_cast<const v4i*>(st + 1024 + lane * 16);
      v4i w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const v2u pk = *reinterpret_cast<const v2u*>(st + 2048 + i * 512 + lane * 8);
        w[i] = (v4i){(int)((pk[0] << 4) & 0xF0F0F0F0u), (int)(pk[0] & 0xF0F0F0F0u), (int)((pk[1] << 4) & 0xF0F0F0F0u),
                     (int)(pk[1] & 0xF0F0F0F0u)};
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[q & 3], (q >> 2) ? x1 : x0, acc[q], 0, 0, 0);
    }
  }
  int f = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) f ^= acc[q][r];
  if (f == 0x13579bdf) sink[0] = f;  // keeps the chains alive
}

// the same register-only loop on v_mfma_i32_16x16x64_i8 (the panel / stream / column kernels' instruction): 16 MFMAs per
// trip = the MACs of 8 x 32x32x32, operands 4 (weights) x 4 (tokens), two sets alternated
__global__ __launch_bounds__(512) void qqq_probe_mfma16_rate_kernel(const v4i* __restrict__ ops, const int iters, int* __restrict__ sink) {
  const int tid = threadIdx.x;
  const v4i* p = ops + ((size_t)blockIdx.x * 512 + tid) * 12;
  v4i a[2][4], b[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[s][i] = p[s * 6 + i];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[s][i] = p[s * 6 + 4 + i];
  }
  v4i acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = (v4i){0, 0, 0, 0};
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0][q & 3], b[0][(q >> 2) & 1], acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1][q & 3], b[1][(q >> 2) & 1], acc[q], 0, 0, 0);
  }
  int f = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) f ^= acc[q][r];
  if (f == 0x13579bdf) sink[0] = f;
}

extern "C" int qqq_dev_probe_mfma(int kind, const void* a, const void* b, void* out, int dev, void* stream) {
  DevGuard guard(dev);
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
  if (e != hipSuccess) return dev_fail(e, "probe launch");
  return QQQ_OK;
}

extern "C" int qqq_dev_probe_glds(const void* src, const void* perm, void* dst, int dev, void* stream) {
  DevGuard guard(dev);
  hipLaunchKernelGGL(qqq_probe_glds_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                     static_cast<const v4u*>(src), static_cast<const int*>(perm), static_cast<v4u*>(dst));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dev_fail(e, "probe launch");
  return QQQ_OK;
}

extern "C" int qqq_dev_probe_dequant(const void* q, const void* s0, const void* s1, void* out, int n, int dev,
                                     void* stream) {
  DevGuard guard(dev);
  if (n <= 0) return QQQ_OK;
  hipLaunchKernelGGL(qqq_probe_dequant_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned*>(q), static_cast<const unsigned short*>(s0),
                     static_cast<const unsigned short*>(s1), static_cast<unsigned*>(out), n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dev_fail(e, "probe launch");
  return QQQ_OK;
}

extern "C" int qqq_dev_probe_fill(const void* src, size_t wg_stride, size_t bytes_per_wg, int nwg, int reps, int unroll,
                                  void* sink, int dev, void* stream, float* ms_out) {
  DevGuard guard(dev);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return dev_fail(hipGetLastError(), "event");
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
  if (e != hipSuccess) return dev_fail(e, "qqq_dev_probe_fill");
  return QQQ_OK;
}

extern "C" int qqq_dev_probe_mfma_rate(int mode, const void* ops, int nwg, int iters, void* sink, int dev, void* stream,
                                       float* ms_out) {
  DevGuard guard(dev);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return dev_fail(hipGetLastError(), "event");
  auto launch = [&]() {
    const v4i* o = static_cast<const v4i*>(ops);
    int* sk = static_cast<int*>(sink);
    if (mode == 0) hipLaunchKernelGGL(qqq_probe_mfma_rate_kernel<0>, dim3(nwg), dim3(512), 0, st, o, iters, sk);
    else if (mode == 1) hipLaunchKernelGGL(qqq_probe_mfma_rate_kernel<1>, dim3(nwg), dim3(512), 0, st, o, iters, sk);
    else if (mode == 2) hipLaunchKernelGGL(qqq_probe_mfma_rate_kernel<2>, dim3(nwg), dim3(512), 0, st, o, iters, sk);
    else hipLaunchKernelGGL(qqq_probe_mfma16_rate_kernel, dim3(nwg), dim3(512), 0, st, o, iters, sk);  // mode 3
  };
  launch();  // warm-up (and clock settling)
  (void)hipEventRecord(e0, st);
  launch();
  (void)hipEventRecord(e1, st);
  hipError_t e = hipStreamSynchronize(st);
  if (e == hipSuccess) e = hipEventElapsedTime(ms_out, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) return dev_fail(e, "qqq_dev_probe_mfma_rate");
  return QQQ_OK;
}

// Where does a workgroup run?  Every workgroup records HW_REG_XCC_ID and HW_REG_HW_ID (CU / SH / SE ids) and then holds its CU for `hold_us`
// so that the grid spreads over every CU the stream may use.
__global__ __launch_bounds__(64) void qqq_probe_placement_kernel(unsigned* __restrict__ out, const int hold_us) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const unsigned long long t0 = wall_clock64();  // 100 MHz
  while (wall_clock64() - t0 < (unsigned long long)hold_us * 100ull) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
}

extern "C" int qqq_dev_probe_placement(const uint32_t* cu_mask, int mask_words, int nwg, int hold_us, void* out, int dev, void* stream) {
  g_dev_err[0] = 0;
  if (!out || nwg <= 0) return QQQ_ERR_ARG;
  DevGuard guard(dev);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipStream_t own = nullptr;
  if (cu_mask && mask_words > 0) {
    hipError_t e = hipExtStreamCreateWithCUMask(&own, (uint32_t)mask_words, cu_mask);
    if (e != hipSuccess) return dev_fail(e, "hipExtStreamCreateWithCUMask");
    st = own;
  }
  hipLaunchKernelGGL(qqq_probe_placement_kernel, dim3(nwg), dim3(64), 0, st, static_cast<unsigned*>(out), hold_us);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && own) e = hipStreamSynchronize(own);
  if (own) (void)hipStreamDestroy(own);
  if (e != hipSuccess) return dev_fail(e, "qqq_dev_probe_placement");
  return QQQ_OK;
}

static int bench_loop(qqq_gemm_ex_fn gemm_ex, qqq_gemm_ex2_fn gemm_ex2, const void* A, const void* const* Bs, const void* const* W8s, int nB, void* C, void* D,
                      const void* s1, const void* s2, const void* s3, int prob_m, int prob_n, int prob_k,
                      void* workspace, int groupsize, int dev, void* stream, int max_par,
                      const qqq_tune_t* tune, int iters, float* ms_each);

extern "C" int qqq_dev_bench_gemm(qqq_gemm_ex_fn gemm_ex, const void* A, const void* const* Bs, int nB, void* C, void* D,
                                  const void* s1, const void* s2, const void* s3, int prob_m, int prob_n, int prob_k,
                                  void* workspace, int groupsize, int dev, void* stream, int max_par,
                                  const qqq_tune_t* tune, int iters, float* ms_each) {
  return bench_loop(gemm_ex, nullptr, A, Bs, nullptr, nB, C, D, s1, s2, s3, prob_m, prob_n, prob_k, workspace, groupsize, dev, stream, max_par, tune, iters, ms_each);
}

extern "C" int qqq_dev_bench_gemm2(qqq_gemm_ex2_fn gemm_ex2, const void* A, const void* const* Bs, const void* const* W8s, int nB, void* C, void* D,
                                   const void* s1, const void* s2, const void* s3, int prob_m, int prob_n, int prob_k,
                                   void* workspace, int groupsize, int dev, void* stream, int max_par,
                                   const qqq_tune_t* tune, int iters, float* ms_each) {
  return bench_loop(nullptr, gemm_ex2, A, Bs, W8s, nB, C, D, s1, s2, s3, prob_m, prob_n, prob_k, workspace, groupsize, dev, stream, max_par, tune, iters, ms_each);
}

static int bench_loop(qqq_gemm_ex_fn gemm_ex, qqq_gemm_ex2_fn gemm_ex2, const void* A, const void* const* Bs, const void* const* W8s, int nB, void* C, void* D,
                      const void* s1, const void* s2, const void* s3, int prob_m, int prob_n, int prob_k,
                      void* workspace, int groupsize, int dev, void* stream, int max_par,
                      const qqq_tune_t* tune, int iters, float* ms_each) {
  g_dev_err[0] = 0;
  if ((!gemm_ex && !gemm_ex2) || iters <= 0 || nB <= 0 || !Bs || !ms_each) return QQQ_ERR_ARG;
  DevGuard guard(dev);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t* ev = new hipEvent_t[2 * iters];
  int made = 0, rc = QQQ_OK;
  for (; made < 2 * iters; ++made)
    if (hipEventCreate(&ev[made]) != hipSuccess) {
      rc = dev_fail(hipGetLastError(), "hipEventCreate");
      break;
    }
  if (rc == QQQ_OK) {
    for (int i = 0; i < iters && rc == QQQ_OK; ++i) {
      (void)hipEventRecord(ev[2 * i], st);
      if (gemm_ex2)
        rc = gemm_ex2(A, Bs[i % nB], C, D, s1, s2, s3, prob_m, prob_n, prob_k, workspace, groupsize, dev, stream, -1, -1, -1,
                      max_par, tune, nullptr, nullptr, W8s ? W8s[i % nB] : nullptr);
      else
        rc = gemm_ex(A, Bs[i % nB], C, D, s1, s2, s3, prob_m, prob_n, prob_k, workspace, groupsize, dev, stream, -1, -1, -1,
                     max_par, tune, nullptr, nullptr);
      (void)hipEventRecord(ev[2 * i + 1], st);
    }
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess && rc == QQQ_OK) rc = dev_fail(e, "hipStreamSynchronize");
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

extern "C" const char* qqq_dev_last_error(void) { return g_dev_err; }
