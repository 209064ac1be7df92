// Micro-benchmark (round 6): is the ~36 KB of vector loads a CU holds in flight (tools/l2_fill_bench.hip) a limit in BYTES of return data or in LINES
// (requests)?  If bytes, a "touch" -- one byte per 128-byte line, 64 lines per wave instruction, issued D stages ahead of the 16-byte loads that use the data -- would turn
// the HBM stream of the small-m kernels into L2 hits for 1/16 of the in-flight budget.  One workgroup of 512 threads per CU (forced by its LDS allocation), every CU the same:
//   stream        8 KB per stage of the workgroup's own contiguous range (HBM), U stages per trip
//   stream+touch  the same; wave w of 8 also touches (global_load_ubyte, lane l -> line l) the 64 lines of stage s + D + w of the trip, not waited for until the next trip
//   mix / mix+touch   per stage additionally 16 KB of ONE shared 128-row panel (L2 hits): the 128-token loop's mix
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/touch_prefetch_bench.hip -o /tmp/tpb && /tmp/tpb
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void ld16(u32x4& d, const void* p) {
  if (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(d) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(d) : "v"(p) : "memory");
}

// TOUCH: 0 none, 1 global_load_ubyte (a return slot of one dword per lane), 2 global_load_dwordx4 of the line's first 16 bytes (control: same lines, full-size return)
template <int U, int TOUCH, bool MIX, bool NT>
__global__ void __launch_bounds__(512) touch_kernel(const unsigned char* __restrict__ panel, const unsigned char* __restrict__ big, unsigned* out, int trips,
                                                     int dist, int panel_stages, int row_stride, long long bytes_per_wg) {
  extern __shared__ unsigned char lds_pad[];  // occupancy only
  const int t = threadIdx.x, wg = blockIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned char* own = big + (long long)wg * bytes_per_wg + (long long)wg * 4352;
  const unsigned char* p = own + (long long)t * 16;
  const unsigned char* tp = own + (long long)lane * 128 + (long long)(dist + wave) * 8192;  // this wave's touch of the trip: stage dist + wave
  const int row = t >> 3, col = (t & 7) * 16;
  const int slice = wg & 3, st0 = panel_stages * slice / 4, nst = panel_stages / 4;
  int s_at = 0;
  u32x4 acc = {0, 0, 0, 0};
  unsigned tb = 0;
  u32x4 tq = {0, 0, 0, 0};
#pragma unroll 1
  for (int s = 0; s < trips; ++s) {
    u32x4 v[U], w[MIX ? 2 * U : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ld16<NT>(v[u], p + (long long)u * 8192);
      if (MIX) {
        const long long koff = (long long)(st0 + s_at) * 128;
        ld16<false>(w[2 * u], panel + (long long)row * row_stride + koff + col);
        ld16<false>(w[2 * u + 1], panel + (long long)(row + 64) * row_stride + koff + col);
        if (++s_at == nst) s_at = 0;
      }
    }
    // the touch is the trip's youngest load: the wait below leaves it in flight (vmcnt(1)), the next trip's wait covers it
    if (TOUCH == 1 && wave < U) asm volatile("global_load_ubyte %0, %1, off" : "=&v"(tb) : "v"(tp) : "memory");
    if (TOUCH == 2 && wave < U) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(tq) : "v"(tp) : "memory");
    p += (long long)U * 8192;
    tp += (long long)U * 8192;
    if (TOUCH && wave < U) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // (wave-uniform)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; ++u) {
      asm volatile("" : "+v"(v[u]));
      acc ^= v[u];
    }
    if (MIX) {
#pragma unroll
      for (int u = 0; u < 2 * U; ++u) {
        asm volatile("" : "+v"(w[u]));
        acc ^= w[u];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(tb), "+v"(tq)::"memory");
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w ^ tb ^ tq.x) == 0x12345678u) out[wg] = acc.x;
  if (acc.x == 0x87654321u) lds_pad[t] = 1;
}

template <int U, int TOUCH, bool MIX, bool NT>
static void run(const char* name, const unsigned char* panel, const unsigned char* big, unsigned* out, int trips, int dist, long long bytes_per_wg, int ncu) {
  auto k = touch_kernel<U, TOUCH, MIX, NT>;
  const int lds = 100 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(ncu), dim3(512), lds, 0, panel, big, out, trips, dist, 168, 21760, bytes_per_wg);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float m;
    CK(hipEventElapsedTime(&m, e0, e1));
    ms.push_back(m);
  }
  std::sort(ms.begin(), ms.end());
  const double us = ms[2] * 1e3, stages = (double)trips * U;
  const double kb = stages * (MIX ? 24.0 : 8.0);
  printf("%-34s U=%d D=%2d %s  %8.1f us  %6.1f KB/us/CU  stream %5.2f TB/s  %.3f us per stage\n", name, U, dist, NT ? "nt" : "- ", us, kb / us,
         stages * 8192.0 * ncu / us / 1e6, us / stages);
}

int main() {
  int ncu = 256;
  const long long bytes_per_wg = 7ll << 20;  // 7 MiB per workgroup: 1.75 GiB in all, beyond every cache
  unsigned char *big, *panel;
  unsigned* out;
  CK(hipMalloc(&big, bytes_per_wg * ncu + (8 << 20)));
  CK(hipMemset(big, 1, bytes_per_wg * ncu + (8 << 20)));
  CK(hipMalloc(&panel, 128 * 21760 + 4096));
  CK(hipMemset(panel, 2, 128 * 21760 + 4096));
  CK(hipMalloc(&out, 4096));
  const int stages = 832;  // 6.5 MiB of the 7
#define ROW(U, D)                                                                                                  \
  run<U, 0, false, false>("stream", panel, big, out, stages / U, 0, bytes_per_wg, ncu);                          \
  run<U, 0, false, true>("stream", panel, big, out, stages / U, 0, bytes_per_wg, ncu);                           \
  run<U, 1, false, false>("stream + byte touch", panel, big, out, stages / U, D, bytes_per_wg, ncu);             \
  run<U, 2, false, false>("stream + 16-byte touch (control)", panel, big, out, stages / U, D, bytes_per_wg, ncu); \
  run<U, 0, true, false>("mix", panel, big, out, stages / U, 0, bytes_per_wg, ncu);                               \
  run<U, 1, true, false>("mix + byte touch", panel, big, out, stages / U, D, bytes_per_wg, ncu);                 \
  run<U, 2, true, false>("mix + 16-byte touch (control)", panel, big, out, stages / U, D, bytes_per_wg, ncu);
  ROW(4, 8)
  ROW(4, 16)
  ROW(8, 16)
  ROW(8, 32)
  ROW(2, 8)
  return 0;
}
