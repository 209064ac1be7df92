// Micro-benchmark: how many bytes per microsecond ONE compute unit of an MI355X can pull through its vector memory path when every CU
// does the same -- the ceiling behind the panel kernel's 128-token loop (24 KB per 128-k stage at 43-53 KB/us per CU) and the
// wide kernel's 256 x 128 tiles.  No MFMA, no LDS traffic, no barriers: workgroups of 512 threads (8 waves; one workgroup per CU, forced by
// its LDS allocation) issue 16-byte loads U deep, fold them with XOR and store one word at the end.
//
// Address patterns (what the GEMM kernels do, taken apart):
//   own-l2     every workgroup walks its own 96 KB window again and again            -> L2 hits, nothing shared
//   panel      all workgroups walk ONE activation panel (128 rows x K bytes, row-major, 128 B per row and stage: 8 lanes per row)
//              in four K slices, in lockstep                                           -> L2 hits, every line wanted by 64 CUs at once
//   panel-rot  the same, each workgroup starting at its own stage of its slice (wraps)  -> the same lines, not at the same time
//   stream     every workgroup streams its own contiguous range of a 2 GiB buffer      -> HBM
//   mix        per stage 8 KB of `stream` + 16 KB of `panel` (the 128-token loop's mix); mix-rot: with the rotated panel
// Columns `stream` / `panel`: the cache-policy bits on that part's loads (sc0 / sc1 / nt).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/l2_fill_bench.hip -o /tmp/l2fb && /tmp/l2fb
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

enum { OWN_L2 = 0, PANEL = 1, PANEL_ROT = 2, STREAM = 3, MIX = 4, MIX_ROT = 5 };
// cache-policy bits of a load (gfx942 / gfx950: sc0 / sc1 = scope, nt = non-temporal)
enum { PLAIN = 0, NT = 1, SC1 = 2, SC0SC1 = 3, SC0SC1NT = 4, SC0 = 5 };
static const char* const kModName[] = {"-", "nt", "sc1", "sc0sc1", "sc0sc1nt", "sc0"};

// every load is inline asm (the modifiers), so the waits are too: all loads of a trip, one s_waitcnt vmcnt(0), then the registers are
// handed back to the compiler through empty asm statements (asm volatile keeps its order)
template <int MOD>
__device__ __forceinline__ void ld16(u32x4& d, const void* p) {
  if (MOD == PLAIN) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(d) : "v"(p) : "memory");
  if (MOD == NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(d) : "v"(p) : "memory");
  if (MOD == SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(d) : "v"(p) : "memory");
  if (MOD == SC0SC1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(d) : "v"(p) : "memory");
  if (MOD == SC0SC1NT) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=&v"(d) : "v"(p) : "memory");
  if (MOD == SC0) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=&v"(d) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void landed(u32x4 (&v)[N]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int u = 0; u < N; ++u) asm volatile("" : "+v"(v[u]));
}

// one "stage" = 24 KB per workgroup in the mixes (16 KB panel + 8 KB stream), 16 KB in the pure patterns; U = loads in flight per thread
template <int PATTERN, int U, int THREADS, int SMOD, int PMOD>
__global__ void __launch_bounds__(THREADS) fill_kernel(const unsigned char* __restrict__ panel, const unsigned char* __restrict__ big, unsigned* out,
                                                        int stages, int total, int row_stride, long long stream_bytes_per_wg) {
  // stages: 128-byte stages of the whole panel (split into 4 slices; a slice is walked round and round); total: trips (pure patterns) / stages (panel patterns) to run
  extern __shared__ unsigned char lds_pad[];  // occupancy only
  const int t = threadIdx.x, wg = blockIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  constexpr int ROWS = 128;                       // rows of the panel
  constexpr int PER_STAGE = ROWS * 128 / 16;      // 16-byte pieces of a panel stage: 1024
  constexpr int PANEL_LOADS = PER_STAGE / THREADS;  // per thread and stage (2 at 512 threads)
  const int slice = wg & 3, nslice = 4;
  const int st0 = stages * slice / nslice, st1 = stages * (slice + 1) / nslice, nst = st1 - st0;
  const int rot = (PATTERN == PANEL_ROT || PATTERN == MIX_ROT) ? (int)(((unsigned)(wg >> 2) * 2654435761u >> 8) % (unsigned)nst) : 0;
  const unsigned char* own = big + (long long)wg * stream_bytes_per_wg + (long long)wg * 4352;  // (+ 17 x 256 B per workgroup: the windows must not all start on the same L2 channel)
  if (PATTERN == OWN_L2) {
    const int window = 96 * 1024 / 16;  // pieces
    int pos = t;
#pragma unroll 1
    for (int s = 0; s < total; ++s) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ld16<PMOD>(v[u], own + (long long)pos * 16);
        pos += THREADS;
        if (pos >= window) pos -= window;
      }
      landed(v);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  } else if (PATTERN == STREAM) {
    const unsigned char* p = own + (long long)t * 16;
#pragma unroll 1
    for (int s = 0; s < total; ++s) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) ld16<SMOD>(v[u], p + (long long)u * THREADS * 16);
      p += (long long)U * THREADS * 16;
      landed(v);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  } else {
    // panel patterns: U / PANEL_LOADS stages of the panel per trip (+ in the mixes 8 KB of the own stream per stage = 1 load per thread)
    constexpr int SPT = U / PANEL_LOADS;  // stages per trip
    constexpr bool MIXED = (PATTERN == MIX || PATTERN == MIX_ROT);
    constexpr int STHREADS = THREADS < 512 ? THREADS : 512;  // threads that carry the stream
    constexpr int WL = 8192 / (STHREADS * 16);               // stream loads per such thread and stage: 8 KB per stage
    const int row = t >> 3, col = (t & 7) * 16;
    const unsigned char* sp = own + (long long)t * 16;
    int s_at = rot;
#pragma unroll 1
    for (int s = 0; s + SPT <= total; s += SPT) {
      u32x4 v[U], w[SPT * WL];
#pragma unroll
      for (int q = 0; q < SPT * WL; ++q) w[q] = (u32x4){0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < SPT; ++q) {
        const long long koff = (long long)(st0 + s_at) * 128;
#pragma unroll
        for (int l = 0; l < PANEL_LOADS; ++l)
          ld16<PMOD>(v[q * PANEL_LOADS + l], panel + (long long)(row + l * (THREADS / 8)) * row_stride + koff + col);
        if (MIXED && (THREADS <= 512 || t < 512))  // (1024 threads: the stage's 8 KB of the stream are the first eight waves' loads)
#pragma unroll
          for (int e = 0; e < WL; ++e) ld16<SMOD>(w[q * WL + e], sp + ((long long)q * WL + e) * STHREADS * 16);
        if (++s_at == nst) s_at = 0;
      }
      if (MIXED) sp += (long long)SPT * WL * STHREADS * 16;
      landed(v);
      if (MIXED) landed(w);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
      if (MIXED) {
#pragma unroll
        for (int q = 0; q < SPT * WL; ++q) acc ^= w[q];
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[wg] = acc.x;  // (never true for the data used: keeps the loads)
  if (acc.x == 0x87654321u) lds_pad[t] = 1;  // (the allocation must stay)
}

// ---- scalar-path probes -------------------------------------------------------------------------------------------------------------
// The scalar unit has its own cache and its own road to L2.  A wave that only issues `s_load_dword` -- one per 128-byte line, results never
// read -- pulls lines into L2 without holding vector-L1 lines while they are under way.  (a) how many lines per microsecond can ONE wave per
// CU prefetch that way (spf_kernel);  (b) the mix again with the stream's lines prefetched D stages ahead by a ninth wave, the eight
// consumer waves loading them as L2 hits (mixpf_kernel; a barrier per trip keeps the prefetcher in step, D = 0: barrier only).
__device__ __forceinline__ const unsigned char* uniform_ptr(const unsigned char* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (const unsigned char*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void sprefetch_line(const unsigned char* base, unsigned off) {
  // the result lands in a fixed register nothing else uses (a returning load may overwrite it at any time: it must not be anybody's temporary)
  asm volatile("s_load_dword s96, %0, %1" : : "s"(base), "s"(off) : "s96", "memory");
}
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) spf_kernel(const unsigned char* __restrict__ big, unsigned* out, int lines, long long stream_bytes_per_wg) {
  extern __shared__ unsigned char lds_pad[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned char* p = uniform_ptr(big + (long long)blockIdx.x * stream_bytes_per_wg + (long long)wave * 128);
#pragma unroll 1
  for (int i = 0; i < lines; i += 16 * WAVES) {  // 16 lines per wave and trip, the waves interleaved line by line
#pragma unroll
    for (int j = 0; j < 16; ++j) sprefetch_line(p, (unsigned)(j * WAVES * 128));
    p += 16 * WAVES * 128;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lines < 0) lds_pad[threadIdx.x] = 1;
  if (lines == -12345) out[0] = 1;
}

template <int U, int SMOD, int D, int PW>
__global__ void __launch_bounds__(512 + 64 * (PW > 0 ? PW : 1)) mixpf_kernel(const unsigned char* __restrict__ panel, const unsigned char* __restrict__ big, unsigned* out,
                                                                             int stages, int total, int row_stride, long long stream_bytes_per_wg) {
  // PW scalar-only waves behind the eight consumer waves; pacing through a stage counter in LDS (consumer wave 0 posts the stage it has
  // finished, a prefetch wave does not run more than D stages ahead of it); D = 0: no prefetch (the extra wave only polls)
  extern __shared__ unsigned char lds_pad[];
  constexpr int THREADS = 512, SPT = U / 2, NPW = PW > 0 ? PW : 1;
  volatile int* posted = reinterpret_cast<volatile int*>(lds_pad);
  const int t = threadIdx.x, wg = blockIdx.x;
  const int slice = wg & 3;
  const int st0 = stages * slice / 4, nst = stages * (slice + 1) / 4 - st0;
  const unsigned char* own = big + (long long)wg * stream_bytes_per_wg + (long long)wg * 4352;
  u32x4 acc = {0, 0, 0, 0};
  if (t == 0) posted[0] = 0;
  __syncthreads();
  if (t >= THREADS) {
    const int pw = __builtin_amdgcn_readfirstlane((t - THREADS) >> 6);
    const unsigned char* p = uniform_ptr(own + (long long)pw * 128);
    if (D > 0) {
#pragma unroll 1
      for (int s = 0; s < total + 0; ++s) {  // stage s of the stream: lines pw, pw + NPW, ...
        while (s - __builtin_amdgcn_readfirstlane(posted[0]) > D) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int j = 0; j < 64 / NPW; ++j) sprefetch_line(p, (unsigned)(j * NPW * 128));
        p += 8192;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else {
    const int row = t >> 3, col = (t & 7) * 16;
    const unsigned char* sp = own + (long long)t * 16;
    int s_at = 0;
#pragma unroll 1
    for (int s = 0; s + SPT <= total; s += SPT) {
      u32x4 v[U], w[SPT];
#pragma unroll
      for (int q = 0; q < SPT; ++q) {
        const long long koff = (long long)(st0 + s_at) * 128;
        ld16<PLAIN>(v[2 * q], panel + (long long)row * row_stride + koff + col);
        ld16<PLAIN>(v[2 * q + 1], panel + (long long)(row + 64) * row_stride + koff + col);
        ld16<SMOD>(w[q], sp + (long long)q * THREADS * 16);
        if (++s_at == nst) s_at = 0;
      }
      sp += (long long)SPT * THREADS * 16;
      landed(v);
      landed(w);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
#pragma unroll
      for (int q = 0; q < SPT; ++q) acc ^= w[q];
      if (t == 0) posted[0] = s + SPT;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[wg] = acc.x;
  if (acc.x == 0x87654321u) lds_pad[64 + t] = 1;
}

template <typename F>
static double median_us(F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int rep = 0; rep < 12; ++rep) {
    CK(hipEventRecord(a, 0));
    launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float t;
    CK(hipEventElapsedTime(&t, a, b));
    if (rep >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  return ms[ms.size() / 2] * 1e3;
}

struct Result {
  double us, kb_per_us_cu, tb_s;
};

template <int PATTERN, int U, int THREADS, int SMOD, int PMOD>
static Result run(const unsigned char* panel, const unsigned char* big, unsigned* out, int wgs, int stages, int total, int row_stride, long long per_wg, int lds_bytes) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  auto kern = fill_kernel<PATTERN, U, THREADS, SMOD, PMOD>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  std::vector<float> ms;
  for (int rep = 0; rep < 12; ++rep) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(THREADS), lds_bytes, 0, panel, big, out, stages, total, row_stride, per_wg);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float t;
    CK(hipEventElapsedTime(&t, a, b));
    if (rep >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double us = ms[ms.size() / 2] * 1e3;
  // bytes per workgroup: stages (trips) run x bytes per stage (trip)
  double per_stage;
  if (PATTERN == OWN_L2 || PATTERN == STREAM)
    per_stage = (double)U * THREADS * 16;
  else
    per_stage = 16384.0 + ((PATTERN == MIX || PATTERN == MIX_ROT) ? 8192.0 : 0.0);
  const double bytes = per_stage * total;
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  return {us, bytes / 1024.0 / us, bytes * wgs / us * 1e-6};
}

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs; one workgroup per CU (dynamic LDS 100 KB), 16-byte loads, median of 10 launches\n", prop.name, cus);
  // the panel: 128 rows x 21760 B (the BASELINE layer's activation block at 128 tokens: 2.8 MB, fits every XCD's L2)
  const int row_stride = 21760, rows = 128;
  const int stride = row_stride, stages = row_stride / 128;  // 170 stages, 42-43 per slice, walked `passes` times
  const int passes = argc > 1 ? atoi(argv[1]) : 16;
  const int total = stages / 4 * passes / 8 * 8;
  unsigned char *panel, *big;
  unsigned* out;
  const long long per_wg = 8ll << 20;  // 8 MiB of private stream per workgroup: 2 GiB in all
  CK(hipMalloc(&panel, (size_t)rows * stride + 4096));
  CK(hipMalloc(&big, (size_t)per_wg * cus + 4096));
  CK(hipMalloc(&out, 4096 * 4));
  CK(hipMemset(panel, 1, (size_t)rows * stride));
  CK(hipMemset(big, 2, (size_t)per_wg * cus));
  CK(hipMemset(out, 0, 4096 * 4));
  CK(hipDeviceSynchronize());
  const int lds = 100 * 1024;
  printf("# panel %d rows x %d B (%.1f MB), %d stages of 128 B per row in 4 slices, each walked %d times (%d stages per workgroup)\n", rows, stride, rows * (double)stride / 1e6, stages, passes, total);
  printf("%-10s %3s %5s %9s %9s %9s %12s %8s\n", "pattern", "U", "thr", "stream", "panel", "us", "KB/us/CU", "TB/s");
#define ROW(name, P, U, T, SM, PM, TOTAL)                                                                                              \
  {                                                                                                                                   \
    Result r = run<P, U, T, SM, PM>(panel, big, out, cus, stages, TOTAL, stride, per_wg, lds);                                        \
    printf("%-10s %3d %5d %9s %9s %9.1f %12.1f %8.2f\n", name, U, T, kModName[SM], kModName[PM], r.us, r.kb_per_us_cu, r.tb_s);       \
    fflush(stdout);                                                                                                                   \
  }
  // own window in L2 (16 MB per workgroup)
  ROW("own-l2", OWN_L2, 8, 512, PLAIN, PLAIN, 256)
  ROW("own-l2", OWN_L2, 16, 512, PLAIN, PLAIN, 128)
  ROW("own-l2", OWN_L2, 8, 256, PLAIN, PLAIN, 512)
  ROW("own-l2", OWN_L2, 8, 1024, PLAIN, PLAIN, 128)
  ROW("own-l2", OWN_L2, 8, 512, PLAIN, SC1, 256)
  ROW("own-l2", OWN_L2, 8, 512, PLAIN, SC0SC1, 256)
  // own stream from HBM: 4 MB per workgroup, 1 GB in all
  ROW("stream", STREAM, 8, 512, PLAIN, PLAIN, 64)
  ROW("stream", STREAM, 8, 512, NT, PLAIN, 64)
  ROW("stream", STREAM, 8, 512, SC1, PLAIN, 64)
  ROW("stream", STREAM, 8, 512, SC0SC1, PLAIN, 64)
  ROW("stream", STREAM, 8, 512, SC0SC1NT, PLAIN, 64)
  ROW("stream", STREAM, 4, 1024, PLAIN, PLAIN, 64)
  // the shared activation panel
  ROW("panel", PANEL, 8, 512, PLAIN, PLAIN, total)
  ROW("panel", PANEL, 8, 512, PLAIN, NT, total)
  ROW("panel", PANEL, 8, 512, PLAIN, SC1, total)
  ROW("panel", PANEL, 8, 512, PLAIN, SC0SC1, total)
  ROW("panel", PANEL, 8, 256, PLAIN, PLAIN, total)
  ROW("panel", PANEL, 8, 1024, PLAIN, PLAIN, total)
  ROW("panel-rot", PANEL_ROT, 8, 512, PLAIN, PLAIN, total)
  // the 128-token loop's mix: 16 KB of the panel + 8 KB of the own stream per stage
  ROW("mix", MIX, 8, 512, PLAIN, PLAIN, total)
  ROW("mix", MIX, 8, 512, NT, PLAIN, total)
  ROW("mix", MIX, 8, 512, SC1, PLAIN, total)
  ROW("mix", MIX, 8, 512, SC0SC1, PLAIN, total)
  ROW("mix", MIX, 8, 512, SC0SC1NT, PLAIN, total)
  ROW("mix", MIX, 8, 512, SC0, PLAIN, total)
  ROW("mix", MIX, 8, 512, NT, NT, total)
  ROW("mix", MIX, 8, 512, SC0SC1, SC0SC1, total)
  ROW("mix", MIX, 8, 512, SC0SC1, SC1, total)
  ROW("mix", MIX, 16, 512, PLAIN, PLAIN, total)
  ROW("mix", MIX, 8, 256, PLAIN, PLAIN, total)
  ROW("mix-rot", MIX_ROT, 8, 512, PLAIN, PLAIN, total)
  // the own stream on a part of the chip only: what ONE CU pulls from HBM when the HBM is not the bound
  for (int part : {16, 32, 64, 128}) {
    auto kern = fill_kernel<STREAM, 8, 512, NT, PLAIN>;
    const double us = median_us([&] { hipLaunchKernelGGL(kern, dim3(part), dim3(512), lds, 0, panel, big, out, stages, 64, stride, per_wg); });
    printf("stream on %3d CUs (nt)                      %9.1f %12.1f %8.2f\n", part, us, 64 * 64.0 / us, 64 * 65536.0 * part / us * 1e-6);
  }
  // scalar prefetch: lines of 128 B per microsecond and CU, 1 / 2 / 4 waves per CU; 4 MB of the own stream per workgroup
  {
    const int lines = 4 * 1024 * 1024 / 128;
    CK(hipFuncSetAttribute((const void*)spf_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)spf_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)spf_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    double us = median_us([&] { hipLaunchKernelGGL(spf_kernel<1>, dim3(cus), dim3(64), lds, 0, big, out, lines, per_wg); });
    printf("scalar prefetch, 1 wave per CU: %9.1f us  %8.1f lines/us/CU = %6.1f KB/us/CU of L2 fill, %6.2f TB/s\n", us, lines / us, lines / us / 8, lines * 128.0 * cus / us * 1e-6);
    us = median_us([&] { hipLaunchKernelGGL(spf_kernel<2>, dim3(cus), dim3(128), lds, 0, big, out, lines, per_wg); });
    printf("scalar prefetch, 2 waves per CU: %8.1f us  %8.1f lines/us/CU = %6.1f KB/us/CU of L2 fill, %6.2f TB/s\n", us, lines / us, lines / us / 8, lines * 128.0 * cus / us * 1e-6);
    us = median_us([&] { hipLaunchKernelGGL(spf_kernel<4>, dim3(cus), dim3(256), lds, 0, big, out, lines, per_wg); });
    printf("scalar prefetch, 4 waves per CU: %8.1f us  %8.1f lines/us/CU = %6.1f KB/us/CU of L2 fill, %6.2f TB/s\n", us, lines / us, lines / us / 8, lines * 128.0 * cus / us * 1e-6);
    us = median_us([&] { hipLaunchKernelGGL(spf_kernel<1>, dim3(32), dim3(64), lds, 0, big, out, lines, per_wg); });
    printf("scalar prefetch, 1 wave per CU on 32 CUs: %8.1f us  %8.1f lines/us/CU = %6.1f KB/us/CU\n", us, lines / us, lines / us / 8);
  }
  // the mix with the stream prefetched into L2 by a ninth, scalar-only wave D stages ahead
#define MIXPF(SM, DD, PW)                                                                                                              \
  {                                                                                                                                   \
    auto kern = mixpf_kernel<8, SM, DD, PW>;                                                                                          \
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                      \
    const double us = median_us([&] { hipLaunchKernelGGL(kern, dim3(cus), dim3(512 + 64 * (PW > 0 ? PW : 1)), lds, 0, panel, big, out, stages, total, stride, per_wg); }); \
    printf("mix + %d scalar prefetch wave(s) %2d stages ahead (stream loads %-3s) %9.1f us %8.1f KB/us/CU  %6.3f us per stage\n", PW, DD, kModName[SM], us, \
           24.0 * total / us, us / total);                                                                                            \
    fflush(stdout);                                                                                                                   \
  }
  MIXPF(PLAIN, 0, 0)
  MIXPF(NT, 0, 0)
  MIXPF(PLAIN, 4, 1)
  MIXPF(PLAIN, 4, 2)
  MIXPF(PLAIN, 4, 4)
  MIXPF(PLAIN, 8, 4)
  MIXPF(PLAIN, 16, 4)
  MIXPF(NT, 8, 4)
  MIXPF(PLAIN, 8, 8)
  MIXPF(PLAIN, 16, 8)
  return 0;
}
