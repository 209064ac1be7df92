// Micro-benchmark (round 6): what does a wave that is ALONE on its SIMD (the wide kernel: 4 waves per workgroup, 512 registers each) pay per instruction it issues next
// to its MFMAs?  Shader-clock cycles per v_mfma_i32_16x16x64_i8 (s_memtime around the loop, so the number does not depend on the clock the chip settles at) for a loop of 64
// independent in-place MFMAs per trip (inline asm, accumulators tied, as in qqq_wide.hip.h) plus, per pattern:
//   0  nothing else                          1 / 2 / 3  that many independent VALU instructions behind every MFMA
//   4  one ds_read_b128 per 4 MFMAs (the kernel's fragment re-reads)        5  pattern 4 + 3 VALU per 4 MFMAs (the per-channel unpack's share)
//   6  one LDS-DMA (buffer_load_dwordx4 ... lds) + its M0 write per 16 MFMAs (the activation staging)
//   7  one buffer_load_dwordx4 into registers per 32 MFMAs (the weight-ring refill)
//   8  4 + 5 + 6 + 7 together: the wide kernel's per-channel step without its waits
// once with ONE wave per SIMD (256 threads, 64 accumulator quads per wave) and once with TWO (512 threads, 32 quads per wave: the MFMAs of a trip halve, the extras per MFMA stay).
// One workgroup per CU (LDS allocation), operands zero (the clock stays high; cycles are what is reported).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_issue_bench.hip -o /tmp/mib && /tmp/mib
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

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int... S, class F>
__device__ __forceinline__ void static_for_(std::integer_sequence<int, S...>, F&& f) {
  (f(std::integral_constant<int, S>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <int WPS, int PAT>
__global__ __launch_bounds__(256 * WPS) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void issue_kernel(const unsigned char* __restrict__ g, unsigned long long* __restrict__ out,
                                                                                                        const int trips, const int fill) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NACC = 64 / WPS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  v4i acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (v4i){0, 0, 0, 0};
  v4i a[4], x[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = (v4i){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (v4i){0, 0, 0, 0};
  for (int i = tid; i < 32768 / 16; i += 256 * WPS) reinterpret_cast<v4i*>(smem)[i] = (v4i){0, 0, 0, 0};
  __syncthreads();
  unsigned v0 = tid, v1 = tid * 3, v2 = tid * 5, v3 = tid * 7, v4 = tid * 11, v5 = tid * 13;
  const unsigned long long ga = (unsigned long long)(g + (size_t)blockIdx.x * 65536);
  const v4u desc = {(unsigned)__builtin_amdgcn_readfirstlane((int)ga), (unsigned)__builtin_amdgcn_readfirstlane((int)(ga >> 32)), 0xffffffffu, 0x00020000u};
  const unsigned voff = (unsigned)(tid * 16);
  const unsigned lds_dst = (unsigned)(32768 + wn * 1024);
  const unsigned xrd = (unsigned)(lane * 16);
  v4u ring[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane(tid);
  const unsigned long long km = 0x5555555555555555ull;
  if (fill) {  // operands from memory (random bytes): the clock the chip settles at depends on the VALUES, the cycle counts do not
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = reinterpret_cast<const v4i*>(g)[(size_t)((blockIdx.x & 7) * 20 + i) * 512 + tid];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = reinterpret_cast<const v4i*>(g)[(size_t)((blockIdx.x & 7) * 20 + 4 + i) * 512 + tid];
  }
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(x[i]));  // (opaque: hipcc must not re-materialise the zeros inside the loop)
  asm volatile("" : "+v"(ring[0]), "+v"(ring[1]));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < trips; ++it) {
    static_for<NACC>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a[k % 4]), "v"(x[(k / 4) % 16]));
      if constexpr (PAT >= 1 && PAT <= 3) asm volatile("v_and_b32 %0, 0xf0f0f0f0, %0" : "+v"(v0));
      if constexpr (PAT >= 2 && PAT <= 3) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(v1));
      if constexpr (PAT == 3) asm volatile("v_and_b32 %0, 0xf0f0f0f0, %0" : "+v"(v2));
      if constexpr (PAT == 4 || PAT == 5 || PAT == 8) {
        if constexpr (k % 4 == 3) x[(k / 4) % 16] = *reinterpret_cast<const v4i*>(smem + xrd + ((k / 4) % 16) * 1024 + ((k / 4) % 2) * 16384);
      }
      if constexpr (PAT == 5 || PAT == 8) {
        if constexpr (k % 4 == 0) asm volatile("v_and_b32 %0, 0xf0f0f0f0, %0" : "+v"(v3));
        if constexpr (k % 4 == 1) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(v4));
        if constexpr (k % 4 == 2) asm volatile("v_and_b32 %0, 0xf0f0f0f0, %0" : "+v"(v5));
      }
      if constexpr (PAT == 6 || PAT == 8) {
        if constexpr (k % 16 == 8) asm volatile("s_add_u32 m0, %0, %1" : : "s"(lds_dst), "n"(4096 * ((k / 16) % 4)) : "scc");
        if constexpr (k % 16 == 9) asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(desc) : "memory");
      }
      // 9: three VALU behind every OTHER MFMA (1.5 per MFMA on average: does an empty slot give back what a full one lost?)
      if constexpr (PAT == 9 && k % 2 == 0) asm volatile("v_and_b32 %0, 0xf0f0f0f0, %0\n\tv_lshlrev_b32 %1, 4, %1\n\tv_and_b32 %2, 0xf0f0f0f0, %2" : "+v"(v0), "+v"(v1), "+v"(v2));
      // 10 / 11: two / three SALU instructions behind every MFMA;  12: one VALU + two SALU
      if constexpr (PAT == 10 || PAT == 11) asm volatile("s_nop 0\n\ts_add_u32 %0, %0, 1" : "+s"(sa) : : "scc");
      if constexpr (PAT == 11) asm volatile("s_mov_b64 vcc, %0" : : "s"(km) : "vcc");
      if constexpr (PAT == 12) asm volatile("v_and_b32 %0, 0xf0f0f0f0, %0\n\ts_mov_b64 vcc, %2\n\ts_add_u32 %1, %1, 1" : "+v"(v0), "+s"(sa) : "s"(km) : "scc", "vcc");
      // 13: the quad transpose's piece (s_mov_b64 vcc + two DPP selects) behind every other MFMA;  14: the same three instructions, one behind each of three MFMAs out of four... i.e. spread
      if constexpr (PAT == 13 && k % 2 == 0)
        asm volatile("s_mov_b64 vcc, %4\n\tv_cndmask_b32_dpp %0, %2, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %1, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                     : "=&v"(v0), "=&v"(v1) : "v"(v2), "v"(v3), "s"(km) : "vcc");
      if constexpr (PAT == 14 && k % 4 == 0) asm volatile("s_mov_b64 vcc, %0" : : "s"(km) : "vcc");
      if constexpr (PAT == 14 && k % 4 == 1) asm volatile("v_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(v0) : "v"(v2), "v"(v3) : "vcc");
      if constexpr (PAT == 14 && k % 4 == 2) asm volatile("v_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(v1) : "v"(v3), "v"(v2) : "vcc");
      // 15: one VALU behind every MFMA + one s_waitcnt that never blocks;  16: ds_read_b128 per 4 + 2 VALU per MFMA (every slot at 2 or 3 extras)
      if constexpr (PAT == 15) asm volatile("v_and_b32 %0, 0xf0f0f0f0, %0\n\ts_waitcnt vmcnt(60)" : "+v"(v0));
      if constexpr (PAT == 16) {
        asm volatile("v_and_b32 %0, 0xf0f0f0f0, %0\n\tv_lshlrev_b32 %1, 4, %1" : "+v"(v0), "+v"(v1));
        if constexpr (k % 4 == 3) x[(k / 4) % 16] = *reinterpret_cast<const v4i*>(smem + xrd + ((k / 4) % 16) * 1024 + ((k / 4) % 2) * 16384);
      }
      if constexpr (PAT == 7 || PAT == 8) {
        if constexpr (k % 32 == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(ring[(k / 32) % 2]) : "v"(voff), "s"(desc), "n"(2048));
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // once per trip: everything the trip issued has landed (the kernel's counted waits never drain; this costs a little MORE than they do)
    if constexpr (PAT >= 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[0]), "+v"(ring[1])::"memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  int f = (int)(v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ ring[0][0] ^ ring[1][0] ^ sa);
#pragma unroll
  for (int i = 0; i < NACC; ++i) f ^= acc[i][0] ^ acc[i][1] ^ acc[i][2] ^ acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) f ^= x[i][0];
  if (tid == 0) {
    out[2 * blockIdx.x] = t1 - t0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (f == 0x13579bdf) out[0] = (unsigned long long)f;
}

template <int WPS, int PAT>
static void run(const char* name, const unsigned char* g, unsigned long long* out, int ncu, int fill = 0) {
  auto k = issue_kernel<WPS, PAT>;
  const int lds = 96 * 1024, trips = 4000;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  std::vector<unsigned long long> h(2 * ncu);
  double best_c = 1e30, best_r = 0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best_ms = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(ncu), dim3(256 * WPS), lds, 0, g, out, trips, fill);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best_ms = std::min(best_ms, ms);
    CK(hipMemcpy(h.data(), out, sizeof(unsigned long long) * 2 * ncu, hipMemcpyDeviceToHost));
    std::vector<double> c, rt;
    for (int i = 0; i < ncu; ++i) c.push_back((double)h[2 * i]), rt.push_back((double)h[2 * i + 1]);
    std::sort(c.begin(), c.end());
    std::sort(rt.begin(), rt.end());
    if (c[ncu / 2] < best_c) best_c = c[ncu / 2], best_r = rt[ncu / 2];
  }
  const double mfmas = (double)trips * (64 / WPS);  // per wave
  // s_memtime counts at a constant 100 MHz on this part if it equals s_memrealtime; print both and the wall time per MFMA of the SIMD (WPS waves share it)
  const double tops = (double)ncu * 4 * WPS * mfmas * 32768.0 / (best_ms * 1e-3) / 1e12;
  printf("%d wave(s)/SIMD %s pattern %d %-58s  memtime/MFMA %7.2f  realtime(10ns)/MFMA %7.3f  -> %6.2f ns per MFMA of the SIMD; launch %8.1f us = %6.0f TOPS (events)\n", WPS,
         fill ? "random" : "zeros ", PAT, name, best_c / mfmas, best_r / mfmas, best_r * 10.0 / (mfmas * WPS), best_ms * 1e3, tops);
  fflush(stdout);
}

int main() {
  const int ncu = 256;
  unsigned char* g;
  unsigned long long* out;
  CK(hipMalloc(&g, (size_t)ncu * 65536 + 65536));
  {
    std::vector<unsigned char> h((size_t)ncu * 65536 + 65536);
    unsigned sd = 12345;
    for (auto& b : h) sd = sd * 1664525u + 1013904223u, b = (unsigned char)(sd >> 24);
    CK(hipMemcpy(g, h.data(), h.size(), hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&out, sizeof(unsigned long long) * 2 * ncu));
#define BOTH(P, NAME)                 \
  run<1, P>(NAME, g, out, ncu);       \
  run<2, P>(NAME, g, out, ncu);
  BOTH(0, "MFMAs only")
  BOTH(1, "+ 1 VALU per MFMA")
  BOTH(2, "+ 2 VALU per MFMA")
  BOTH(3, "+ 3 VALU per MFMA")
  BOTH(4, "+ 1 ds_read_b128 per 4 MFMAs")
  BOTH(5, "+ 1 ds_read_b128 and 3 VALU per 4 MFMAs")
  BOTH(6, "+ 1 LDS-DMA (x4, + M0) per 16 MFMAs")
  BOTH(7, "+ 1 buffer_load_dwordx4 per 32 MFMAs")
  BOTH(8, "+ all of 4..7 (the per-channel step without its waits)")
  BOTH(9, "+ 3 VALU behind every other MFMA")
  BOTH(10, "+ 2 SALU per MFMA")
  BOTH(11, "+ 3 SALU per MFMA")
  BOTH(12, "+ 1 VALU + 2 SALU per MFMA")
  BOTH(13, "+ transpose piece (s_mov vcc + 2 DPP) behind every other MFMA")
  BOTH(14, "+ the same three instructions, one per MFMA (3 of 4)")
  BOTH(15, "+ 1 VALU + 1 s_waitcnt (never blocks) per MFMA")
  BOTH(16, "+ 2 VALU per MFMA + 1 ds_read_b128 per 4")
  run<1, 0>("MFMAs only", g, out, ncu, 1);
  run<2, 0>("MFMAs only", g, out, ncu, 1);
  run<1, 0>("MFMAs only, 128 workgroups", g, out, 128, 0);
  run<2, 0>("MFMAs only, 128 workgroups", g, out, 128, 0);
  run<1, 0>("MFMAs only, 8 workgroups", g, out, 8, 0);
  run<2, 0>("MFMAs only, 8 workgroups", g, out, 8, 0);
  return 0;
}
