/*
 * qqq_amd_dev.h -- C-ABI of the TEST / TUNING companion library (qqq_amd/libqqq_amd_dev.so).
 *
 * Not part of the drop-in boundary (that is include/qqq_amd.h, libqqq_amd.so): hardware probes used by
 * tests/test_gpu_probe.py, a read-bandwidth probe, and the event-timed call loop of bench.py / tools.
 * The GEMM kernels are not compiled into this library; qqq_dev_bench_gemm times the operator library's
 * qqq_w4a8_gemm_ex through the function pointer it is given.
 */
#ifndef QQQ_AMD_DEV_H_
#define QQQ_AMD_DEV_H_

#include "qqq_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* signature of qqq_w4a8_gemm_ex (include/qqq_amd.h) */
typedef int (*qqq_gemm_ex_fn)(const void* A, const void* B, void* C, void* D, const void* s1, const void* s2,
                              const void* s3, int prob_m, int prob_n, int prob_k, void* workspace, int groupsize,
                              int dev, void* stream, int thread_k, int thread_n, int sms, int max_par,
                              const qqq_tune_t* tune, int32_t* acc_out, const void* bias);

/* signature of qqq_w4a8_gemm_ex2 (the same + the layer's expanded int8 weights) */
typedef int (*qqq_gemm_ex2_fn)(const void* A, const void* B, void* C, void* D, const void* s1, const void* s2,
                               const void* s3, int prob_m, int prob_n, int prob_k, void* workspace, int groupsize,
                               int dev, void* stream, int thread_k, int thread_n, int sms, int max_par,
                               const qqq_tune_t* tune, int32_t* acc_out, const void* bias, const void* W8);

/* Runs `iters` calls of `gemm_ex` back to back on `stream`, call i using the weight buffer Bs[i % nB] (rotate
 * >= 4 x 89 MB buffers to defeat the 256 MiB Infinity Cache), each bracketed by its own hipEvent pair recorded on
 * `stream`; synchronises the stream and writes the `iters` durations in milliseconds to ms_each (host memory). */
int qqq_dev_bench_gemm(qqq_gemm_ex_fn gemm_ex, const void* A, const void* const* Bs, int nB, void* C, void* D,
                       const void* s1, const void* s2, const void* s3, int prob_m, int prob_n, int prob_k,
                       void* workspace, int groupsize, int dev, void* stream, int max_par, const qqq_tune_t* tune,
                       int iters, float* ms_each);

/* the same through qqq_w4a8_gemm_ex2: call i also gets the expanded weights W8s[i % nB] (W8s may be NULL: none) */
int qqq_dev_bench_gemm2(qqq_gemm_ex2_fn gemm_ex2, const void* A, const void* const* Bs, const void* const* W8s, int nB, void* C, void* D,
                        const void* s1, const void* s2, const void* s3, int prob_m, int prob_n, int prob_k,
                        void* workspace, int groupsize, int dev, void* stream, int max_par, const qqq_tune_t* tune,
                        int iters, float* ms_each);

/* One MFMA on raw per-lane operands, so that the lane<->element maps the kernels rely on are checked on the device.
 * kind 16: v_mfma_i32_16x16x64_i8 (a,b: 64 lanes x 16 B; out: 64 x 4 int32)
 * kind 32: v_mfma_i32_32x32x32_i8 (a,b: 64 lanes x 16 B; out: 64 x 16 int32) */
int qqq_dev_probe_mfma(int kind, const void* a, const void* b, void* out, int dev, void* stream);
/* copies 64 x 16 B through LDS with global_load_lds; lane l reads src chunk perm[l] */
int qqq_dev_probe_glds(const void* src, const void* perm, void* dst, int dev, void* stream);
/* the kernels' per-group int4 -> int8 re-quantiser (the counterpart of dequant_per_group, csrc/qqq_gemm.cu:167-210)
 * on raw operands: for i < n, out[2i] / out[2i+1] = the two int8 quadruples made of packed word q[i] (uint32) with
 * the fp16 scales (bit patterns) s0[i] / s1[i] */
int qqq_dev_probe_dequant(const void* q, const void* s0, const void* s1, void* out, int n, int dev, void* stream);
/* Read-bandwidth probe (tools/probe_fill.py): `nwg` workgroups of 512 threads each stream `bytes_per_wg` bytes
 * `reps` times from src + wg_stride * workgroup (wg_stride 0: a shared L2-resident window = per-CU L2->L1 fill rate;
 * wg_stride == bytes_per_wg: disjoint windows = HBM streaming); `unroll` 2 or 8 independent 16-byte loads per thread.
 * Writes the duration of one launch in milliseconds to ms_out (host memory). */
int qqq_dev_probe_fill(const void* src, size_t wg_stride, size_t bytes_per_wg, int nwg, int reps, int unroll, void* sink,
                       int dev, void* stream, float* ms_out);
/* Sustained matrix-pipe rate probe (tools/mfma_ceiling.py): `nwg` workgroups of 8 waves, each wave `iters` k-steps of
 * 8 x v_mfma_i32_32x32x32_i8 (the tiled kernel's 4 x 2 fragment grid) on operands from `ops` (>= nwg * 64 KiB of
 * random bytes or zeros).  mode 0: register operands only; 1: + LDS fragment reads at the tiled kernel's volume;
 * 2: + the per-channel int4 unpack; 3: register operands only, on v_mfma_i32_16x16x64_i8 (16 per k-step).  Times ONE launch (after a warm-up launch); ms_out in host memory. */
int qqq_dev_probe_mfma_rate(int mode, const void* ops, int nwg, int iters, void* sink, int dev, void* stream,
                            float* ms_out);
/* Placement probe: `nwg` one-wave workgroups each record {HW_REG_XCC_ID, HW_REG_HW_ID} into out[2 * workgroup] (uint32 pairs, device memory) and hold
 * their CU for `hold_us` microseconds.  cu_mask != NULL: launched on a temporary stream created with that CU mask (mask_words 32-bit words,
 * hipExtStreamCreateWithCUMask) and synchronised before return -- how the bit -> (XCD, CU) map of the mask was read off the hardware; NULL: on `stream`. */
int qqq_dev_probe_placement(const uint32_t* cu_mask, int mask_words, int nwg, int hold_us, void* out, int dev, void* stream);
const char* qqq_dev_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* QQQ_AMD_DEV_H_ */
