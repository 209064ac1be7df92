/*
 * qqq_amd.h -- C-ABI of the MI355X-native (gfx950) W4A8 GEMM behind QQQ's `qqq_gemm` operator.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  Every entry point
 * cites the reference interface (HandH1998/QQQ, /root/reference) it replaces.  The reference-side
 * binding a QQQ / vLLM maintainer would add is shown in INTEGRATION.md.
 *
 * All device pointers must be resident on device `dev`; every call only ENQUEUES work on
 * `stream` (a hipStream_t passed as void*; NULL = the legacy default stream) and returns without
 * synchronising -- the same contract as the reference (csrc/qqq_gemm.cu:1089).
 * The library allocates no device memory, frees nothing and keeps no state between calls (reference ownership rules:
 * qlinear_marlin.py:97-133) -- except one CU-masked stream + two events per (device, sms, caller stream) once a caller passes an `sms` cap.
 *
 * ABI history.  2: tune fields glds .. split_m.  3: tune.skew; stream tune.fused = 3 changed meaning -- it WAS the in-launch fold over write-through slabs, it IS the
 * arrival-order slot protocol (the last arrival keeps its tile in LDS and adds the others' slots; early ticket, uneven slices by `skew`); measured level with slabs + reduce
 * launch (22.2 vs 21.8 us at 16 tokens), reachable through tune only.  4 (round 6): tune.w8, tune.fused bit 64, qqq_expand_int8 / qqq_w4a8_gemm_ex2 /
 * qqq_quantlinear_forward2 (the opt-in load-time int8 expansion of a layer); the `sms` CU mask laid out as measured.
 */
#ifndef QQQ_AMD_H_
#define QQQ_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QQQ_AMD_ABI_VERSION 4

/* return codes; 0/1/2 are the reference's (csrc/qqq_gemm.cu:947-948, :1002-1003) */
#define QQQ_OK 0
#define QQQ_ERR_PROB_SHAPE 1 /* (m,n,k) not compatible with thread_k/thread_n / group size  */
#define QQQ_ERR_KERN_SHAPE 2 /* no kernel for thread_k/thread_n/groupsize                    */
#define QQQ_ERR_HIP 16       /* a HIP runtime call failed; see qqq_amd_last_error()          */
#define QQQ_ERR_ARG 17       /* NULL or misaligned pointer (A/B/C/D/s3/bias: 16 B, s2: 8 B)   */

/*
 * Replaces `int qqq_cuda(...)` (csrc/qqq_gemm.cu:950-969), argument for argument:
 *   A   int8  [m,k] row-major                                (qlinear_marlin.py:32)
 *   B   int32 [k/16, n*16/8] Marlin/QQQ packed int4 weights  (qlinear_marlin.py:33, pack() :181-262)
 *   C   int32 [max_par*64, n] scratch ("reduce buffer")      (qlinear_marlin.py:34) -- used for split-K
 *       partial sums; contents on return are unspecified, as in the reference
 *   D   fp16  [m,n] row-major output                          (qlinear_marlin.py:35)
 *   s1  f32   [m,1] per-token activation scales               (qlinear_marlin.py:36)
 *   s2  f32   [1,n] per-channel weight scales, stored order   (qlinear_marlin.py:37)
 *   s3  f16   [k/groupsize, n] per-group scales, stored order; ignored when groupsize == -1 (:38)
 *   workspace int32, >= n/128*max_par entries, all zero on entry, all zero on return (:39, .cu:1068)
 *   groupsize -1 (per-channel) or 128                        (csrc/qqq_gemm.cu:1065, :990)
 *   thread_k, thread_n, sms, max_par: reference tuning knobs (qqq_gemm.h:32-35).  thread_k/thread_n
 *       are validated exactly like the reference (is_valid_config, .cu:867-897; CALL_IF table
 *       :935-945) so the same calls fail with the same code, but they do not select CDNA4 tiles;
 *       sms (reference: the number of persistent threadblocks, -1 = every SM; csrc/qqq_gemm.cu:998) is a CU cap: with 0 < sms < the device's
 *       CU count the call's kernels run on a library-owned CU-masked stream (max(sms, 8) CUs, spread evenly over the XCDs: mask bit i is a CU of
 *       XCD i % 8, an XCD without a bit is not restricted) forked from / joined into `stream`; not while `stream` is being captured
 *       (INTEGRATION.md 3); max_par bounds the rows of C that may be used (max_par*64).
 * D[i,j] = fp16_rn( (f32_rn(sum_k A[i,k]*Wq[k,j]) * s2[j]) * s1[i] ), Wq as the reference kernel
 * forms it (csrc/qqq_gemm.cu:146-151, :167-210, :695-700).  int32 accumulators are bit-exact.
 */
int qqq_w4a8_gemm(const void* A, const void* B, void* C, void* D, const void* s1, const void* s2,
                  const void* s3, int prob_m, int prob_n, int prob_k, void* workspace, int groupsize,
                  int dev, void* stream, int thread_k, int thread_n, int sms, int max_par);

/* Tuning / test hooks for the same operation.  All fields 0 => automatic (== qqq_w4a8_gemm). */
typedef struct qqq_tune {
  int kernel;  /* 0 auto, 1 = "stream" (weights straight to VGPRs, 16x16x64 MFMA, small m),
                  2 = "tiled" (LDS-staged 32x32x32 MFMA tiles, large m),
                  3 = "column" (decode: 32 columns x all of K per workgroup, no split-K),
                  4 = "panel" (m-blocks of up to 128 tokens: all tokens of an m-block x bm columns x a K slice per
                      workgroup, weights straight to VGPRs, activations shared through LDS, in-launch split-K),
                  5 = "wide" (256 -- mt = 8: 128 -- tokens x 256 -- bm = 128: 128 -- columns per workgroup, four waves with 512
                      registers each, up to 256 int32 accumulators per lane, weights straight to VGPRs, activations into LDS by
                      LDS-DMA, in-launch split-K through row-major slots of C: from ~320 tokens up) */
  int ksplit;  /* 0 auto, else number of K slices (partials go through C)                   */
  int waves;   /* stream: waves per workgroup (4, 8 or 16); panel (bm = 128): 4 or 8 (two k-groups); 0 auto */
  int fused;   /* split-K finish; 0 auto.  stream: 1 = last-arriving workgroup reduces the slabs in-launch (one ticket per tile in
                  workspace, release / acquire fences), 3 = in-launch through arrival-order slots (two ticket words per tile; a slice
                  writes its partial tile through to the slot of its arrival index, the last arrival keeps its own in LDS and adds
                  the others'; uneven slices by `skew`), 2 = slabs + separate reduce launch.  tiled: 1 = in-launch (K slices of a tile meet in tile-sized int32
                  slots of C, tickets in workspace), 2 = ksplit [m,n] slabs in C + separate reduce launch.
                  tiled / panel / wide in-launch hand-off, OR-ed in: 4 = formal agent-scope acquire fence in front of the
                  fold, 8 = agent-scope release on the depositor's completion count (both off as shipped: the deposits
                  are written through and read with agent-scope loads; the switches exist so that tests run both ways),
                  16 = wide: never keep a deposit in the XCD's L2 (as shipped, slices of a tile that find each other on one
                  XCD do: DESIGN.md 3.4.2), 32 = panel: plain grid order (as shipped the grid of a split K is walked so
                  that the slices of a tile run on ONE XCD whatever the number of strips), 64 = wide, two K slices of
                  256-column tiles: the EXCHANGE hand-off (each slice deposits the row half the other one finishes and finishes
                  its own, even slices; as shipped one slice deposits everything and the other folds, uneven slices -- the two
                  measure level); out (qqq_w4a8_plan): set when the plan exchanges; 128 = wide, two K slices (measurement): the
                  slices of a tile on NEIGHBOURING XCDs instead of one (less fabric traffic in the loop, deposits across the
                  fabric: 5 - 10 % slower, profiles/r06_wide_slices_neighbour_xcds.txt)                               */
  int bm;      /* tiled: rows per workgroup tile (64, 128, 256); panel: COLUMNS per workgroup (128, 256); wide: COLUMNS per
                  workgroup (256; 128 with mt = 16 only: 32 columns per wave); 0 auto */
  int glds;    /* tiled: 1 = direct global->LDS loads, 2 = register staged; 0 auto.
                  wide: 2 = persistent tile walk wherever it applies (one workgroup per CU walks its run of tiles, the next
                  tile's first stages are fetched under the current tile's last ones, LDS-free flush at the seam; needs
                  ksplit = 1, K >= 1024 and at least one tile per CU), 1 = one tile per workgroup, 0 = automatic
                  (the walk for K <= 8192 when there is more than one 256 x 256 tile per CU)                      */
  int pf;      /* stream: prefetch depth in 4 KiB steps per wave (3, 5, 7); column: 1 KiB steps per wave
                  (2..12); panel: weight ring depth in 128-k stages (2, 3, 4; 8 for mt <= 4); wide: weight ring depth in
                  64-k steps -- 4 since round 6 whatever is asked (the packed weights come in as one-word loads, 8 per step: 8 steps
                  would pass the 63 loads a wave can have in flight; -DQQQ_WIDE_DWORD=0 builds take 4 or 8) */
  int stages;  /* tiled + LDS-DMA: ring depth 2..7 (0 auto); panel: activation lead in stages -- with pf = 4 it is 2
                  unless 4 is asked for, otherwise it equals pf */
  int mt;      /* stream: 16-token tiles per workgroup (1..4); column: 1..2; panel: 1, 2, 4, 8; wide: 16, 8; 0 auto */
  int pw;      /* tiled, wide: weight strips per XCD panel of the tile order (4, 8, 16, 32); panel (bm = 256, mt = 8):
                  32-column sets per wave (1, or 2 = 4 waves x 64 columns x 2 k-groups); 0 auto            */
  int nslots;  /* out (qqq_w4a8_plan only): tile-sized slots of C used by the tiled in-launch split-K  */
  int split_m; /* M split (automatic dispatch only; rows are independent): a token count one past a whole number of tiles / rounds of the wide
                  kernel is run as two launches on the same stream -- rows [0, split_m), then the remainder as a call of its own -- when the
                  cost models price the pair at least 7 % below the single launch (N = 8192, K = 21760: 4097 tokens 624 -> 464 us).
                  in: -1 = never split this call, 0 = automatic.  out (qqq_w4a8_plan): the first launch's rows, 0 = one launch;
                  the other out fields then describe the plan of the WHOLE call, which is not the one that runs           */
  int skew;    /* panel / wide, in-launch split-K: uneven K slices -- the LAST slice gets this many 128-k stages more than an even share (the
                  others share what is left evenly), so that it arrives last and finds the other slices' deposits already in memory
                  instead of waiting a hand-off latency for them (arrival order still decides who folds: a matter of time, never of
                  correctness).  in: -1 = even slices, 0 = automatic, 1..63 stages.  out (qqq_w4a8_plan): the stages used.  ABI 3.
                  stream (fused = 3): the same in 64-k steps (1..255). */
  int w8;      /* wide kernel: the layer's expanded int8 weights of qqq_expand_int8 (ABI 4).  in (qqq_w4a8_gemm_ex2): 0 = use them where the call has
                  them and the plan is the wide kernel's, -1 = ignore them (A/B timing); in (qqq_w4a8_plan): 1 = plan for a call that has them.
                  out (qqq_w4a8_plan): 1 = the planned loop reads the expanded weights. */
} qqq_tune_t;

/* As qqq_w4a8_gemm; `tune` may be NULL; if `acc_out` != NULL the raw int32 accumulators
 * ([m,n] row-major; x16 convention in per-channel mode) are also written there; if
 * `bias` != NULL (fp16 [n]) the epilogue adds it in fp16 after the fp16 round, i.e. exactly the
 * reference's separate `D + self.bias` (qlinear_marlin.py:287) without the extra pass over D. */
int qqq_w4a8_gemm_ex(const void* A, const void* B, void* C, void* D, const void* s1, const void* s2,
                     const void* s3, int prob_m, int prob_n, int prob_k, void* workspace,
                     int groupsize, int dev, void* stream, int thread_k, int thread_n, int sms,
                     int max_par, const qqq_tune_t* tune, int32_t* acc_out, const void* bias);

/*
 * Opt-in load-time re-layout of a layer (sits beside QuantLinear.pack, qlinear_marlin.py:181-262; default OFF: nothing changes
 * for a caller that never calls it, and the packed int4 tensor B stays the layer's checkpoint format).
 * A per-group weight is a pure function of (B, s_group): qqq_expand_int8 runs the reference's in-loop re-quantisation (dequant_per_group,
 * csrc/qqq_gemm.cu:167-210: w8 = low byte of fp16((u - 8) * s + 1152) ^ 0x80, bit for bit, wrap region included) ONCE per weight and stores the int8
 * result in the wide kernel's MFMA operand order:
 *   W8[k / 64][n / 64][2 hf + b][lane = 16 h + 4 c + jt][i]  =  w8(k = 64 (k / 64) + 16 h + i, n = 64 (n / 64) + 16 jt + 8 b + 4 hf + c),   i = 0..15
 * (k * n bytes, device memory owned by the caller -- twice the packed tensor; B is still needed: small-m calls keep reading it).
 * qqq_w4a8_gemm_ex2 is qqq_w4a8_gemm_ex with that tensor as a last argument (NULL = none): where the plan of a per-group call is the wide kernel's
 * (MFMA-bound calls, from a few hundred tokens up) its loop reads W8 -- no transpose, no re-quantiser, no group scales: the per-channel loop minus its
 * unpack -- and produces the SAME int32 accumulators and the same D bit for bit; everywhere else W8 is ignored.  groupsize 128, or -1: a per-channel
 * layer expands the same way (the operand `q & 0xF0F0F0F0` / `(q << 4) & 0xF0F0F0F0` = 16 w4 of csrc/qqq_gemm.cu:146-151 as int8; s3 is not read) and
 * measures level with its packed form -- its loop hides the unpack; k % 128 == 0, n % 64 == 0, k * n < 4 GiB; both enqueue on `stream` and return.
 */
int qqq_expand_int8(const void* B, const void* s3, void* W8, int k, int n, int groupsize, int dev, void* stream);
int qqq_w4a8_gemm_ex2(const void* A, const void* B, void* C, void* D, const void* s1, const void* s2,
                      const void* s3, int prob_m, int prob_n, int prob_k, void* workspace,
                      int groupsize, int dev, void* stream, int thread_k, int thread_n, int sms,
                      int max_par, const qqq_tune_t* tune, int32_t* acc_out, const void* bias, const void* W8);

/* The dispatch decision qqq_w4a8_gemm_ex would take for this problem, without touching the GPU (pure host
 * logic; used by tests and tools).  have_scratch / have_workspace: whether C / workspace would be non-NULL.
 * plan_out: kernel, ksplit, fused, waves, pf, mt (stream / column), bm, glds, stages, pw (tiled), bm, mt, pf, stages, pw (panel) as chosen;
 * nslots = number of tile-sized slots of C used by the tiled in-launch split-K (0 = slabs or no split). */
int qqq_w4a8_plan(int prob_m, int prob_n, int prob_k, int groupsize, int max_par, int have_scratch,
                  int have_workspace, const qqq_tune_t* tune, qqq_tune_t* plan_out);

/* Diagnostic (pure host logic): the dispatcher's cost models' price, in microseconds on an MI355X, of each kernel family for this problem --
 * out[0] column, [1] stream, [2] panel, [3] wide; <= 0 where a family is not a candidate.  tools/cost_model_report.py holds these against the
 * committed hardware measurements (profiles/r05_cost_model_error.txt). */
int qqq_w4a8_model_us(int prob_m, int prob_n, int prob_k, int groupsize, int max_par, double* out);

/*
 * Fused per-token dynamic int8 quantisation; replaces the ~8 torch launches of
 * QuantLinear.dynamic_quant (qlinear_marlin.py:265-268):
 *   s1[i]  = f32( f16_rn( max_k|x[i,k]| * (1.0f/127.0f) ) )     (torch-GPU lowering of .div(127.0))
 *   xq[i,k]= int8( clamp( rint( f32(x[i,k]) / s1[i] ), -128, 127 ) )
 * x fp16 [m,k] row-major, xq int8 [m,k], s1 f32 [m].  k must be a multiple of 8.
 */
int qqq_dynamic_quant(const void* x, void* xq, void* s1, int m, int k, int dev, void* stream);

/* QuantLinear.forward in one host call (qlinear_marlin.py:270-288): qqq_dynamic_quant(x -> xq, s1) followed by
 * qqq_w4a8_gemm_ex(xq, ..., bias) on the same stream.  x fp16 [m,k]; xq int8 [m,k] and s1 f32 [m] are caller-owned
 * scratch/outputs; the other arguments as in qqq_w4a8_gemm_ex.  Exists because at decode sizes the two kernels take
 * less GPU time than two trips through a Python binding take on the host. */
int qqq_quantlinear_forward(const void* x, void* xq, void* s1, const void* B, void* C, void* D, const void* s2,
                            const void* s3, int m, int n, int k, void* workspace, int groupsize, int dev,
                            void* stream, int max_par, const void* bias);

/* ... with the layer's expanded int8 weights (qqq_expand_int8; NULL = none), as qqq_w4a8_gemm_ex2 */
int qqq_quantlinear_forward2(const void* x, void* xq, void* s1, const void* B, void* C, void* D, const void* s2,
                             const void* s3, int m, int n, int k, void* workspace, int groupsize, int dev,
                             void* stream, int max_par, const void* bias, const void* W8);

/*
 * int4 packer / unpacker for the Marlin/QQQ weight layout -- replaces the python-loop interleave of
 * QuantLinear.pack (qlinear_marlin.py:228-248; layout from _get_perms, :147-176):
 *   codes int8 [k,n] row-major (signed int4 in [-8,7] per-channel, unsigned u in [0,15] per-group)
 *   B     int32 [k/16, 2n]; word B[kt][128*ng + 16*c + 4*kq + jt] holds k = 16*kt + 4*kq + r (r = 0..3) of columns
 *         n = 64*ng + 16*jt + 8*b + c (b = 0,1); nibble p <-> (b,r) = (1-(p&1), p>>1) per-channel,
 *         ((p&3)>>1, 2*(p&1) + (p>>2)) per-group.
 * on_device != 0: both buffers are device memory, the work is enqueued on `stream`; on_device == 0: both are host
 * memory and the call is synchronous (offline checkpoint conversion).  k % 16 == 0, n % 64 == 0, 8-byte aligned.
 */
int qqq_pack_int4(const void* codes, void* B, int k, int n, int grouped, int on_device, int dev, void* stream);
int qqq_unpack_int4(const void* B, void* codes, int k, int n, int grouped, int on_device, int dev, void* stream);

int qqq_amd_abi_version(void);
const char* qqq_amd_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* QQQ_AMD_H_ */
