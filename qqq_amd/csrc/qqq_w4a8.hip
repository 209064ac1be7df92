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
//  * "column" kernel (decode: m <= 8, up to 32 tokens while m*K is small): HBM-bound.  32 weight columns x all of K per workgroup, so a wide layer
//    fills the chip without split-K (one launch per call); packed words re-distributed between lanes with DPP; non-temporal weight loads (round 5).
//  * "stream" kernel (a few tens of tokens; small layers up to ~256): HBM-bound.  v_mfma_i32_16x16x64_i8; a lane (i = 8*g + c, h)
//    loads its 64 weight bytes of k-tile 4*s + h straight from HBM into VGPRs (no LDS: the
//    weights are used once), a wave eats 128 columns x 64 k = 4 KiB per step, waves of a
//    workgroup split K and reduce through LDS, workgroups split K through int32 slabs in the
//    reduce buffer C (int32 addition is associative: bit-exact for every split).
//  * "panel" kernel (about 64 .. 1024 tokens): all tokens of a 128-token m-block x 128 / 256 columns x a K slice per
//    workgroup; the waves split the columns (weights HBM -> VGPR, once per workgroup), the activations are shared
//    through LDS, software-pipelined 16x16x64 MFMAs, in-launch split-K with one slot of C per depositing slice.
//    Uneven K slices (round 5): the last slice is a few stages longer, arrives last and finds the other deposits complete; the slices of a tile run on one XCD.
//  * "wide" kernel (from ~320 tokens up, qqq_wide.hip.h): 256 x 256 / 256 x 128 / 128 x 256 tiles, four waves with 512 registers each, every
//    instruction of the loop placed by hand around in-place MFMAs -- since round 6 against the measured issue model of a wave alone on its SIMD: one-instruction
//    items dealt to the issue slots by capacity, weights as one-word loads (no transpose), running load cursors --, activations by LDS-DMA, persistent tile walk on
//    short-K layers; opt-in expanded int8 weights (qqq_expand_int8) for the per-group mode.
//  * "tiled" kernel (round 1; tune.kernel = 2 only since round 5 -- the fuzzers' independent reference, the fallback beyond 4 GB of packed
//    weights): v_mfma_i32_32x32x32_i8; BMx256 tiles, BK = 128, activations and RAW packed weights staged in LDS by LDS-DMA (XOR-swizzled
//    16-byte chunks so that every fragment read is bank-conflict free), continuous fragment pipeline, XCD-aware tile order,
//    in-launch split-K through tile-sized slots of C.
//  Host side: make_plan() picks family / tile / split from measured cost models whose rates are all GENERATED (qqq_rates.h, tools/fit_rates.py: the panel and wide
//  kernels' tables, the small-m forms of the column / stream kernels, the 64-token m-block form) -- the stream kernel's hand-fitted many-m-block branch on large layers left the automatic path in round 6 --
//  and held against the committed measurements by tools/cost_model_report.py; the C-ABI entry points are at the end of the file.
//
// The accumulators are the reference's: per-channel weights enter as 16*w4 (high nibble of each
// byte) and pack() has pre-divided s_channel by 16; per-group weights are re-quantised to int8
// with ONE packed-fp16 FMA exactly like dequant_per_group (csrc/qqq_gemm.cu:167-210).

#include <cmath>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "qqq_common.hip.h"
#include "qqq_stream.hip.h"
#include "qqq_column.hip.h"
#include "qqq_panel.hip.h"
#include "qqq_tiled.hip.h"
#include "qqq_wide.hip.h"
#include "qqq_small.hip.h"
#include "qqq_rates.h"

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

// `sms` as a CU cap (the reference launches `sms` persistent threadblocks, csrc/qqq_gemm.cu:998, :1016-1036): the call's kernels run on a
// library-owned stream created with a CU mask of `sms` CUs, forked from and joined back into the caller's stream with two events -- the
// caller sees the usual stream order and `sms` CUs' worth of occupancy, the other CUs stay free for whatever else it runs (RCCL's kernels
// in the sharded sweep).
// The mask, as read off the hardware (tools/cu_mask_probe.py, profiles/r06_cu_mask_probe.txt): bit i is CU i / 8 of XCD
// i % 8 -- the bits are dealt round-robin over the XCDs --, and an XCD whose bits are ALL zero is not restricted at all.  So the FIRST `sms`
// bits are set (an even spread: sms / 8 CUs per XCD, the first sms % 8 XCDs one more), and the smallest cap that holds is 8 (one CU per
// XCD): 0 < sms < 8 runs on 8 CUs.  (Round 5 set every (CUs / sms)-th bit: sms = 64 enabled all of XCDs 3 and 7 and left the other six
// unrestricted -- no cap at all.)
// One stream + two events per (device, sms, caller stream) -- callers on independent streams do not serialise on one masked stream --, created
// on first use and kept for the life of the process (at most 64 entries; beyond that callers share the (device, sms) entry of the NULL stream);
// the fork / launch / join sequence of a call holds the entry's lock.  A call whose stream is being CAPTURED is not masked: kernel nodes carry
// no CU mask (a replayed graph is never capped), and pulling a shared library-owned stream into somebody's capture would make every other
// capped call on it part of that graph (or an error) until the capture ends.
struct MaskedStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  std::mutex mu;
};
static int masked_cus(int sms) { return sms < 8 ? 8 : sms; }
static MaskedStream* masked_stream(int dev, int sms, int cus, hipStream_t caller) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, hipStream_t>, MaskedStream*> cache;
  std::lock_guard<std::mutex> lk(mu);
  if (cache.size() >= 64 && cache.find({dev, sms, caller}) == cache.end()) caller = nullptr;
  auto it = cache.find({dev, sms, caller});
  if (it != cache.end()) return it->second;
  uint32_t mask[16] = {};
  for (int i = 0; i < masked_cus(sms) && i < cus && i < 512; ++i) mask[i >> 5] |= 1u << (i & 31);
  MaskedStream* m = new MaskedStream;
  if (hipExtStreamCreateWithCUMask(&m->s, (uint32_t)((cus + 31) / 32), mask) != hipSuccess ||
      hipEventCreateWithFlags(&m->fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&m->join, hipEventDisableTiming) != hipSuccess) {
    delete m;  // (a stream or event that was created stays with the runtime: this path is an out-of-resources error)
    return nullptr;
  }
  cache[{dev, sms, caller}] = m;
  return m;
}
// the fork into a masked stream, joined back into the caller's stream on EVERY way out of the call (error returns included: the caller's
// stream must not be left waiting on nothing, nor the masked stream's work un-ordered against what the caller enqueues next)
struct MaskedFork {
  MaskedStream* ms = nullptr;
  hipStream_t caller = nullptr;
  std::unique_lock<std::mutex> lock;
  hipError_t join() {
    if (!ms) return hipSuccess;
    hipError_t e = hipEventRecord(ms->join, ms->s);
    if (e == hipSuccess) e = hipStreamWaitEvent(caller, ms->join, 0);
    ms = nullptr;
    lock.unlock();
    return e;
  }
  ~MaskedFork() { (void)join(); }
};

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
  int skew;    // panel: 128-k stages the last K slice gets on top of an even share (0 = even slices)
  int hflags;  // in-launch split-K hand-off switches (tune.fused bits 2 / 3 / 4): 1 = formal acquire fence, 2 = release on publish, 4 = never L2-local deposits
  hipStream_t stream;
};

template <int MT, bool GROUPED, int WAVES, int PF>
static hipError_t launch_stream_t(const LaunchArgs& a, int ksplit, int fused) {
#ifdef QQQ_DEV_WIDE_ONLY  // measurement builds (tools/ab.py variants of the wide kernel): the other families are not compiled
  return hipErrorNotSupported;
#else
  dim3 grid((a.N + 127) / 128, ksplit & 0xffff, (a.M + 16 * MT - 1) / (16 * MT));
  hipLaunchKernelGGL((qqq_stream_kernel<MT, GROUPED, WAVES, PF>), grid, dim3(WAVES * 64), 0, a.stream,
                     a.A, a.B, a.C, a.D, a.s1, a.s2, a.s3, a.acc_out, a.tickets, a.bias, a.M, a.N, a.K,
                     ksplit, fused);
  return hipGetLastError();
#endif
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
#ifdef QQQ_DEV_WIDE_ONLY  // measurement builds (tools/ab.py variants of the wide kernel): the other families are not compiled
  return hipErrorNotSupported;
#else
  dim3 grid(a.N / 32, ksplit, (a.M + 16 * MT - 1) / (16 * MT));
  hipLaunchKernelGGL((qqq_column_kernel<MT, GROUPED, WAVES, PF>), grid, dim3(WAVES * 64), 0, a.stream, a.A,
                     a.B, a.C, a.D, a.s1, a.s2, a.s3, a.acc_out, a.bias, a.M, a.N, a.K, ksplit);
  return hipGetLastError();
#endif
}

template <bool GROUPED>
static hipError_t launch_column_g(const LaunchArgs& a, int mt, int pf, int ksplit, int waves) {
  if (waves == 16 && mt == 1) return launch_column_t<1, GROUPED, 16, 3>(a, ksplit);  // (tune.waves = 16: sixteen waves split K inside the workgroup)
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

static hipError_t launch_column(const LaunchArgs& a, bool grouped, int mt, int pf, int ksplit, int waves = 8) {
  return grouped ? launch_column_g<true>(a, mt, pf, ksplit, waves) : launch_column_g<false>(a, mt, pf, ksplit, waves);
}

template <int MT, bool GROUPED, int WN, int KG, int PFS, int XL, int HW = 1>
static hipError_t launch_panel_t(const LaunchArgs& a, int ksplit) {
#ifdef QQQ_DEV_WIDE_ONLY  // measurement builds (tools/ab.py variants of the wide kernel): the other families are not compiled
  return hipErrorNotSupported;
#else
  constexpr int ROWS = 16 * MT, BN = 32 * WN * HW;
  constexpr int XBUF = (qqq_panel_relaxed(KG, PFS, XL, HW) ? 4 : KG == 2 ? 2 : 3) * ROWS * 128;
  constexpr int EP = ROWS * (BN + 4) * 4;
  constexpr int RED = (KG == 2) ? WN * MT * HW * 2048 : 0;
  constexpr int LDS = XBUF > EP ? (XBUF > RED ? XBUF : RED) : (EP > RED ? EP : RED);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static bool attr_set[64] = {};  // per instantiation, per device
  auto kern = qqq_panel_kernel<MT, GROUPED, WN, KG, PFS, XL, HW>;
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur < 0 || cur >= 64 || !attr_set[cur]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    if (cur >= 0 && cur < 64) attr_set[cur] = true;
  }
  dim3 grid((a.N + BN - 1) / BN, ksplit, (a.M + ROWS - 1) / ROWS);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WN * KG), LDS, a.stream, a.A, a.B, a.C, a.D, a.s1, a.s2, a.s3, a.acc_out,
                     a.tickets, a.bias, a.M, a.N, a.K, ksplit | ((a.hflags & 0xff) << 16) | ((ksplit > 1 ? a.skew & 0x3f : 0) << 24));
  return hipGetLastError();
#endif
}

template <int MT, bool GROUPED, int PFS, int XL>
static hipError_t launch_panel_shape(const LaunchArgs& a, int bn, int waves, int cw, int ksplit) {
  if constexpr (MT == 8 && PFS >= 3 && (PFS == XL || (PFS == 4 && XL == 2))) {  // 64 columns per wave (two k-groups of 4 waves)
    if (bn == 256 && cw == 2) return launch_panel_t<MT, GROUPED, 4, 2, PFS, XL, 2>(a, ksplit);
    // (128-column strips as four waves x 64 columns x two k-groups -- one wave per SIMD, half the fragment reads and unpack work per MFMA --
    //  were instantiated and measured in round 5: 46.1 vs 37.3 us at 128 tokens, 21.7 vs 16.5 on 4096 x 4096: nobody hides a lone
    //  wave's stalls.  Not kept; profiles/r05_uneven_k_slices.txt has the lines.)
  }
  if (bn == 256) return launch_panel_t<MT, GROUPED, 8, 1, PFS, XL>(a, ksplit);
  if (waves == 4) return launch_panel_t<MT, GROUPED, 4, 1, PFS, XL>(a, ksplit);
  // (a NINTH wave feeding the activations by LDS-DMA -- so that they do not queue behind the weight loads in the compute waves' in-order
  //  return queues -- was built and measured in round 5: bit-exact, 150 cases, and level with this kernel everywhere, 35.4 vs 35.4 us at 128
  //  tokens: it is the L2 <-> CU traffic of the activations that costs, not how it is issued.  Not kept; profiles/r05_panel_feeder.txt)
  return launch_panel_t<MT, GROUPED, 4, 2, PFS, XL>(a, ksplit);
}

template <int MT, bool GROUPED>
static hipError_t launch_panel_pf(const LaunchArgs& a, int bn, int waves, int cw, int pfs, int xl, int ksplit) {
  // XL (activation lead) = PFS (weight lead) unless asked otherwise: loads return in order, so a shorter activation
  // lead would force the weight loads issued before it to land early and cut their effective lead to XL + 1
  if (pfs <= 2) return launch_panel_shape<MT, GROUPED, 2, 2>(a, bn, waves, cw, ksplit);
  if constexpr (MT <= 4) {
    if (pfs >= 8) return launch_panel_shape<MT, GROUPED, 8, 8>(a, bn, waves, cw, ksplit);
  }
  if (pfs >= 6 || pfs == 3) return launch_panel_shape<MT, GROUPED, 3, 3>(a, bn, waves, cw, ksplit);
  if (xl == 2) return launch_panel_shape<MT, GROUPED, 4, 2>(a, bn, waves, cw, ksplit);
  return launch_panel_shape<MT, GROUPED, 4, 4>(a, bn, waves, cw, ksplit);
}

template <bool GROUPED>
static hipError_t launch_panel_g(const LaunchArgs& a, int mt, int bn, int waves, int cw, int pfs, int xl, int ksplit) {
  switch (mt) {
    case 1: return launch_panel_pf<1, GROUPED>(a, bn, waves, cw, pfs, xl, ksplit);
    case 2: return launch_panel_pf<2, GROUPED>(a, bn, waves, cw, pfs, xl, ksplit);
    case 4: return launch_panel_pf<4, GROUPED>(a, bn, waves, cw, pfs, xl, ksplit);
    default: return launch_panel_pf<8, GROUPED>(a, bn, waves, cw, pfs, xl, ksplit);
  }
}

static hipError_t launch_panel(const LaunchArgs& a, bool grouped, int mt, int bn, int waves, int cw, int pfs, int xl, int ksplit) {
  return grouped ? launch_panel_g<true>(a, mt, bn, waves, cw, pfs, xl, ksplit) : launch_panel_g<false>(a, mt, bn, waves, cw, pfs, xl, ksplit);
}

template <int BM, int MTW, int JW, int NB, bool GROUPED, int NS>
static hipError_t launch_tiled_t(const LaunchArgs& a, int ksplit, int nslots, int pw) {
#ifdef QQQ_DEV_WIDE_ONLY  // measurement builds (tools/ab.py variants of the wide kernel): the other families are not compiled
  return hipErrorNotSupported;
#else
  constexpr int WAVES = (BM / (32 * MTW)) * (4 / JW) * (2 / NB);
  constexpr int NT = WAVES * 64;
  constexpr int STAGE = 8 * 2048 + BM * 128 + ((NS > 0 && GROUPED) ? WAVES * 512 : 0);
  constexpr int RING = ((NS == 5 || NS == 6 || NS == 7) ? 3 : (NS > 0 ? NS : 2)) * STAGE;
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
                     a.acc_out, a.bias, a.M, a.N, a.K, ksplit | (a.hflags << 16), tiles_m, tiles_n, a.tickets, nslots, pw);
  return hipGetLastError();
#endif
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
      if (stages == 7) return launch_tiled_t<256, 2, 2, 2, GROUPED, 7>(a, ksplit, nslots, pw);
      return launch_tiled_t<256, 2, 2, 2, GROUPED, 3>(a, ksplit, nslots, pw);
  }
}

static hipError_t launch_tiled(const LaunchArgs& a, bool grouped, int bm, int stages, int ksplit, int nslots, int pw) {
  return grouped ? launch_tiled_bm<true>(a, bm, stages, ksplit, nslots, pw)
                 : launch_tiled_bm<false>(a, bm, stages, ksplit, nslots, pw);
}

// compute units of device `dev` (queried for THAT device, cached)
static int device_cus_of(int dev) {
  static int cus[64] = {};
  if (dev >= 0 && dev < 64 && cus[dev] > 0) return cus[dev];
  int n = 0;
  if (dev < 0 || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (dev >= 0 && dev < 64) cus[dev] = n;
  return n;
}
// The CU count plans and launches of THIS call work with: qqq_w4a8_gemm_ex sets it from the device it was given (capped by the
// reference's `sms` argument) before it plans, so the plan and the launch grid can never disagree -- the cost models' rounds of
// workgroups, the K splits that fill one round and the tile walk's grid all read it (round 6; ADVICE round 5: a capped call used to be
// planned for a full chip); qqq_w4a8_plan -- pure host logic, no HIP call -- plans for the MI355X's 256.
static thread_local int t_call_cus = 256;
static int device_cus() { return t_call_cus; }
struct CallCus {
  int prev;
  explicit CallCus(int n) : prev(t_call_cus) { t_call_cus = n > 0 ? n : 256; }
  ~CallCus() { t_call_cus = prev; }
};

// the persistent tile walk needs whole tiles (no K split), a K range longer than its prefetch leads and at least one tile per
// workgroup of its grid (a multiple of 8: workgroup b lands on XCD b % 8)
static bool wide_chain_ok(int M, int N, int K, int rows, int bn, int ksplit) {
  const long long tl = (long long)((M + rows - 1) / rows) * ((N + bn - 1) / bn);
  return ksplit == 1 && K / 128 >= 8 && tl >= (long long)(device_cus() & ~7) && (device_cus() & ~7) >= 8;
}

// MODE: 0 per-channel, 1 per-group (re-quantised in the loop), 2 = expanded int8 weights (a.B is the W8 tensor of qqq_expand_int8)
template <int MODE, int MT, int P, int RS, int HW, bool CHAIN = false>
static hipError_t launch_wide_t(const LaunchArgs& a, int pw, int ksplit) {
  constexpr int ROWS = 16 * MT, BN = 128 * HW;
  constexpr int XBUF = P * ROWS * 128, EP = (HW == 2 ? 8 * MT : 16 * MT) * (BN + 4) * 4 + 16;  // + the ticket exchange word
  constexpr int CH = XBUF + 2 * (ROWS * 4 + BN * 6);  // CHAIN: stage buffers + two scale regions, no epilogue image
  constexpr int LDS = CHAIN ? CH : (XBUF > EP ? XBUF : EP);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static bool attr_set[64] = {};  // per instantiation, per device
  auto kern = qqq_wide_kernel<MODE, MT, P, RS, HW, CHAIN>;
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur < 0 || cur >= 64 || !attr_set[cur]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    if (cur >= 0 && cur < 64) attr_set[cur] = true;
  }
  const int tiles_m = (a.M + ROWS - 1) / ROWS, tiles_n = (a.N + BN - 1) / BN;
  const int grid = CHAIN ? (device_cus() & ~7) : tiles_m * tiles_n * ksplit;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, a.stream, a.A, a.B, a.C, a.D, a.s1, a.s2, a.s3,
                     a.acc_out, a.tickets, a.bias, a.M, a.N, a.K, tiles_m, tiles_n, pw,
                     ksplit | ((a.hflags & 0xff) << 16) | ((!CHAIN && ksplit > 1 ? a.skew & 0x3f : 0) << 24));
  return hipGetLastError();
}

// mt: 16 (256-token tiles) or 8 (128-token tiles); pf: weight ring in 64-k steps (4 or 8); four LDS stage buffers;
// bn: 256 columns per workgroup, or (mt = 16 only) 128: 32 columns per wave
template <int MODE, int MT>
static hipError_t launch_wide_m(const LaunchArgs& a, int pf, int pw, int ksplit) {
  return pf == 4 ? launch_wide_t<MODE, MT, 4, 4, 2>(a, pw, ksplit) : launch_wide_t<MODE, MT, 4, 8, 2>(a, pw, ksplit);
}
// the persistent tile walk: the ring depth each mode runs by default (per-channel 4 steps, per-group 8; expanded weights: 256 x 256 tiles only)
static hipError_t launch_wide_chain(const LaunchArgs& a, int mode, int mt, int bn, int pw) {
  // (expanded weights: ring of 4 steps -- 64 registers; with 8 the walk's seam spills inside the stage loop: 32 registers, 75 scratch instructions)
  if (mode == 2) return launch_wide_t<2, 16, 4, 4, 2, true>(a, pw, 1);
  const bool grouped = mode == 1;
  constexpr int GRS = QQQ_WIDE_DWORD != 0 ? 4 : 8;  // per-group ring depth of the walk (dword weight loads: 8 loads per step -- a ring of 8 steps would pass vmcnt's 63)
  if (bn == 128) return grouped ? launch_wide_t<1, 16, 4, GRS, 1, true>(a, pw, 1) : launch_wide_t<0, 16, 4, 4, 1, true>(a, pw, 1);
  if (mt == 8) return grouped ? launch_wide_t<1, 8, 4, GRS, 2, true>(a, pw, 1) : launch_wide_t<0, 8, 4, 4, 2, true>(a, pw, 1);
  return grouped ? launch_wide_t<1, 16, 4, GRS, 2, true>(a, pw, 1) : launch_wide_t<0, 16, 4, 4, 2, true>(a, pw, 1);
}
// Automatic choice (profiles/r04_tile_walk_sweep.txt: plain vs walk over nine layer shapes x five token counts x both modes).
// A seam costs 5.5 us (per-group 7) where the plain grid pays 9 us between two tiles of a CU (epilogue 6.3 + relaunch 0.2 +
// prologue 2.6), and the walk's stage loop pays ~2-3 % for its per-stage bookkeeping: it wins where tiles are short and every CU
// gets more than one -- K <= 6144 (4096 / 5120-deep layers: +4 ... +8 % from two tiles per CU on; +9 ... +15 % on a slow box,
// profiles/r04_walk_zero_operands.txt).  At K = 8192 the sweep's box read it neutral (-1.4 ... +1.7 %), three other boxes +1 ... +6 %
// (profiles/r04_dispatch_check_mid_shapes.txt, r04_walk_larger_k.txt): on, since the end of round 4.  K = 11008: -3.6 ... +2 % box to box,
// K = 21760: -6 ... +1 %: off.  256 x 256 tiles only: the 128-column shape loses with it, the 128-token shape gains less.
static bool wide_chain_pays(long long tiles, int K, int mt, int bn) {
  return mt == 16 && bn == 256 && K / 128 <= 64 && tiles > (long long)(device_cus() & ~7);
}
static hipError_t launch_wide(const LaunchArgs& a, int mode, int mt, int bn, int pf, int pw, int ksplit, bool chain = false) {
  if (chain) return launch_wide_chain(a, mode, mt, bn, pw);
  if (mode == 2) {
    // expanded weights: ring of 4 steps in every shape (the ring holds ready operands -- 16 registers per step, not 8; 4 measured 1 - 1.5 % ahead of 8 on the
    // 256 x 256 tiles, profiles/r06_w8_first_numbers.txt; with 8 the 128-token shape parks ring registers in accumulation registers)
    if (bn == 128) return launch_wide_t<2, 16, 4, 4, 1>(a, pw, ksplit);
    if (mt == 8) return launch_wide_t<2, 8, 4, 4, 2>(a, pw, ksplit);
    return launch_wide_t<2, 16, 4, 4, 2>(a, pw, ksplit);
  }
  const bool grouped = mode == 1;
  if (bn == 128) {
    if (grouped) return pf == 4 ? launch_wide_t<1, 16, 4, 4, 1>(a, pw, ksplit) : launch_wide_t<1, 16, 4, 8, 1>(a, pw, ksplit);
    return pf == 4 ? launch_wide_t<0, 16, 4, 4, 1>(a, pw, ksplit) : launch_wide_t<0, 16, 4, 8, 1>(a, pw, ksplit);
  }
  if (mt == 8) return grouped ? launch_wide_m<1, 8>(a, pf, pw, ksplit) : launch_wide_m<0, 8>(a, pf, pw, ksplit);
  return grouped ? launch_wide_m<1, 16>(a, pf, pw, ksplit) : launch_wide_m<0, 16>(a, pf, pw, ksplit);
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- cost models (microseconds) used by the automatic dispatch; constants fitted to profiles/r02_dispatch_check*.txt ----
// tiled: rounds(tiles x ksplit over 256 CUs) x tile_time(rows, K / ksplit, rate(shape)) + split-K cost,
// per-shape rates (TOPS at large m) measured on MI355X.  Bigger tiles are more efficient per MFMA but quantise worse
// over the CUs; split-K fills idle CUs at the price of int32 partial-sum traffic.  Wave shapes per mode: per-channel
// keeps 64x128 wave tiles (least LDS traffic); per-group uses the column-owner shapes (258 / 130): every weight
// re-quantised once per workgroup.
static double tiled_estimate(int M, int N, int K, bool grouped, bool have_scratch, long long cap_rows, long long cap_tickets,
                             bool slabs_only, int* bm_out, int* ks_out) {
  const long long strips = (N + 255) / 256;
  const long long cap_ints = cap_rows * (long long)N;
  auto slot_count = [&](int rows, int ks) -> int {
    if (!have_scratch || ks < 2 || slabs_only) return 0;
    const long long tl = (long long)((M + rows - 1) / rows) * strips;
    long long S = cap_ints / (tl * rows * 256);
    if (S > ks - 1) S = ks - 1;
    while (S > 0 && tl * (1 + S) > cap_tickets) --S;
    return (int)S;
  };
  // {several workgroups co-resident per CU, a single one} -- small tiles lose efficiency when alone on a CU
  const double rate256 = grouped ? 1950.0 : 2500.0;
  const double rate128[2] = {grouped ? 1360.0 : 2050.0, grouped ? 1320.0 : 1650.0};
  const double rate64[2] = {grouped ? 800.0 : 1560.0, grouped ? 650.0 : 1170.0};
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
        *bm_out = code;
        *ks_out = ks;
      }
    }
  };
  consider(256, &rate256, grouped ? 258 : 256);
  consider(128, rate128, grouped ? 130 : 131);
  consider(64, rate64, 64);
  return best;
}

// 1 ... 32 tokens: the column kernel (one launch; every 32-column workgroup re-reads the m x K activations) against the stream kernel (K slices + reduce launch).
// Both are linear forms whose coefficients are GENERATED (qqq_rates.h, kQqqSmall; tools/fit_rates.py: least squares over the forced column / stream measurements of ten
// dispatch grids, 3 - 4 % mean error per form -- round 5's last re-measurement, taken with enough rotating weight copies that the Infinity Cache serves nobody):
//   column, per-channel: launch + the weights (c1 per MB) + the activations every workgroup reads again (c2 per 1e6 bytes, per round of 256 workgroups) + the serial depth
//   of a chip that is not full (c3 per 1000 k, scaled by the share of idle CUs: N = 3584, K = 18944 at decode 16.2 us where the bytes alone say 10.7);
//   per-group: bound by the re-quantiser -- g1 per 1000 k per round, flat up to 16 tokens, g2 for the second 16-token tile (growing as ((m - 16) / 16)^0.75) -- + the weights;
//   stream: a + b weight passes at 5 TB/s up to 16 tokens, a + c (m - 24) / 16 + b passes from 17; per mode.
static double column_small_estimate(int M, int N, int K, bool grouped) {
  const int wgs = N / 32, rounds = (wgs + device_cus() - 1) / device_cus();  // (CUs of THIS call: the device's, or the `sms` cap)
  const double mb = (double)N * K / 2.0e6;
  if (!grouped) {
    const double* c = kQqqSmall.col_pc;
    const double idle = wgs < device_cus() ? 1.0 - wgs / (double)device_cus() : 0.0;
    return c[0] + c[1] * mb + c[2] * 1e-6 * (double)K * M * rounds + c[3] * 1e-3 * K * idle;
  }
  const double* g = kQqqSmall.col_g;
  const double r = rounds == 1 ? 1.0 : 0.92 * rounds;
  double us = g[0] + g[1] * 1e-3 * K * r + g[3] * mb;
  if (M > 16) us += g[2] * 1e-3 * K * r * pow((M - 16) / 16.0, 0.75);
  return us;
}
static double stream_small_estimate(int M, int N, int K, bool grouped) {
  const double pb = (double)N * K / 2.0 / 5.0e6;
  const int gi = grouped ? 1 : 0;
  const double us = M <= 16 ? kQqqSmall.st16[gi][0] + kQqqSmall.st16[gi][1] * pb
                            : kQqqSmall.st32[gi][0] + kQqqSmall.st32[gi][1] * (M - 24) / 16.0 + kQqqSmall.st32[gi][2] * pb;
  if (!grouped) return us;
  // per-group a slice is also bound by its re-quantiser: 9.7 us + 2.1 us per 1000 k of the slice -- what a wide layer's unsplit strips pay
  // (N = 20480, K = 7168: 24.9 us against the column kernel's 21.1; profiles/r04_stream_ksplit_wide_n.txt).  The K split as make_plan picks it:
  const long long base = (long long)((N + 127) / 128) * (M <= 16 ? 1 : (M + 31) / 32);
  int ks = (int)((device_cus() + base / 2) / base);
  if (ks > 1 && base * ks > device_cus()) --ks;
  if (M <= 16 && base >= 128) ks = 1;
  const int cap = K / 64 / (M <= 16 ? 8 : 16);
  if (ks > cap) ks = cap;
  if (ks < 1) ks = 1;
  const double requant = 9.7 + 2.1e-3 * (double)K / ks;
  return us > requant ? us : requant;
}

// stream, 65 ... 256 tokens (two to four 64-token m-blocks) on layers up to ~40 MB: the loop, not the weight stream, sets the time: a + b per 64-k step of a slice, per
// round of 256 workgroups (x 1.2 from the second round on), + for a K split the slabs and the reduce launch (s0 + s1 per MB of int32 slabs); one set of rates per mode,
// GENERATED (qqq_rates.h, kQqqSmall.stmid: 137 points per mode, 3 - 4 % mean error; round 4's hand fit -- 8.65 us + 0.153 per step, x 1.235 per-group, 2 + 0.4 per MB --
// read 4 - 8 % low on the cold re-measurement).  Returns the time and the split it is reached with -- "fill 256 workgroups" (the rule for <= 64 tokens) splits
// short-K layers that are better left whole (N = 8192, K = 3072 at 128 tokens: 19.4 us in two slices, 15.7 unsplit).
static double stream_mid_estimate(int M, int N, int K, bool grouped, int ks_cap, int* ks_out) {
  const long long base = (long long)((N + 127) / 128) * ((M + 63) / 64);
  const int KS = K / 64;
  // The SPLIT is chosen by round 4's hand-fitted rates (right at 44 of its 48 points), the PRICE of that split comes from the generated ones: the forced stream
  // kernel's plan then does not depend on the table, so the tool that fits the table from measurements of that plan reaches a fixed point in one pass.
  double best_rule = 1e30, price = 1e30;
  *ks_out = 1;
  const double* c = kQqqSmall.stmid[grouped ? 1 : 0];
  for (int ks = 1; ks <= 8 && ks <= ks_cap; ++ks) {
    if (ks > 1 && KS / ks < 8) break;  // (8-wave bodies: at least a step per wave)
    const double rounds = (double)((base * ks + device_cus() - 1) / device_cus());
    const double steps = rounds * (rounds > 1.0 ? 1.2 : 1.0) * KS / ks, slab_mb = (double)M * N * 4.0 * ks / 1.0e6;
    const double rule = 8.65 + 0.153 * (grouped ? 1.235 : 1.0) * steps + (ks > 1 ? 2.0 + 0.4 * slab_mb : 0.0);
    if (rule < best_rule) {
      best_rule = rule;
      price = c[0] + c[1] * steps + (ks > 1 ? c[2] + c[3] * slab_mb : 0.0);
      *ks_out = ks;
    }
  }
  return price;
}

// stream: every 64-token m-block streams the whole weight matrix (the first from HBM, the others mostly from L2 /
// Infinity Cache), plus launch, LDS reduce and the separate split-K reduce launch
static double stream_estimate(int M, int N, int K, bool grouped, bool have_scratch = true, long long cap_rows = 1 << 30) {
  double per_block = (double)N * K / 2.0 / 5.0e6;  // the weight matrix at ~5 TB/s
  if (per_block < 2.5) per_block = 2.5;
  const int mblocks = (M + 63) / 64;
  if (mblocks >= 2 && mblocks <= 4 && (double)N * K / 2.0 / 5.0e6 < 8.0) {
    // (only splits the plan can realise: slabs need C, and ksplit x M rows of it)
    int ks, ks_cap = have_scratch ? (int)(cap_rows / M < 8 ? cap_rows / M : 8) : 1;
    return stream_mid_estimate(M, N, K, grouped, ks_cap < 1 ? 1 : ks_cap, &ks);
  }
  // measured: 2 / 3 / 4 m-blocks take 1.85 / 3.2 / 3.3 weight passes
  // (one m-block: 16 / 32 / 48 / 64 tokens measured at 0.6 / 0.8 / 1.0 / 1.2 -- the 16-token tiles of a block share the weights
  //  in registers but not the MFMA / VALU time)
  const double passes = (mblocks == 1) ? 0.4 + 0.0125 * M : (mblocks == 2) ? 1.85 : (mblocks == 3) ? 3.2 : 3.3 + 0.8 * (mblocks - 4);
  // fixed part: 2 and 3 m-blocks in one round of workgroups measured at 10.6 (3 blocks: 18.0 / 25.0 / 68 us at 1.7 / 4.5 / 17.8 us
  // per pass; 2 blocks: 16.3 / 18.9 / 43.5; profiles/r02_dispatch_check_handoff.txt); more than 256 workgroups even unsplit
  // (n = 11008: 86 strips x 3) is a second round
  double us = ((mblocks == 2 || mblocks == 3) ? 10.6 : 9.0 + 2.0 * mblocks) + per_block * passes;
  if ((long long)((N + 127) / 128) * mblocks > device_cus()) us *= 1.35;
  if (mblocks == 1 && M > 32) {
    // 33 ... 64 tokens: per token count a line in the weight bytes (no floor: the 8 MB layers sit ON the line), with a step where the fourth 16-token tile
    // starts (49 tokens); one form per mode, coefficients generated (qqq_rates.h, kQqqSmall.st64: 68 points per mode, 3 % mean error)
    const double pb = (double)N * K / 2.0 / 5.0e6;
    const double four = M > 48 ? 1.0 : 0.0;
    const double* f = kQqqSmall.st64[grouped ? 1 : 0];
    return f[0] + f[1] * M + f[2] * four + pb * (f[3] + f[4] * M + f[5] * four);
  }
  return grouped ? us * 1.15 : us;
}

// panel: all tokens of an m-block (16 ... 128) x bn-column strips x K slices.  Priced from the GENERATED table qqq_rates.h (tools/fit_rates.py: one
// linear form per (strip shape, m-block, mode), least squares over every forced panel variant of profiles/r05_dispatch_check_*.txt -- 2600 measurements, 2 ... 5 % mean
// error per group): us = rounds x (a + c [split] + d (slices - 2) + b x stages per workgroup).  Round 5 replaced the hand-fitted constants of rounds 2 - 4 here (they
// were 10 ... 35 % high once the uneven K slices had shortened the hand-off).
static double panel_estimate(int M, int N, int K, bool grouped, bool have_scratch, long long cap_rows, long long cap_tickets,
                             int* bn_out, int* ks_out, int* cw_out, int* mt_out = nullptr) {
  if (mt_out) *mt_out = 0;  // 0: the m-block the token count implies (16 / 32 / 64 / 128 rows)
  const int mt = (M <= 16) ? 1 : (M <= 32) ? 2 : (M <= 64) ? 4 : 8;
  const int mti = mt == 1 ? 0 : mt == 2 ? 1 : mt == 4 ? 2 : 3;
  const int rows = 16 * mt;
  const long long mblocks = (M + rows - 1) / rows;
  const int NST = (K / 64 + 1) / 2;
  double best = 1e30;
  for (int shape = 0; shape < 3; ++shape) {  // 128-column strips; 256-column strips; 256-column strips with 64 columns per wave (128-token m-blocks)
    if (shape == 2 && mt != 8) continue;
    const int bn = shape == 0 ? 128 : 256;
    const QqqPanelRate& r = kQqqPanelRates[shape][mti][grouped ? 1 : 0];
    if (r.b <= 0.0) continue;
    const long long tl = mblocks * ((N + bn - 1) / bn);
    for (int ks = 1; ks <= 4; ++ks) {
      if (ks > 1 && (!have_scratch || 2 * tl > cap_tickets || tl * rows * bn * (ks - 1) > cap_rows * (long long)N || ks > NST / 4)) break;  // slots: tiles x (ks-1) x rows x bn ints inside C
      const double rounds = (double)((tl * ks + device_cus() - 1) / device_cus());
      const double us = rounds * (r.a + (ks > 1 ? r.c : 0.0) + (ks > 2 ? r.d * (ks - 2) : 0.0) + r.b * (double)NST / ks);
      if (us < best) {
        best = us;
        *bn_out = bn;
        *ks_out = ks;
        *cw_out = shape == 2 ? 2 : 1;
      }
    }
  }
  // More than 64 tokens as SEVERAL 64-token m-blocks (round 5, profiles/r05_panel_feeder.txt): on layers whose 128-token tiles leave CUs idle or hold few
  // stages each, twice the workgroups of half the size finish sooner although every m-block streams the weights again (from L2 / the Infinity Cache) --
  // 4096 x 4096 at 128 / 256 / 512 tokens 13.1 / 15.2 / 18.4 us against 15.2 / 17.7 / 19.4, 4096 x 11008 and 11008 x 4096 at 128 tokens 18.3 / 18.5 against
  // 21.5 / 20.0, 8192 x 8192 21.3 against 22.1; not on the BASELINE layer (40.1 vs 35.4: the weights come from HBM twice) and not beyond one round of
  // workgroups (11008 x 4096 at 256 tokens 27.7 vs 23.3).  Priced with the 64-token form above; one round only.
  // (the form holds on narrow layers as well: 74 single-round points with N < 4096 in profiles/r05_dispatch_check_*.txt, mean error 3.9 %)
  if (mt == 8 && mt_out) {
    const long long mb4 = (M + 63) / 64, tl = mb4 * ((N + 127) / 128);
    for (int ks = 1; ks <= 4; ++ks) {
      if (tl * ks > device_cus()) break;
      if (ks > 1 && (!have_scratch || 2 * tl > cap_tickets || tl * 64 * 128 * (ks - 1) > cap_rows * (long long)N || ks > NST / 4)) break;
      // (rates GENERATED -- qqq_rates.h, kQqqSmall.panel64, tools/fit_rates.py: the single-round points of the column panel64 in profiles/r05_dispatch_check_*.txt,
      //  165 per mode, 2.5 - 2.7 % mean error: launch + fill + epilogue unsplit / split, us per stage, every further m-block a share of a weight pass at bw MB/us)
      const double* q = kQqqSmall.panel64[grouped ? 1 : 0];
      const double stage_us = ((double)NST / ks) * q[2];
      const double bytes_us = (1.0 + q[3] * (double)(mb4 - 1)) * ((double)N * K / 2.0e6) / q[4];
      const double us = (ks == 1 ? q[0] : q[1]) + (stage_us > bytes_us ? stage_us : bytes_us);
      if (us < best) {
        best = us;
        *bn_out = 128;
        *ks_out = ks;
        *cw_out = 1;
        *mt_out = 4;
      }
    }
  }
  return best;
}

// wide: three tile shapes, one tile per CU and round.  Fitted on one box after the LDS-DMA staging
// (profiles/r03_dispatch_check_wide3.txt): time = 3.7 + rounds * (fixed + hand-off + stages * t_stage * load), with
//   256 x 256 (mt 16, bn 256): fixed 12 us (first operands from HBM with every CU in its prologue at once ~4, epilogue ~7),
//                              1.25 us per 128-k stage (per-group 1.635: the re-quantiser of a lone wave is issue-bound);
//   256 x 128 (mt 16, bn 128): fixed 7, 0.70 (0.96) per stage: a weight operand still feeds 256 tokens, twice the LDS traffic
//                              per MFMA -- the shape that fills the chip from ~600 tokens, and per-group on 4096-wide layers;
//   128 x 256 (mt 8,  bn 256): fixed 7, 0.725 (1.17): twice the unpack / re-quantise work per MFMA.  (Round 4, cold A/B on six layer shapes at
//                              640 ... 2048 tokens, profiles/r04_wide_w8_vs_w128.txt: 256 x 128 is 1 ... 4.5 % ahead of 128 x 256 per-channel wherever the
//                              tile counts do not decide -- 0.72 / 0.71 had it the other way round.)
// Two K slices (256-token tiles): hand-off 15 us per 256 KiB of partial tile (kept in the XCD's L2 when its slices share one -- round 4 --,
// folded by the last arrival; 20 us written through).
// (w8: the call has the expanded int8 weights.  The loop is then the per-channel loop minus its unpack, with two more 16-byte
//  loads per step: priced like a per-channel call (load rule) with rates of its own -- the third column of kQqqWideRates, fitted from profiles/r06_w8_dispatch_check_main.txt.)
static double wide_estimate(int M, int N, int K, bool grouped, bool have_scratch, long long cap_rows, long long cap_tickets, int* ks_out,
                            int* mt_out, int* bn_out, bool w8 = false) {
  *ks_out = 1;
  *mt_out = 16;
  *bn_out = 256;
  if ((long long)N * K / (w8 ? 1 : 2) >= (1ll << 32) || (K % 128) != 0) return 1e30;  // 32-bit offsets into the weights; whole stages
  if (w8) grouped = false;
  const int NST = K / 128;
  double best = 1e30;
  for (int shape = 0; shape < 3; ++shape) {
    const int mt = shape == 2 ? 8 : 16, bn = shape == 1 ? 128 : 256;
    const int rows = 16 * mt;
    const long long tl = (long long)((M + rows - 1) / rows) * ((N + bn - 1) / bn);
    // (round 5: fixed / t_stage / hand-off per shape and mode from the GENERATED table qqq_rates.h -- tools/fit_rates.py, least squares over every forced wide variant of
    //  profiles/r05_dispatch_check_*.txt, 2.3 ... 3.9 % mean error per group; the hand-fitted values above -- 12 / 7 us, 1.25 / 0.70 / 0.725 us per stage, 15 us per
    //  256 KiB -- are the history: the uneven K slices took the hand-off to 11-12.6)
    const QqqWideRate& wr = kQqqWideRates[shape][w8 ? 2 : grouped ? 1 : 0];  // (round 6: calls that have expanded weights are priced from a fit of their own)
    const double t_stage = wr.t_stage;
    const double fixed = wr.fixed;
    for (int ks = 1; ks <= (mt == 16 ? 2 : 1); ++ks) {
      // one slot of rows x bn ints per tile and depositing slice inside C, two ticket words per tile
      if (ks > 1 && (!have_scratch || 2 * tl > cap_tickets || tl * rows * bn * (ks - 1) > cap_rows * (long long)N || ks > NST / 4)) break;
      // rounds: workgroups of later rounds start as CUs free up, but the XCDs' queues drain unevenly -- close to the ceiling of
      // the ratio (288-344 tiles measured 1.7-1.8 rounds, 688 tiles 2.8-3).  A partly filled single round runs each tile
      // faster: the part is power-limited (half the CUs busy: 0.8 of the full-chip stage time, 0.85 per-group; then quadratic)
      const double x = (double)(tl * ks) / (double)device_cus();
      const double cx = (double)((tl * ks + device_cus() - 1) / device_cus());
      const double rounds = x <= 1.0 ? 1.0 : cx - 0.3 * (cx - x);
      const double lo = grouped ? 0.85 : 0.80, rel = x <= 0.5 ? 0.0 : (x - 0.5) / 0.5;
      const double load = x <= 1.0 ? lo + (1.0 - lo) * rel * rel : 1.0;
      // (15 us per 256 KiB of partial tile since the deposits stay in the XCD's L2, 20-23 written through: profiles/r04_wide_xcd_local_deposits.txt)
      const double handoff = ks > 1 ? wr.handoff * (double)(rows * bn) / 65536.0 : 0.0;
      const double us = 3.7 + rounds * (fixed + handoff + ((double)NST / ks) * t_stage * load);
      if (us < best) {
        best = us;
        *ks_out = ks;
        *mt_out = mt;
        *bn_out = bn;
      }
    }
  }
  return best;
}

// panel, split K: stages the last slice gets on top of an even share (tune.skew = 0).  Measured (profiles/r05_uneven_k_slices.txt: N = 8192, K = 21760 at
// 64 / 128 / 256 tokens, both modes, and 4096 x 4096 at 128 / 256): the best skew makes the last slice's extra loop time -- skew stages, plus the
// skew / (ks - 1) stages every other slice is shorter by -- about the latency of a deposit (write-through drain + publish: ~1.6 us, + ~1.2 us per
// 64 KiB of partial tile): 4 stages at 128 tokens (37.3 -> 36.1 us; per-group 3: 45.5 -> 43.4), 3 in two slices (256 tokens: 56.9 -> 54.4).  Less than
// that is slower than even slices (the last slice arrives last but still waits), more only lengthens the longest slice.
static int panel_auto_skew(int mt, int bn, bool grouped, int NST, int ksplit) {
  (void)NST;
  const double t_stage = ((mt == 8 && bn == 128) ? 0.506 : 0.13 + 0.042 * mt) * (bn == 256 ? 1.9 : 1.0) * (grouped ? (mt == 8 ? 1.45 : 1.6) : 1.0);
  const double latency = 1.6 + 1.2 * (16.0 * mt * bn) / 16384.0;
  const int sk = (int)(latency / (t_stage * ksplit / (ksplit - 1.0)) + 0.5);
  return sk < 1 ? 1 : sk;
}
// stream, split K: how the slices meet (tune.fused = 0) -- 2 = slabs + a reduce launch, 3 = in-launch through arrival-order slots.  Not measured yet: 2.
static int stream_auto_fused(int M, int N, int K, bool grouped, int ksplit) {
  (void)M; (void)N; (void)K; (void)grouped; (void)ksplit;
  return 2;
}
// ... and the skew of the slot protocol, in 64-k steps: the slices stream the matrix together in about N K / 2 / 5 TB/s, a deposit (8 ... 32 KiB written
// through, drained, counted) takes ~2 us
static int stream_auto_skew(int N, int K, int ksplit) {
  const double pass_us = (double)N * K / 2.0 / 5.0e6;
  const double t_step = pass_us * ksplit / (K / 64.0);
  const int sk = (int)(2.0 / (t_step * ksplit / (ksplit - 1.0)) + 0.5);
  return sk < 1 ? 1 : sk;
}
// wide, split K: the same rule with the wide kernel's deposits (a 256 x 256 partial tile goes through the LDS transposition and out as 256 KiB of
// row-major int32: ~20 us from the depositor's last MFMA to "complete", half of that for the 128-column tiles) and stage times (wide_estimate).
// Measured (profiles/r05_uneven_k_slices_wide.txt): N = 8192, K = 21760 per-group at 1024 tokens 173.1 -> 165.9 us (skew 6; 4: 167.6, 8: 167.0), per-channel
// two slices of 256 x 256 140.3 -> 133.2 (8); 256 x 128 tiles in two slices at 384 / 512 tokens 80.8 -> 76.4 / 85.6 -> 82.2 (6), per-group 512: 102.8 -> 98.6 (3-6);
// Llama-2-7B down_proj (4096 x 11008) at 1024 tokens 50.1 -> 48.6, per-group 62.3 -> 58.1 (3).
static int wide_auto_skew(int mt, int bn, bool grouped, int NST, int ksplit, bool w8 = false) {
  (void)NST;
  if (w8) grouped = false;
  const double t_stage = kQqqWideRates[(mt == 16 && bn == 256) ? 0 : (mt == 16) ? 1 : 2][w8 ? 2 : grouped ? 1 : 0].t_stage;
  const double latency = 20.0 * (16.0 * mt * bn) / 65536.0;
  const int sk = (int)(latency / (t_stage * ksplit / (ksplit - 1.0)) + 0.5);
  return sk < 1 ? 1 : sk;
}

// The dispatch decision of one call, as plain data (pure host logic: also exported as qqq_w4a8_plan so
// that it can be inspected and tested without a GPU).
struct Plan {
  int kernel;  // 1 stream, 2 tiled, 3 column, 4 panel, 5 wide
  int ksplit;
  int fused;   // stream: 1 / 3 in-launch, 2 separate reduce.  tiled: 1 in-launch slots, 2 slabs + reduce
  int mt, waves, pf;      // stream
  int bm, stages, nslots, pw; // tiled
  int chain;                  // wide: 1 = persistent tile walk (one workgroup per CU walks its run of tiles)
  int skew;                   // panel: extra 128-k stages of the last K slice
  int w8;                     // wide: 1 = the loop reads the expanded int8 weights (the call has them)
  int exch;                   // wide, two K slices of 256-column tiles: 1 = exchange hand-off (each slice finishes one row half), even slices
};

// (t.w8: 1 = the call has the expanded int8 weights of its layer -- gemm_ex2 sets it from its W8 argument --, -1 = ignore them)
static Plan make_plan(const int M, const int N, const int K, const bool grouped, const int max_par,
                      const bool have_C, const bool have_ws, qqq_tune_t t, double* est_out = nullptr) {
  const bool have_w8 = t.w8 > 0 && (long long)N * K < (1ll << 32) && (K % 128) == 0;
  Plan pl;
  memset(&pl, 0, sizeof(pl));
  double est = -1.0;  // the chosen family's modelled time (us) when the choice is the models' (automatic dispatch); -1 otherwise
  const long long cap_rows = (long long)(max_par > 0 ? max_par : 0) * 64;  // rows of C we may use
  const bool have_scratch = have_C && cap_rows > 0;
  const void* workspace = have_ws ? reinterpret_cast<const void*>(1) : nullptr;

  // ---- kernel choice (measured on MI355X, profiles/) ----
  // decode (m <= 16): the "column" kernel (32-column workgroups over all of K, no split-K, no reduce launch);
  // m <= 128: the HBM-bound "stream" kernel (128-column strips x K slices); above, LDS tiles.
  int kernel = t.kernel;
  const bool column_ok = (N % 64) == 0 && (K % 64) == 0;
  if (kernel == 0) {
    // measured (profiles/r01_tune_decode.txt, r02_decode_sweep.txt): every column workgroup re-reads the m x K
    // activations (per-lane 16-byte loads of 16 rows), so beyond m = 8 it only wins while that stays cheap -- but then
    // up to 32 tokens (two 16-token tiles per wave), where it saves the stream kernel's reduce launch: the two small cost models above
    // (beyond 512 column workgroups -- two rounds of the chip -- the stream kernel's one round of K slices wins even at decode:
    // N = 28672, K = 8192: 23.0 vs 25.3 us per-channel, 29.1 vs 30.4 per-group, profiles/r04_dispatch_check_shapes_before.txt)
    // (narrow layers -- the k / v projections of grouped-query attention, N = 1024 / 512: 32 / 16 column workgroups, 6.3 vs 8.8 us and 6.2 vs 9.4 us
    // at decode, profiles/r04_dispatch_check_merged.txt, r04_dispatch_check_qwen_mistral.txt; below that not measured)
    // (per-group up to 16 tokens the cap of three rounds holds as at decode: N = 22016, K = 4096 at 16 tokens 15.0 vs 17.3 us)
    // Per-group the column kernel is bound by its re-quantiser (time ~ K per round), so on long-K layers the stream kernel's K slices win even
    // at decode (N = 3584, K = 18944: 15.4 vs 19.4 us; N = 4096, K = 14336: 13.5 vs 15.8): the two small models decide from one token on.
    // (a tie goes to the column kernel -- one launch instead of two; 2 % is where the measured regret of the rule is smallest: 0.27 % mean over 238 points against 0.32 % without)
    const bool col_cheaper = column_small_estimate(M, N, K, grouped) < 1.02 * stream_small_estimate(M, N, K, grouped);
    const bool column = column_ok && N / 32 >= 16 && N / 32 <= ((M <= 8 || (grouped && M <= 16)) ? 768 : 512) &&
                        (grouped ? (M <= 32 && col_cheaper) : (M <= 8 || (M <= 32 && col_cheaper)));
    if (column) kernel = 3;
    else kernel = (M <= 128 || (K % 128) != 0) ? 1 : 2;
    if (column_ok && M <= 32) est = column ? column_small_estimate(M, N, K, grouped) : stream_small_estimate(M, N, K, grouped);
    // 17 ... 32 tokens: the panel kernel's 32-token m-blocks are a third candidate -- on very wide layers 256-column strips in two or three K slices beat both
    // (N = 20480, K = 7168 at 32 tokens: 21.7 us against 27.2 / 32.6; N = 28672 / 29568: 10 - 12 %; 10 of the 192 measured points, profiles/r05_dispatch_check_*.txt).
    // It has to be clearly ahead (5 %: the three models are each good to 3 - 4 %).  (Up to 16 tokens it won 3 of 116 measured points, and the one time the models
    // picked it there -- N = 20480, K = 7168 -- the clock said 20.2 us against the stream kernel's 16.6: not a candidate.)
    if (column_ok && M > 16 && M <= 32 && t.bm == 0 && t.mt == 0 && t.ksplit <= 0) {
      const long long cap_tk = have_ws ? (long long)(N / 128) * (max_par > 0 ? max_par : 0) : 0;
      int pbn = 128, pks = 1, pcw = 1;
      const double e_panel = panel_estimate(M, N, K, grouped, have_scratch && have_ws, cap_rows, cap_tk, &pbn, &pks, &pcw);
      if (e_panel < 0.95 * est) {
        kernel = 4;
        est = e_panel;
        t.bm = pbn;
        t.ksplit = pks;
      }
    }
    // Above the decode regime the family is picked by the three cost models.  The panel kernel is also the MFMA path
    // with LDS-shared activations for K % 128 == 64 at any m (the tiled kernel needs 128-k blocks).
    if (column_ok && !column && M > 32) {
      const long long cap_tk = have_ws ? (long long)(N / 128) * (max_par > 0 ? max_par : 0) : 0;
      int pbn = 128, pks = 1, pcw = 1, pmt = 0;
      const double e_panel = ((long long)(M + 127) / 128 <= 65535) ? panel_estimate(M, N, K, grouped, have_scratch && have_ws, cap_rows, cap_tk, &pbn, &pks, &pcw, &pmt) : 1e30;
      // (round 6: above 64 tokens the stream kernel is a candidate only where its GENERATED form prices it -- 2 ... 4 m-blocks on layers up to ~40 MB, kQqqSmall.stmid.  The
      //  regime beyond -- larger layers, five m-blocks and more -- was the last hand-fitted branch of the dispatcher (stream_estimate below its first return); in the 48 points of it
      //  that the committed grids measured the stream kernel is never more than 3 % ahead of the best panel / wide shape, and the plan never chose it: out of the automatic path.
      //  tune.kernel = 1 still runs it; qqq_w4a8_model_us still prices it.)
      const int s_mblocks = (M + 63) / 64;
      const bool stream_candidate = s_mblocks == 1 || (s_mblocks <= 4 && (double)N * K / 2.0 / 5.0e6 < 8.0);
      const double e_stream = (M <= 256 && stream_candidate) ? stream_estimate(M, N, K, grouped, have_scratch, cap_rows) : 1e30;
      // (the tiled family -- round 1's LDS-tiled 32x32x32 kernel -- is no longer a candidate of the automatic dispatch: in the 903 measured
      // dispatch points of round 4 it never won one (M = 4096: 585.9 vs 449.2 us).  It stays reachable through tune.kernel = 2 -- the
      // differential fuzzers' independent reference -- and as the fallback for packed weights beyond 4 GB, where the wide kernel's 32-bit
      // offsets end.)
      int wks = 1, wmt = 16, wbn = 256;
      const double e_wide = (M > 256) ? wide_estimate(M, N, K, grouped, have_scratch && have_ws, cap_rows, cap_tk, &wks, &wmt, &wbn, have_w8) : 1e30;
      est = e_wide < e_panel ? e_wide : e_panel;
      if (e_stream < est) est = e_stream;
      if (e_wide < e_panel && e_wide < e_stream) {
        kernel = 5;
        // (the K split was costed for the model's own tile shape: a caller who pins mt / bm gets one slice unless it asks)
        if (t.mt == 0 && t.bm == 0) {
          t.mt = wmt;
          t.bm = wbn;
          if (t.ksplit <= 0) t.ksplit = wks;
        }
      } else if (e_panel <= e_stream) {
        kernel = 4;
        if (t.bm == 0 && t.pw == 0 && t.mt == 0 && pcw == 2) t.pw = 2;
        if (t.bm == 0 && t.mt == 0 && pmt) t.mt = pmt;  // several 64-token m-blocks instead of 128-token ones
        if (t.bm == 0) t.bm = pbn;
        if (t.ksplit <= 0) t.ksplit = pks;
      } else {
        kernel = 1;
      }
    }
  }
  if (est_out) *est_out = est;
  if ((kernel == 3 || kernel == 4 || kernel == 5) && !column_ok) kernel = 1;
  if (kernel == 5 && (K % 128) != 0) kernel = 4;                        // whole 128-k stages only
  if (kernel == 5 && (long long)N * K / 2 >= (1ll << 32)) kernel = 2;  // 32-bit offsets into the packed weights
  if (kernel == 2 && (K % 128) != 0) kernel = 1;
  pl.kernel = kernel;
  int ksplit = 1;

  if (kernel == 5) {
    // wide: 256 (mt = 8: 128) tokens x 256 columns per workgroup, 4 waves with 512 registers each; in-launch split-K with
    // one slot of C per depositing slice (row-major partial tiles) and two ticket words per tile
    pl.mt = (t.mt == 8) ? 8 : 16;                         // 16-token m-tiles per workgroup: 256- or 128-token tiles
    pl.bm = (t.bm == 128 && pl.mt == 16) ? 128 : 256;     // columns per workgroup: 64 or (256-token tiles only) 32 per wave
    // (128 x 128 tiles compile from the same template and were measured: 39 us at M=128 against the panel kernel's 36, 4 % ahead
    // of it only at 160-256 tokens in two K slices -- not instantiated; profiles/r03_wide_128x128.txt)
    pl.stages = 1;                                        // activation lead: the LDS-DMA of a stage is issued a full stage ahead
    // weight ring in 64-k steps.  Round 3 had per-channel 4 / per-group 8 (8 measured 1.5 % ahead per-group THEN).  Round 5, same A/B with the variants repeated in the
    // interleaved list (profiles/r05_wide_ring_depth.txt): for the 256 x 256 tiles it is now the other way round -- per-channel 8: 449.8 vs 456.2 us at 4096 tokens,
    // per-group 4: 584.1 vs 594.4 -- and the compiled code says why: of the two instantiations of a mode the slower one carries the register spills (69 / 27 scratch
    // instructions against 3, all in the epilogue, none in the loop: tools/code_object.py); the ring depth moves hipcc's allocation at the seam between loop and epilogue.
    // The other shapes (no spills either way) measure level and keep their depths.
    const bool big_tile = (pl.mt == 16 && pl.bm == 256);
    pl.w8 = have_w8 ? 1 : 0;
    pl.pf = pl.w8 ? 4 : (t.pf == 8 || t.pf == 4) ? t.pf : (big_tile ? (grouped ? 4 : 8) : (grouped ? 8 : 4));
    if (QQQ_WIDE_DWORD != 0 && !pl.w8) pl.pf = 4;  // (dword weight loads: 8 loads per step -- a ring of 8 steps would pass vmcnt's 63)
    pl.pw = (t.pw == 4 || t.pw == 8 || t.pw == 16 || t.pw == 32) ? t.pw : 8;
    const int rows = 16 * pl.mt;
    const long long tl = (long long)((M + rows - 1) / rows) * ((N + pl.bm - 1) / pl.bm);
    ksplit = t.ksplit > 0 ? t.ksplit : 1;
    ksplit = clampi(ksplit, 1, (K / 128) / 4 > 0 ? (K / 128) / 4 : 1);
    if (ksplit > 255) ksplit = 255;  // the arrival word's ticket field is 8 bits (the XCC nibbles sit above it; they are used up to 6 slices)
    if (!have_scratch || workspace == nullptr) ksplit = 1;
    const long long cap_tk = (long long)(N / 128) * (max_par > 0 ? max_par : 0);
    if (2 * tl > cap_tk) ksplit = 1;
    while (ksplit > 1 && tl * rows * pl.bm * (ksplit - 1) > cap_rows * (long long)N) --ksplit;
    pl.ksplit = ksplit;
    pl.fused = 1;
    pl.skew = 0;
    // two slices of 256-column tiles: the exchange hand-off (qqq_wide.hip.h) on request -- tune.fused bit 64.  Measured level with the classic fold over uneven
    // slices or up to 2 % behind it (N = 8192, K = 21760 at 768 / 1024 tokens, both modes, and 4096 x 11008: profiles/r06_wide_exchange_handoff.txt): either way the
    // tile's partial sums -- 33.5 MB chip-wide at 1024 tokens -- cross the fabric once out and once back at the chip's write bandwidth, which is what the hand-off costs.
    pl.exch = (ksplit == 2 && pl.bm == 256 && t.skew <= 0 && (t.fused & 64)) ? 1 : 0;
    if (ksplit > 1 && !pl.exch) {  // uneven K slices (tune.skew: -1 never, 0 automatic, else stages): every slice keeps at least 4 stages
      int sk = t.skew > 0 ? t.skew : (t.skew == 0 ? wide_auto_skew(pl.mt, pl.bm, grouped, K / 128, ksplit, pl.w8 != 0) : 0);
      const int room = K / 128 - 4 * ksplit;
      if (sk > room) sk = room;
      if (sk > 63) sk = 63;
      pl.skew = sk > 0 ? sk : 0;
    }
    // the persistent tile walk (t.glds: 1 = never, 2 = whenever it applies, 0 = automatic); its ring depth is the mode's default
    pl.chain = (t.glds != 1 && wide_chain_ok(M, N, K, rows, pl.bm, ksplit) && (t.glds == 2 || wide_chain_pays(tl, K, pl.mt, pl.bm))) ? 1 : 0;
    if (pl.chain && pl.w8 && !big_tile) pl.chain = 0;  // (the walk over expanded weights is instantiated for 256 x 256 tiles only)
    if (pl.chain) pl.pf = pl.w8 ? 4 : (grouped && QQQ_WIDE_DWORD == 0) ? 8 : 4;
    return pl;
  }

  if (kernel == 4) {
    // panel: all tokens of an m-block (16*mt <= 128) x bn columns x a K slice per workgroup; in-launch split-K with one
    // slot of C per depositing slice and two ticket words per (m-block, strip) tile
    int mt = (t.mt == 1 || t.mt == 2 || t.mt == 4 || t.mt == 8) ? t.mt : (M <= 16 ? 1 : M <= 32 ? 2 : M <= 64 ? 4 : 8);
    const int bn = (t.bm == 256) ? 256 : 128;
    const int waves = (bn == 256) ? 8 : (t.waves == 4 ? 4 : 8);
    const int rows = 16 * mt;
    const long long mblocks = (M + rows - 1) / rows, strips = (N + bn - 1) / bn;
    const int NST = (K / 64 + 1) / 2;
    ksplit = t.ksplit;
    if (ksplit <= 0) ksplit = clampi((int)(device_cus() / (strips * mblocks)), 1, 4);  // never more than one round of workgroups (on the CUs this call may use)
    ksplit = clampi(ksplit, 1, NST / 4 > 0 ? NST / 4 : 1);
    if (!have_scratch || workspace == nullptr) ksplit = 1;
    if (ksplit > 1) {
      const long long cap_tk = (long long)(N / 128) * (max_par > 0 ? max_par : 0);
      if (2 * mblocks * strips > cap_tk) ksplit = 1;
      // one slot of rows x bn ints per tile and depositing slice, all inside the max_par*64 x n ints of C (bn-wide strips
      // may overhang n)
      while (ksplit > 1 && mblocks * strips * rows * bn * (ksplit - 1) > cap_rows * (long long)N) --ksplit;
    }
    pl.mt = mt;
    pl.bm = bn;
    pl.waves = waves;
    const bool cw2 = (t.pw == 2 && bn == 256 && mt == 8);
    pl.pf = (t.pf == 2 || t.pf == 3 || t.pf == 8) ? t.pf : (t.pf == 0 && bn == 256 && !cw2 ? 3 : 4);  // weight prefetch depth in stages
    if (pl.pf == 8 && mt > 4) pl.pf = 4;
    // activation prefetch depth in stages: 2 with the 4-deep weight ring (the 64-column shape: 14 registers of slack, no spill
    // with the four-buffer, barrier-every-other-stage ring, profiles/r02_panel_cw2.txt; the 32-column shapes: 1-4 % faster than
    // depth 4 on every shape measured, profiles/r02_panel_prefetch_depth.txt)
    pl.stages = (pl.pf == 4 && t.stages != 4) ? 2 : pl.pf;
    // 32-column sets per wave: 2 = 4 waves x 64 columns x 2 k-groups for the 256-column, 128-token shape
    pl.pw = (cw2 && pl.pf >= 3 && pl.pf <= 4 && (pl.stages == pl.pf || (pl.pf == 4 && pl.stages == 2))) ? 2 : 1;
    pl.waves = waves;
    pl.ksplit = ksplit;
    pl.fused = 1;
    // uneven K slices (tune.skew: -1 never, 0 automatic, else stages): every slice keeps at least 4 stages
    pl.skew = 0;
    if (ksplit > 1) {
      int sk = t.skew > 0 ? t.skew : (t.skew == 0 ? panel_auto_skew(mt, bn, grouped, NST, ksplit) : 0);
      const int room = NST - 4 * ksplit;
      if (sk > room) sk = room;
      if (sk > 63) sk = 63;
      pl.skew = sk > 0 ? sk : 0;
    }
    return pl;
  }

  if (kernel == 3) {
    const int mt = (t.mt >= 1 && t.mt <= 2) ? t.mt : (M <= 16 ? 1 : 2);
    const int KS = K / 64;
    ksplit = t.ksplit > 0 ? t.ksplit : 1;  // a second launch costs more than idle CUs save (measured)
    ksplit = clampi(ksplit, 1, KS);
    if (!have_scratch) ksplit = 1;
    if (ksplit > 1 && (long long)ksplit * M > cap_rows) ksplit = (int)(cap_rows / M);
    if (ksplit < 1) ksplit = 1;
    pl.mt = mt;
    // sixteen waves per workgroup (the waves split K inside the workgroup) where the re-quantiser binds: per-group up to 8 tokens -2 ... -6 % on five layer shapes
    // (BASELINE layer at decode 23.5 -> 22.05 us); per-channel mixed (+4 % there), from 9 tokens level: eight (profiles/r05_column_16_waves.txt)
    pl.waves = mt == 1 && (t.waves == 16 || (t.waves == 0 && grouped && M <= 8)) ? 16 : 8;
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
      ksplit = (int)((device_cus() + base / 2) / base);
      // a 257th workgroup is a second round (n = 11008: 86 strips x 3 slices) -- for the 4-wave bodies of <= 16 tokens as well (N = 7168, K = 20480:
      // 56 strips x 5 slices 25.5 us, x 4 slices 18.9; profiles/r04_stream_ksplit_wide_n.txt)
      if (ksplit > 1 && base * ksplit > device_cus()) --ksplit;
      // up to 16 tokens (4-wave bodies): 128 workgroups or more already pull the weights at the HBM's pace, a second K slice only adds the reduce
      // launch (N = 18944, K = 3584 at 16 tokens: 10.9 us unsplit, 14.9 in two slices; N = 16384, K = 4096: 11.5 / 12.7; profiles/r04_stream_ksplit_wide_n.txt)
      if (mt == 1 && base >= 128) ksplit = 1;
      ksplit = clampi(ksplit, 1, KS / (2 * waves) > 0 ? KS / (2 * waves) : 1);
      // 65 ... 256 tokens on layers up to ~40 MB: the split the loop model is fastest with (see stream_mid_estimate)
      if (mt == 4 && mblocks >= 2 && mblocks <= 4 && (double)N * K / 2.0 / 5.0e6 < 8.0) stream_mid_estimate(M, N, K, grouped, KS / (2 * waves) > 0 ? KS / (2 * waves) : 1, &ksplit);
    }
    ksplit = clampi(ksplit, 1, KS);
    if (!have_scratch) ksplit = 1;
    if (ksplit > 1 && (long long)ksplit * M > cap_rows) ksplit = (int)(cap_rows / M);
    if (ksplit < 1) ksplit = 1;
    int fused = t.fused & 3;  // (bits 2.. are the in-launch hand-off switches of the other families)
    if (fused == 0) fused = stream_auto_fused(M, N, K, grouped, ksplit);
    // tickets: one int per (m-block, strip) for the fenced fold over the slabs (1), two for the slot protocol (3); the reference guarantees
    // n/128*max_par ints.  Slots (3): (ksplit - 1) tiles of 16*mt x 128 ints per (m-block, strip), inside the rows of C we may use.
    if (fused == 1 && (workspace == nullptr || (long long)mblocks * strips > (long long)(N / 128) * max_par)) fused = 2;
    if (fused == 3 && (workspace == nullptr || 2ll * mblocks * strips > (long long)(N / 128) * max_par ||
                       (long long)mblocks * strips * (ksplit - 1) * (16ll * mt * 128) > cap_rows * (long long)N))
      fused = 2;
    pl.skew = 0;
    if (fused == 3 && ksplit > 1) {  // uneven slices, in 64-k steps (tune.skew: -1 never, 0 automatic); every slice keeps two steps per wave
      int sk = t.skew > 0 ? t.skew : (t.skew == 0 ? stream_auto_skew(N, K, ksplit) : 0);
      const int room = KS - 2 * waves * ksplit;
      if (sk > room) sk = room;
      if (sk > 255) sk = 255;
      pl.skew = sk > 0 ? sk : 0;
    }
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
    if (!have_scratch || ks < 2 || (t.fused & 3) == 2) return 0;
    const long long tl = (long long)((M + rows - 1) / rows) * strips;
    long long S = cap_ints / (tl * rows * 256);
    if (S > ks - 1) S = ks - 1;
    while (S > 0 && tl * (1 + S) > cap_tickets) --S;
    return (int)S;
  };
  int bm = t.bm;
  if (bm != 64 && bm != 128 && bm != 256 && bm != 258 && bm != 259 && bm != 130 && bm != 131) {
    int best_ks = 1;
    (void)tiled_estimate(M, N, K, grouped, have_scratch, cap_rows, cap_tickets, (t.fused & 3) == 2, &bm, &best_ks);
    if (t.ksplit <= 0) t.ksplit = best_ks;
  }
  // glds: 2 = register-staged, 1 = LDS-DMA ring with `stages` buffers; auto: the DMA ring pays at the
  // 8-wave 256-row tile, register staging is faster for the 4-wave tiles (measured)
  int stages;
  if (t.glds == 2) stages = 0;
  else if (t.glds == 1) stages = ((t.stages >= 2 && t.stages <= 4) || (t.stages == 5 && bm != 64 && bm != 128) || ((t.stages == 6 || t.stages == 7) && bm == 256)) ? t.stages : (bm >= 256 ? 3 : 4);
  else stages = (bm == 256) ? 7 : (bm == 258) ? 2 : (bm == 130) ? 4 : 0;  // measured best per shape (profiles/r01_tune_sweep_*.txt, r02_tiled_ns7.txt)
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

// ---- M split (rows are independent).  A token count one past a whole number of rounds of 256 x 256 tiles costs a partial extra round of the wide
// kernel: N = 8192, K = 21760: 4096 tokens 451 us, 4097 tokens 624 us -- and 464 us as 4096 + 1 tokens in two launches on the same stream; 2049 tokens
// 341 -> 240 us; N = 4096, K = 11008 at 4100 tokens 188 -> 134 us (profiles/r04_ragged_m.txt).  For the automatic dispatch, when the whole call is the
// wide kernel's: the rows that fill whole tiles (or whole rounds) go first, the remainder (at most 2048 tokens) follows as a call of its own, if the
// models price the pair at least 7 % below the single launch.  Returns the first launch's rows, 0 = no split.
// (the largest remainder that is tried; QQQ_AMD_SPLIT_CAP overrides it for measurements -- 512 against 4096 on five layer shapes, profiles/r04_ragged_m.txt:
// the larger remainders gain 5 ... 13 % wherever the models choose them and lose nowhere)
static int split_remainder_cap() {
  static const int cap = [] {
    const char* e = getenv("QQQ_AMD_SPLIT_CAP");
    if (!e) return 2048;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    return (end == e || *end != 0 || v <= 0 || v > (1 << 20)) ? 2048 : (int)v;  // not a positive number: the default, not "never split"
  }();
  return cap;
}
// process-wide switches read once from the environment: bit 4 = QQQ_AMD_NO_LOCAL_DEPOSITS=1 (as tune.fused | 16: split-K deposits are always
// written through -- the kill switch for the XCD-local hand-off, should a driver or partition mode change what HW_REG_XCC_ID / the L2 do)
static int handoff_env_flags() {
  static const int f = [] {
    const char* e = getenv("QQQ_AMD_NO_LOCAL_DEPOSITS");
    return (e && e[0] == '1') ? 4 : 0;
  }();
  return f;
}
static int choose_split(const int M, const int N, const int K, const bool grouped, const int max_par, const bool have_C, const bool have_ws,
                        const qqq_tune_t& t, const Plan& pl, const double est_whole) {
  if (t.split_m < 0 || t.kernel != 0 || t.mt != 0 || t.bm != 0 || t.ksplit > 0 || pl.kernel != 5 || est_whole <= 0.0) return 0;
  const int rows = 16 * pl.mt;
  const long long tiles_n = (N + pl.bm - 1) / pl.bm;
  const int cus = device_cus() > 0 ? device_cus() : 256;
  int cand[3] = {(M / rows) * rows, 0, 0};
  const long long tiles = (long long)((M + rows - 1) / rows) * tiles_n;
  if (tiles > cus) cand[1] = (int)((tiles / cus) * cus / tiles_n) * rows;  // the m-blocks that whole rounds cover
  // ... and the whole rounds of 256 x 256 tiles, whatever shape the whole call was planned in (round 6: with the refitted rates 5000 tokens at the BASELINE layer are
  // planned as 256 x 128 tiles -- five full rounds, 598 us by the model -- whose own candidates do not contain 4096 + 904: 545 us, measured 560 against 620)
  const long long tiles256 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  if (tiles256 > cus) cand[2] = (int)((tiles256 / cus) * cus / ((N + 255) / 256)) * 256;
  int best_m0 = 0;
  double best = 0.93 * est_whole;
  qqq_tune_t tn = t;
  tn.split_m = -1;
  for (int c = 0; c < 3; ++c) {
    const int M0 = cand[c];
    if (M0 <= 0 || M0 >= M || M - M0 > split_remainder_cap() || (c >= 1 && M0 == cand[0]) || (c == 2 && M0 == cand[1])) continue;
    double e0 = -1.0, er = -1.0;
    (void)make_plan(M0, N, K, grouped, max_par, have_C, have_ws, tn, &e0);
    (void)make_plan(M - M0, N, K, grouped, max_par, have_C, have_ws, tn, &er);
    if (e0 > 0.0 && er > 0.0 && e0 + er < best) {
      best = e0 + er;
      best_m0 = M0;
    }
  }
  return best_m0;
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
  double est = -1.0;
  const Plan pl = make_plan(prob_m, prob_n, prob_k, groupsize != -1, max_par, have_scratch != 0, have_workspace != 0, t, &est);
  memset(plan_out, 0, sizeof(*plan_out));
  plan_out->split_m = choose_split(prob_m, prob_n, prob_k, groupsize != -1, max_par, have_scratch != 0, have_workspace != 0, t, pl, est);
  plan_out->kernel = pl.kernel;
  plan_out->ksplit = pl.ksplit;
  plan_out->fused = pl.fused | (pl.kernel == 5 && pl.exch ? 64 : 0);
  plan_out->waves = pl.waves;
  plan_out->pf = pl.pf;
  plan_out->mt = pl.mt;
  plan_out->bm = pl.bm;
  plan_out->stages = pl.stages;
  plan_out->glds = pl.kernel == 2 ? (pl.stages == 0 ? 2 : 1) : pl.kernel == 5 ? (pl.chain ? 2 : 1) : 0;
  plan_out->nslots = pl.nslots;
  plan_out->pw = pl.pw;
  plan_out->skew = pl.skew;
  plan_out->w8 = pl.w8;
  return QQQ_OK;
}

// The cost models' price (us) of each family for one problem, exactly as make_plan evaluates them -- so that ONE tool (tools/cost_model_report.py) can hold
// every model against every committed measurement.  out[0] column, [1] stream, [2] panel, [3] wide; <= 0: not a candidate at this size.
extern "C" int qqq_w4a8_model_us(int prob_m, int prob_n, int prob_k, int groupsize, int max_par, double* out) {
  g_err[0] = 0;
  if (!out || prob_m <= 0 || prob_n <= 0 || prob_k <= 0 || (prob_n % 64) != 0 || (prob_k % 64) != 0) {
    snprintf(g_err, sizeof(g_err), "qqq_w4a8_model_us: bad argument");
    return QQQ_ERR_ARG;
  }
  const int M = prob_m, N = prob_n, K = prob_k;
  const bool grouped = groupsize != -1;
  const long long cap_rows = (long long)(max_par > 0 ? max_par : 0) * 64, cap_tk = (long long)(N / 128) * (max_par > 0 ? max_par : 0);
  for (int i = 0; i < 4; ++i) out[i] = -1.0;
  if (M <= 32) {
    out[0] = column_small_estimate(M, N, K, grouped);
    out[1] = stream_small_estimate(M, N, K, grouped);
    if (M > 8) {
      int a = 0, b = 0, c = 0;
      out[2] = panel_estimate(M, N, K, grouped, cap_rows > 0, cap_rows, cap_tk, &a, &b, &c);
    }
    return QQQ_OK;
  }
  int a = 0, b = 0, c = 0;
  if (M <= 256) out[1] = stream_estimate(M, N, K, grouped, cap_rows > 0, cap_rows);
  int pm = 0;
  out[2] = panel_estimate(M, N, K, grouped, cap_rows > 0, cap_rows, cap_tk, &a, &b, &c, &pm);
  if (M > 256) {
    const double w = wide_estimate(M, N, K, grouped, cap_rows > 0, cap_rows, cap_tk, &a, &b, &c);
    out[3] = w < 1e29 ? w : -1.0;
  }
  return QQQ_OK;
}

extern "C" int qqq_w4a8_gemm_ex(const void* A, const void* B, void* C, void* D, const void* s1,
                                const void* s2, const void* s3, int prob_m, int prob_n, int prob_k,
                                void* workspace, int groupsize, int dev, void* stream, int thread_k,
                                int thread_n, int sms, int max_par, const qqq_tune_t* tune,
                                int32_t* acc_out, const void* bias) {
  return qqq_w4a8_gemm_ex2(A, B, C, D, s1, s2, s3, prob_m, prob_n, prob_k, workspace, groupsize, dev, stream, thread_k, thread_n, sms, max_par,
                           tune, acc_out, bias, nullptr);
}

extern "C" int qqq_w4a8_gemm_ex2(const void* A, const void* B, void* C, void* D, const void* s1,
                                 const void* s2, const void* s3, int prob_m, int prob_n, int prob_k,
                                 void* workspace, int groupsize, int dev, void* stream, int thread_k,
                                 int thread_n, int sms, int max_par, const qqq_tune_t* tune,
                                 int32_t* acc_out, const void* bias, const void* W8) {
  g_err[0] = 0;
  const int rc = ref_shape_check(prob_m, prob_n, prob_k, groupsize, thread_k, thread_n);
  if (rc != QQQ_OK) return rc;
  if (prob_m == 0 || prob_n == 0 || prob_k == 0) return QQQ_OK;  // reference :1002-1003
  if (!A || !B || !D || !s1 || !s2 || (groupsize != -1 && !s3)) {
    snprintf(g_err, sizeof(g_err), "null pointer argument");
    return QQQ_ERR_ARG;
  }
  // the kernels use 16-byte vector / LDS-DMA accesses on A, B, C, D, bias, acc_out, s3 and 8-byte loads on s2
  if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)D | (uintptr_t)bias | (uintptr_t)acc_out |
        (groupsize != -1 ? (uintptr_t)s3 : 0)) & 15) != 0 ||
      (((uintptr_t)s2) & 7) != 0 || (((uintptr_t)s1 | (uintptr_t)workspace) & 3) != 0 || (((uintptr_t)W8) & 15) != 0) {
    snprintf(g_err, sizeof(g_err), "misaligned pointer argument (A/B/C/D/s3/bias/acc_out/W8: 16 bytes, s2: 8, s1/workspace: 4)");
    return QQQ_ERR_ARG;
  }
  const bool grouped = groupsize != -1;
  qqq_tune_t t;
  memset(&t, 0, sizeof(t));
  if (tune) t = *tune;
  // the layer's expanded int8 weights (qqq_expand_int8): used where the plan is the wide kernel's; tune.w8 = -1 ignores them
  t.w8 = (W8 != nullptr && t.w8 >= 0) ? 1 : -1;
  const int M = prob_m, N = prob_n, K = prob_k;
  // the device this call runs on decides the CU count the plan and the launch work with (not "the current device"); `sms`
  // (reference: the number of persistent threadblocks, -1 = every SM) caps it
  int cur_dev = dev;
  if (dev < 0) (void)hipGetDevice(&cur_dev);
  const int cus_dev = device_cus_of(cur_dev);
  const bool capped = sms > 0 && sms < cus_dev;
  CallCus call_cus(capped ? masked_cus(sms) : cus_dev);
  if (handoff_env_flags() & 4) t.fused |= 16;  // QQQ_AMD_NO_LOCAL_DEPOSITS=1: every split-K deposit is written through
  double est = -1.0;
  const Plan pl = make_plan(M, N, K, grouped, max_par, C != nullptr, workspace != nullptr, t, &est);
  if (const int M0 = choose_split(M, N, K, grouped, max_par, C != nullptr, workspace != nullptr, t, pl, est)) {
    // two launches on the same stream, C / workspace shared (each launch leaves the workspace zero); K % 64 == 0 and N % 64 == 0 keep every
    // offset pointer 16-byte aligned
    qqq_tune_t tn = t;
    tn.split_m = -1;
    const int rc0 = qqq_w4a8_gemm_ex2(A, B, C, D, s1, s2, s3, M0, N, K, workspace, groupsize, dev, stream, thread_k, thread_n, sms, max_par, &tn,
                                      acc_out, bias, W8);
    if (rc0 != QQQ_OK) return rc0;
    return qqq_w4a8_gemm_ex2(static_cast<const int8_t*>(A) + (size_t)M0 * K, B, C, static_cast<_Float16*>(D) + (size_t)M0 * N,
                             static_cast<const float*>(s1) + M0, s2, s3, M - M0, N, K, workspace, groupsize, dev, stream, thread_k, thread_n, sms,
                             max_par, &tn, acc_out ? acc_out + (size_t)M0 * N : nullptr, bias, W8);
  }

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
  a.skew = pl.skew;
  a.hflags = (pl.kernel == 2 || pl.kernel == 4 || pl.kernel == 5) ? ((t.fused >> 2) & (pl.kernel == 4 ? 15 : 7)) : 0;  // (panel: bit 3 = plain grid order, the slices of a tile NOT gathered on one XCD)
  if (pl.kernel == 5 && pl.exch) a.hflags |= 8;  // (wide: bit 3 = exchange hand-off of a two-slice split)
  if (pl.kernel == 5 && (t.fused & 128)) a.hflags |= 16;  // (wide, measurement: the two slices of a tile on neighbouring XCDs)

  DeviceGuard guard(dev);
  hipError_t e = hipSuccess;
  bool reduce_launch;
  if ((pl.kernel == 1 || pl.kernel == 3 || pl.kernel == 4) && (M + 16 * pl.mt - 1) / (16 * pl.mt) > 65535) {
    snprintf(g_err, sizeof(g_err), "m=%d exceeds the grid of the small-m kernels (forced by tune)", M);
    return QQQ_ERR_ARG;
  }
  // sms < CUs: the kernels of this call go to the CU-masked stream of (device, sms, caller stream), between a fork from and a join into `stream`
  MaskedFork mf;
  if (capped) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(a.stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    if (cap == hipStreamCaptureStatusNone) {
      MaskedStream* ms = masked_stream(cur_dev, sms, cus_dev, a.stream);
      if (!ms) {
        snprintf(g_err, sizeof(g_err), "sms=%d: could not create a CU-masked stream", sms);
        return QQQ_ERR_HIP;
      }
      mf.lock = std::unique_lock<std::mutex>(ms->mu);
      if ((e = hipEventRecord(ms->fork, a.stream)) != hipSuccess || (e = hipStreamWaitEvent(ms->s, ms->fork, 0)) != hipSuccess)
        return fail_hip(e, "fork into the CU-masked stream");
      mf.ms = ms;
      mf.caller = a.stream;
      a.stream = ms->s;
    }
  }
  if (pl.kernel == 5) {
    if (pl.w8) a.B = static_cast<const unsigned char*>(W8);  // the loop reads the expanded weights; s3 is not touched
    e = launch_wide(a, pl.w8 ? 2 : (grouped ? 1 : 0), pl.mt, pl.bm, pl.pf, pl.pw, pl.ksplit, pl.chain != 0);
    if (e != hipSuccess) return fail_hip(e, "qqq_wide_kernel launch");
    reduce_launch = false;
  } else if (pl.kernel == 4) {
    e = launch_panel(a, grouped, pl.mt, pl.bm, pl.waves, pl.pw, pl.pf, pl.stages, pl.ksplit);
    if (e != hipSuccess) return fail_hip(e, "qqq_panel_kernel launch");
    reduce_launch = false;
  } else if (pl.kernel == 3) {
    e = launch_column(a, grouped, pl.mt, pl.pf, pl.ksplit, pl.waves);
    if (e != hipSuccess) return fail_hip(e, "qqq_column_kernel launch");
    reduce_launch = pl.ksplit > 1;
  } else if (pl.kernel == 1) {
    // kernel arg: 0 = slabs only (separate reduce launch), 1 = in-launch over the slabs with release / acquire fences, 3 = in-launch through
    // arrival-order slots (the last arrival keeps its tile in LDS; uneven slices: skew in 64-k steps rides in bits 16.. of the K split)
    e = launch_stream(a, grouped, pl.mt, pl.waves, pl.pf, pl.ksplit | ((pl.fused == 3 && pl.ksplit > 1 ? pl.skew & 0xff : 0) << 16),
                      pl.ksplit > 1 ? (pl.fused == 1 ? 1 : (pl.fused == 3 ? 3 : 0)) : 0);
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
  if ((e = mf.join()) != hipSuccess) return fail_hip(e, "join from the CU-masked stream");
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
  if (!x || !xq || !s1 || (k % 8) != 0 || ((uintptr_t)x & 15) != 0 || ((uintptr_t)xq & 7) != 0 || ((uintptr_t)s1 & 3) != 0) {
    snprintf(g_err, sizeof(g_err), "qqq_dynamic_quant: bad argument (k must be a multiple of 8, x 16-byte / xq 8-byte aligned)");
    return QQQ_ERR_ARG;
  }
  DeviceGuard guard(dev);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const _Float16* xp = static_cast<const _Float16*>(x);
  int8_t* qp = static_cast<int8_t*>(xq);
  float* sp = static_cast<float*>(s1);
  const int nvec = k / 8;
  const int vpt = (nvec + 255) / 256;
  if ((m <= 512 || nvec > 2048) && vpt > 2 && vpt <= 32) {  // few rows (<= 2 per CU) or very long rows: 16 waves per row
    const int v4 = (nvec + 1023) / 1024;
    if (v4 <= 2)
      hipLaunchKernelGGL((qqq_dynamic_quant_kernel<2, 1024>), dim3(m), dim3(1024), 0, st, xp, qp, sp, k);
    else if (v4 <= 4)
      hipLaunchKernelGGL((qqq_dynamic_quant_kernel<4, 1024>), dim3(m), dim3(1024), 0, st, xp, qp, sp, k);
    else
      hipLaunchKernelGGL((qqq_dynamic_quant_kernel<8, 1024>), dim3(m), dim3(1024), 0, st, xp, qp, sp, k);
  } else if (vpt <= 2)
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

extern "C" int qqq_quantlinear_forward(const void* x, void* xq, void* s1, const void* B, void* C, void* D,
                                       const void* s2, const void* s3, int m, int n, int k, void* workspace,
                                       int groupsize, int dev, void* stream, int max_par, const void* bias) {
  const int rc = qqq_dynamic_quant(x, xq, s1, m, k, dev, stream);
  if (rc != QQQ_OK) return rc;
  return qqq_w4a8_gemm_ex(xq, B, C, D, s1, s2, s3, m, n, k, workspace, groupsize, dev, stream, -1, -1, -1, max_par,
                          nullptr, nullptr, bias);
}

extern "C" int qqq_quantlinear_forward2(const void* x, void* xq, void* s1, const void* B, void* C, void* D,
                                        const void* s2, const void* s3, int m, int n, int k, void* workspace,
                                        int groupsize, int dev, void* stream, int max_par, const void* bias, const void* W8) {
  const int rc = qqq_dynamic_quant(x, xq, s1, m, k, dev, stream);
  if (rc != QQQ_OK) return rc;
  return qqq_w4a8_gemm_ex2(xq, B, C, D, s1, s2, s3, m, n, k, workspace, groupsize, dev, stream, -1, -1, -1, max_par,
                           nullptr, nullptr, bias, W8);
}

// ---- load-time expansion of per-group weights (opt-in): B + s3 -> W8, once per layer; see qqq_small.hip.h for the layout
extern "C" int qqq_expand_int8(const void* B, const void* s3, void* W8, int k, int n, int groupsize, int dev, void* stream) {
  g_err[0] = 0;
  if (k == 0 || n == 0) return QQQ_OK;
  const bool grouped = groupsize != -1;
  if (!B || (grouped && !s3) || !W8 || (grouped && groupsize != 128) || k < 0 || n < 0 || (k % 128) != 0 || (n % 64) != 0 || (k / 64) > 65535 ||
      (long long)n * k >= (1ll << 32) || (((uintptr_t)B | (uintptr_t)W8) & 15) != 0 || (grouped && ((uintptr_t)s3 & 3) != 0)) {
    snprintf(g_err, sizeof(g_err), "qqq_expand_int8: need groupsize -1 or 128, k %% 128 == 0, n %% 64 == 0, n * k < 4 GiB, 16-byte aligned B / W8");
    return QQQ_ERR_ARG;
  }
  DeviceGuard guard(dev);
  if (grouped)
    hipLaunchKernelGGL(qqq_expand_int8_kernel<true>, dim3(n / 64, k / 64), dim3(128), 0, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned*>(B), static_cast<const _Float16*>(s3), static_cast<v4u*>(W8), n);
  else
    hipLaunchKernelGGL(qqq_expand_int8_kernel<false>, dim3(n / 64, k / 64), dim3(128), 0, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned*>(B), static_cast<const _Float16*>(nullptr), static_cast<v4u*>(W8), n);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "qqq_expand_int8_kernel launch");
  return QQQ_OK;
}

// ---- int4 packer / unpacker: device buffers go through the HIP kernels, host buffers through the
// same closed form on the CPU (a thread per slice of k-tiles).  Offline format conversion, not part of the GEMM path.
template <bool PACK>
static void pack_host_range(const int8_t* codes_in, int8_t* codes_out, const unsigned* B_in, unsigned* B_out, int n,
                            int grouped, int kt0, int kt1) {
  int pb[8], pr[8];
  for (int p = 0; p < 8; ++p) qqq_nibble_coords(p, grouped != 0, pb[p], pr[p]);
  const size_t row_words = 2 * (size_t)n;
  for (int kt = kt0; kt < kt1; ++kt)
    for (int ng = 0; ng < n / 64; ++ng)
      for (int wi = 0; wi < 128; ++wi) {
        const int c = wi >> 4, kq = (wi >> 2) & 3, jt = wi & 3;
        const size_t widx = (size_t)kt * row_words + 128 * (size_t)ng + wi;
        if constexpr (PACK) {
          unsigned w = 0;
          for (int p = 0; p < 8; ++p)
            w |= ((unsigned)codes_in[(size_t)(16 * kt + 4 * kq + pr[p]) * n + 64 * ng + 16 * jt + 8 * pb[p] + c] & 0xFu)
                 << (4 * p);
          B_out[widx] = w;
        } else {
          const unsigned w = B_in[widx];
          for (int p = 0; p < 8; ++p) {
            const int u = (int)((w >> (4 * p)) & 0xFu);
            codes_out[(size_t)(16 * kt + 4 * kq + pr[p]) * n + 64 * ng + 16 * jt + 8 * pb[p] + c] =
                (int8_t)((grouped || u < 8) ? u : u - 16);
          }
        }
      }
}

template <bool PACK>
static int pack_entry(const void* src, void* dst, int k, int n, int grouped, int on_device, int dev, void* stream) {
  g_err[0] = 0;
  if (k == 0 || n == 0) return QQQ_OK;
  if (!src || !dst || k < 0 || n < 0 || (k % 16) != 0 || (n % 64) != 0 || (k / 16) > 65535 ||
      ((uintptr_t)src & 7) != 0 || ((uintptr_t)dst & 7) != 0) {
    snprintf(g_err, sizeof(g_err), "qqq_%spack_int4: need k %% 16 == 0 (k <= 1048560), n %% 64 == 0, 8-byte aligned buffers",
             PACK ? "" : "un");
    return QQQ_ERR_ARG;
  }
  if (on_device) {
    DeviceGuard guard(dev);
    const dim3 grid(n / 64, k / 16);
    if constexpr (PACK)
      hipLaunchKernelGGL(qqq_pack_int4_kernel, grid, dim3(128), 0, static_cast<hipStream_t>(stream),
                         static_cast<const int8_t*>(src), static_cast<unsigned*>(dst), n, grouped);
    else
      hipLaunchKernelGGL(qqq_unpack_int4_kernel, grid, dim3(128), 0, static_cast<hipStream_t>(stream),
                         static_cast<const unsigned*>(src), static_cast<int8_t*>(dst), n, grouped);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, PACK ? "qqq_pack_int4_kernel launch" : "qqq_unpack_int4_kernel launch");
    return QQQ_OK;
  }
  const int kts = k / 16;
  unsigned hw = std::thread::hardware_concurrency();
  int nthr = (int)(hw ? hw : 1);
  if (nthr > 32) nthr = 32;
  if ((long long)k * n < (1 << 20)) nthr = 1;
  if (nthr > kts) nthr = kts;
  auto work = [&](int t) {
    const int kt0 = (int)((long long)kts * t / nthr), kt1 = (int)((long long)kts * (t + 1) / nthr);
    pack_host_range<PACK>(static_cast<const int8_t*>(src), static_cast<int8_t*>(dst), static_cast<const unsigned*>(src),
                          static_cast<unsigned*>(dst), n, grouped, kt0, kt1);
  };
  if (nthr == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nthr; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
  }
  return QQQ_OK;
}

extern "C" int qqq_pack_int4(const void* codes, void* B, int k, int n, int grouped, int on_device, int dev, void* stream) {
  return pack_entry<true>(codes, B, k, n, grouped, on_device, dev, stream);
}

extern "C" int qqq_unpack_int4(const void* B, void* codes, int k, int n, int grouped, int on_device, int dev, void* stream) {
  return pack_entry<false>(B, codes, k, n, grouped, on_device, dev, stream);
}

extern "C" int qqq_amd_abi_version(void) { return QQQ_AMD_ABI_VERSION; }
extern "C" const char* qqq_amd_last_error(void) { return g_err; }

#ifdef QQQ_PANEL_TRACE
// measurement builds only (see qqq_panel.hip.h): where the panel kernel's phase clocks go
extern "C" int qqq_trace_set(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(qqq_trace_buf), &p, sizeof p); }
#endif
