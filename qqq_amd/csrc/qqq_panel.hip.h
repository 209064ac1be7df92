// qqq_panel.hip.h -- "panel" kernel (from about 64 tokens up, in m-blocks of up to 128 tokens: weights HBM -> VGPR per wave,
// activations shared through LDS, in-launch split-K; with 64 columns per wave also the large-m kernel of the per-channel mode).  Part of the single translation unit qqq_w4a8.hip (see its header comment for the design).
#ifndef QQQ_AMD_QQQ_PANEL_HIP_H_
#define QQQ_AMD_QQQ_PANEL_HIP_H_

// ------------------------------------------------------------------------------------------
// "panel" kernel: one workgroup = ALL 16*MT tokens of an m-block x BN = 32*WN weight columns x one K slice.
//
//  * The waves of a workgroup split the COLUMNS (wave wn owns the 32 columns of half-group (ng, half), exactly the
//    column kernel's set: chunks c = 4*half + c' of a 64-column group = 256 contiguous bytes of every 16-k row of
//    B), so every weight byte is loaded by exactly one wave, straight from HBM into VGPRs (used once: no LDS,
//    no barrier on the weight path), PFS stages ahead.  The packed words are re-distributed between the lanes of a
//    quad with the 8-VALU 4x4 transpose of qqq_common.hip.h (quad_transpose4); v_mfma_i32_16x16x64_i8, MFMA row
//    i = 4*c' + jt.
//  * The activations (16*MT tokens x 128 k per stage) are the operand every wave needs: they are staged ONCE per
//    workgroup in a double- / triple-buffered LDS image (16-byte chunks XOR-swizzled by the row: every ds_read_b128
//    is conflict free), so the L1 traffic of a workgroup is weights + activations once -- not activations once per
//    wave as in the stream / column kernels, which is what bounds those above m = 16.  They travel global -> VGPR
//    -> ds_write (XL stages ahead), NOT by LDS-DMA: measured on this part, global_load_lds moves ~34 B/clk/CU while
//    plain 16-byte loads reach ~64 B/clk/CU (profiles/r01_probe_fill.txt), and at m = 128 a stage needs 24 KB of
//    operands per 576 matrix-pipe cycles -- the LDS-DMA variant of this kernel was bound by exactly that
//    (profiles/r02_panel_dma_variant.txt).  A welcome side effect: every load is visible to hipcc, which therefore
//    places exact counted s_waitcnt vmcnt(N) itself (loads return in order; the steady-state loop is branch-free).
//  * KG = 2: two k-groups of WN waves; group kg takes the 64-k half kg of every 128-k stage (so one LDS stage
//    feeds both) and the two partial tiles meet in LDS at the end.  BN = 128 with 8 waves: 4 K slices fill 256 CUs
//    at N = 8192, which keeps the split-K partial-sum traffic at 3 x M x N x 4 B.
//  * split-K in-launch: arrival-order tickets as in the tiled kernel, but every depositor owns a slot (slot index =
//    arrival index), so nobody waits except the last arrival, which has by construction only already-arrived
//    workgroups to wait for.  Deposits are lane-linear full-line write-through stores.  With two k-groups and at
//    least two m-tiles BOTH groups finish the tile (each keeps half of the m-tiles after the meeting in LDS).
//  * HW = 2: a wave owns BOTH halves of its 64-column group (two weight loads, 4*MT MFMAs per 64-k step): every
//    activation fragment read from LDS feeds 4 MFMAs instead of 2 and the unpack work per MFMA halves -- the loop
//    of this kernel is bound by VALU ISSUE (VALU and MFMA share the SIMD's issue port), not by the LDS or the
//    matrix pipe: 3.1 VALU per 16-cycle MFMA kept that pipe at 50 %, HW = 2 with the cheap transpose runs 1.45.
// grid = (ceil(N / BN), ksplit, ceil(M / (16*MT)));  block = 64 * WN * KG;  BN = 32 * WN * HW.
// ------------------------------------------------------------------------------------------

// Which shapes run the four-buffer activation ring with a barrier after every OTHER stage (see `stage`): the 64-column shape
// (two k-groups: every buffer is read in one stage only; ring period 4: the stage parity is a compile-time property of the
// unrolled loop) with the 2-stage activation register ring -- with 4 stages it sits at 256 VGPRs and would spill.  The
// 32-column shapes gain nothing from it (measured twice, the second time with the cheaper hand-off and the 2-deep activation
// ring: M = 96-128 1-3 % SLOWER, the extra prologue stage outweighs the barriers saved) and keep the plain two-buffer ring.
__host__ __device__ constexpr bool qqq_panel_relaxed(int KG, int PFS, int XL, int HW) {
  return KG == 2 && HW == 2 && PFS == 4 && XL == 2;
}

// Measurement only (tools/ablate_panel.sh, profiles/r02_panel_cw2_ablation.txt): -DQQQ_PANEL_ABLATE=<bits> removes parts of the
// steady-state loop (either shape; profiles/r02_panel_prefetch_depth.txt has the 32-column one) -- 1 stage-end barrier, 2 activation staging, 4 transpose + shift/mask, 8 weight-ring
// refill, 16 LDS fragment reads.  Results are wrong by construction; never defined in a shipped build.
#ifndef QQQ_PANEL_ABLATE
#define QQQ_PANEL_ABLATE 0
#endif
// Measurement only (tools/trace_panel.py, profiles/r02_panel_timeline.txt): -DQQQ_PANEL_TRACE makes thread 0 of every
// workgroup write the 100 MHz wall clock at the phase boundaries below into a buffer set through `qqq_trace_set`.
#ifdef QQQ_PANEL_TRACE
__device__ unsigned long long* qqq_trace_buf;
#define QQQ_TR(i)                                                                                                       \
  do {                                                                                                                  \
    if (threadIdx.x == 0 && qqq_trace_buf)                                                                              \
      qqq_trace_buf[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64(); \
  } while (0)
#define QQQ_TRV(i, v)                                                                                                   \
  do {                                                                                                                  \
    if (threadIdx.x == 0 && qqq_trace_buf)                                                                              \
      qqq_trace_buf[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (i)] = (unsigned long long)(v); \
  } while (0)
#else
#define QQQ_TR(i) do {} while (0)
#define QQQ_TRV(i, v) do {} while (0)
#endif

template <int MT, bool GROUPED, int WN, int KG, int PFS, int XL, int HW>
__global__ __launch_bounds__(64 * WN * KG) void qqq_panel_kernel(
    const int8_t* __restrict__ A, const unsigned char* __restrict__ B, int32_t* __restrict__ C,
    _Float16* __restrict__ D, const float* __restrict__ s1, const float* __restrict__ s2,
    const _Float16* __restrict__ s3, int32_t* __restrict__ acc_out, int* __restrict__ tickets,
    const _Float16* __restrict__ bias, const int M, const int N, const int K, const int ksplit_hf) {
  // (hand-off switches ride in the upper half of the K-split argument -- tune.fused bits 2 / 3: 1 = the formal agent-scope ACQUIRE
  // fence in front of the fold, 2 = agent-scope RELEASE on the depositor's completion count; see qqq_common.hip.h)
  // (bits 24..29: `skew`, the 128-k stages the LAST K slice gets on top of an even share -- see st_begin below)
  const int ksplit = ksplit_hf & 0xffff, hflags = (ksplit_hf >> 16) & 0xff, skew = (ksplit_hf >> 24) & 0x3f;
  constexpr int NW = WN * KG;            // waves
  constexpr int NT = NW * 64;            // threads
  constexpr int BN = 32 * WN * HW;       // columns per workgroup
  static_assert(HW == 1 || HW == 2, "32-column sets per wave");
  constexpr int ROWS = 16 * MT;          // tokens per workgroup
  constexpr int XB = ROWS * 128;         // bytes of one activation stage
  constexpr int XCH = XB / 16;           // ... in 16-byte chunks
  constexpr int XPT = (XCH + NT - 1) / NT;  // chunks per thread
  constexpr int SPW = 2 / KG;            // 64-k steps (= weight loads) per wave and stage
  static_assert(PFS % XL == 0 && XL >= 2, "ring periods");
  // LDS stage buffers (see `stage`).  RELAX (qqq_panel_relaxed): the image of a stage is written THREE stages
  // ahead into one of four buffers, which leaves room for a barrier at the end of every OTHER stage only.
  constexpr bool RELAX = qqq_panel_relaxed(KG, PFS, XL, HW);
  constexpr int LA = RELAX ? 3 : 2;
  constexpr int NBUF = RELAX ? 4 : (SPW == 1) ? 2 : 3;
  constexpr int EP_STRIDE = BN + 4;      // ints per row of the epilogue image (bank skew)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int xch;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  QQQ_TR(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN;  // column set
  const int kg = wave / WN;  // k-group
  // Workgroup -> (strip, K slice).  The hardware deals consecutive workgroups round-robin to the 8 XCDs; with the plain grid order (strip = x, slice = y) the slices
  // of a tile share an XCD only when the number of strips is a multiple of 8.  For a split K the grid is therefore walked 8 strips at a time through all their
  // slices, so that every tile's slices meet in ONE L2 whatever N is and the finisher finds the other slices' deposits there: Llama-2-7B gate / up (N = 11008: 86 / 43
  // strips) at 48 tokens 16.3 -> 15.5 us, 256-column strips 19.7 -> 17.5; N = 13824 18.2 -> 17.1 / 22.0 -> 18.9; nothing changes where it already was so (N = 4096,
  // 12288) -- profiles/r05_panel_slices_one_xcd.txt.  hflags & 8 (tune.fused bit 5) keeps the plain order: measurement only.
  int strip = blockIdx.x, sp = blockIdx.y;
  if ((hflags & 8) == 0 && ksplit > 1) {
    const int T = gridDim.x, L = blockIdx.x + T * blockIdx.y, full = (T >> 3) * 8 * ksplit;
    if (L < full) {
      const int r = L % (8 * ksplit);
      strip = (L / (8 * ksplit)) * 8 + (r & 7);
      sp = r >> 3;
    } else {
      const int rem = T & 7, l2 = L - full;
      strip = (T & ~7) + l2 % rem;
      sp = l2 / rem;
    }
  }
  const int mblk = blockIdx.z;
  const int mbase = mblk * ROWS;
  const int ngroups = N >> 6;
  const int gl = (wn * HW) >> 1;  // 64-column group of this wave inside the strip
  int ng = strip * (BN / 64) + gl;
  if (ng >= ngroups) ng = ngroups - 1;  // N % BN != 0: surplus waves of the last strip compute on clamped columns, store nothing
  const int half = (HW == 2) ? 0 : (wn & 1);  // first (HW == 2: both) half of the group
  const size_t rowbytes = (size_t)N * 8;

  // ---- K slice in 128-k stages (a trailing 64-k half stage when K % 128 == 64) ----
  const int KS = K >> 6;            // 64-k steps
  const int NST = (KS + 1) >> 1;    // stages
  // Uneven slices (skew > 0, host-checked to leave every slice at least a few stages): the last slice is `skew` stages longer than
  // the others, so its workgroup arrives last at the tile's ticket and finds the other deposits already written instead of waiting
  // a write-through + publish latency (~2.8 us at 128 tokens) for slices that finished together with it.  Who folds is still
  // decided by the arrival order -- a slice that is late for any other reason takes over, the result is the same.
  const int NSE = NST - skew;
  const int st_begin = (int)(((long long)NSE * sp) / ksplit);
  const int st_end = (sp == ksplit - 1) ? NST : (int)(((long long)NSE * (sp + 1)) / ksplit);
  const int nst = st_end - st_begin;
  const bool k_tail = (KS & 1) != 0;  // the last stage of the problem holds one 64-k step only

  // ---- per-lane sources ----
  const int h = lane >> 4, cq = (lane >> 2) & 3, q4 = lane & 3;  // q4: kq as a load lane, jt as an MFMA lane
  const unsigned char* wptr = B + (size_t)h * rowbytes + (size_t)ng * 512 + (4 * half + cq) * 64 + q4 * 16;
  const _Float16* sptr = GROUPED ? (s3 + (size_t)ng * 64 + (4 * half + cq) * 8 + 2 * q4) : nullptr;
  // activation staging: chunk ci = tid + q*NT of the stage image = (row ci >> 3, 16-byte piece ci & 7)
  const unsigned char* xsrc[XPT];
  unsigned xdst[XPT];
  bool xup[XPT];  // piece lies in the upper 64 k of the stage
#pragma unroll
  for (int q = 0; q < XPT; ++q) {
    const int ci = tid + q * NT;
    const int row = (ci >> 3) % ROWS, pos = ci & 7;
    int grow = mbase + row;
    if (grow >= M) grow = M - 1;
    xsrc[q] = reinterpret_cast<const unsigned char*>(A) + (size_t)grow * K + pos * 16;
    xdst[q] = (unsigned)(row * 128 + ((pos ^ ((row >> 1) & 7)) << 4));
    xup[q] = pos >= 4;
  }
  const bool xact = (XCH % NT == 0) || tid < XCH % NT;  // wave-uniform: XCH % NT is a multiple of 64 whenever it is not 0
  static_assert(XCH % 64 == 0, "activation stage");

  // Stage indices past the slice are redirected to its last stage (loaded, never used): the loop stays branch-free.
  auto load_x = [&](const int st_rel, v4u (&r)[XPT]) {
    const int st = st_begin + (st_rel < nst ? st_rel : nst - 1);
    const bool half_only = k_tail && st == NST - 1;
#pragma unroll
    for (int q = 0; q < XPT; ++q)
      if (q + 1 < XPT || xact) {
        // the upper 64 k of a trailing half stage lie outside the row: fetch the lower half again (never used)
        const unsigned char* p = xsrc[q] + (size_t)st * 128 - ((half_only && xup[q]) ? 64 : 0);
        r[q] = *reinterpret_cast<const v4u*>(p);
      }
  };
  auto store_x = [&](const int buf, const v4u (&r)[XPT]) {
#pragma unroll
    for (int q = 0; q < XPT; ++q)
      if (q + 1 < XPT || xact) *reinterpret_cast<v4u*>(smem + buf * XB + xdst[q]) = r[q];
  };
  auto load_w = [&](const int st_rel, const int t, v4u (&dst)[HW]) {
    const int st = st_begin + (st_rel < nst ? st_rel : nst - 1);
    int s = 2 * st + (KG == 2 ? kg : t);
    if (s >= KS) s = KS - 1;
#pragma unroll
    for (int hf = 0; hf < HW; ++hf) {
      if constexpr (((QQQ_W_NT & 4) != 0 && MT <= 4) || ((QQQ_W_NT & 16) != 0 && MT > 4)) dst[hf] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(wptr + (size_t)(4 * s) * rowbytes + 256 * hf));
      else dst[hf] = *reinterpret_cast<const v4u*>(wptr + (size_t)(4 * s) * rowbytes + 256 * hf);
    }
  };
  auto load_sc = [&](const int st_rel, h2 (&dst)[HW]) {
    const int st = st_begin + (st_rel < nst ? st_rel : nst - 1);
#pragma unroll
    for (int hf = 0; hf < HW; ++hf) dst[hf] = *reinterpret_cast<const h2*>(sptr + (size_t)st * N + 32 * hf);
  };

  v4i acc[MT][2 * HW];  // [mt][2*hf + b]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 2 * HW; ++q) acc[mt][q] = (v4i){0, 0, 0, 0};

  v4u wr[PFS * SPW][HW];
  v4u xr[XL][XPT];
  h2 scr[GROUPED ? PFS : 1][HW];
  const unsigned xrd = (unsigned)((lane & 15) * 128);  // + mt*2048; chunk = (4*t + h) ^ ((row >> 1) & 7), row = 16*mt + j
  const int xsw = ((lane & 15) >> 1) & 7;              // (row >> 1) & 7 for row = 16*mt + j

  // MFMA operands of one 64-k step of this wave: weights (two column halves b) and the tokens' activations
  struct Operands {
    v4i a[2 * HW];  // [2*hf + b]
  };
  v4i x[MT];  // activation fragments of the current step; x[mt] is re-read for the next step right behind its last MFMA
  auto read_x = [&](const int stage_abs_rel, const int tk, const int mt) {  // LDS buffer of a stage = stage % NBUF
    const unsigned char* st = smem + (stage_abs_rel % NBUF) * XB;
    x[mt] = *reinterpret_cast<const v4i*>(st + xrd + mt * 2048 + (((4 * tk + h) ^ xsw) << 4));
  };
  auto unpack_w = [&](const v4u& w, const h2 sc, const bool valid, const int hf, Operands& o) {
    unsigned y[4];
    if constexpr ((QQQ_PANEL_ABLATE & 4) != 0) {
      y[0] = w[0]; y[1] = w[1]; y[2] = w[2]; y[3] = w[3];
    } else {
      quad_transpose4(w, y);  // y[kq] = word kq of this lane's jt
    }
    // a step past the end of K (trailing half stage) contributes nothing: `valid` is wave-uniform, so it costs a scalar
    // select on the nibble mask (per-channel) or on the group scale (scale 0 re-quantises every nibble to 0)
    if constexpr (GROUPED) {
      const h2 zero = {(_Float16)0, (_Float16)0};
      const h2 sv = valid ? sc : zero;
      const h2 sb0 = {sv[0], sv[0]}, sb1 = {sv[1], sv[1]};
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        int w0, w1;
        unpack_pair<true>(y[kq], sb0, sb1, w0, w1);
        o.a[2 * hf][kq] = w0;
        o.a[2 * hf + 1][kq] = w1;
      }
    } else {
      const unsigned nm = valid ? QQQ_NIB_MASK : 0u;
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        if constexpr ((QQQ_PANEL_ABLATE & 4) != 0) {
          o.a[2 * hf][kq] = (int)y[kq];
          o.a[2 * hf + 1][kq] = (int)(y[kq] ^ nm);
        } else {
          o.a[2 * hf][kq] = (int)(y[kq] & nm);             // odd nibbles  -> 16*w4 of column n      (b = 0)
          o.a[2 * hf + 1][kq] = (int)((y[kq] << 4) & nm);  // even nibbles -> 16*w4 of column n + 8  (b = 1)
        }
      }
    }
  };

  // One stage of the software pipeline.  While the matrix pipe works on step (i, t) -- operands `cur`, prepared one
  // step earlier -- the wave reads the next step's activations from LDS, unpacks the next step's weights (already in
  // the register ring) and refills that ring slot from HBM; sched_group_barrier spreads that work between the MFMAs.
  // LDS buffers: stage j lives in buffer j % NBUF; stage i+2 is written at the top of stage i (its buffer was last
  // read during stage i-1, or i-2 with two buffers: KG == 2 reads every buffer in one stage only), and becomes visible
  // with the barrier that ends stage i -- one stage before the first (prefetch) read.
  // HW == 2 (64 columns per wave, KG == 2) runs the weight operands half a step ahead instead of a whole one, in place
  // (128 accumulators leave no room for two operand sets): while the MFMAs of column half 0 issue, the wave unpacks THIS
  // step's half 1; while those of half 1 issue, the NEXT step's half 0 -- and refills the ring slot just emptied.
  Operands cur, nxt;
  auto stage = [&](const int i, const int u) {  // u = i % PFS as a compile-time value at every call site
    if constexpr (!(QQQ_PANEL_ABLATE & 2)) {
      store_x((i + LA) % NBUF, xr[(u + LA) % XL]);
      load_x(i + LA + XL, xr[(u + LA) % XL]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (HW == 2) {
      static_assert(HW == 1 || SPW == 1, "64 columns per wave: two k-groups");
      const int un = (u + 1) % PFS;
      constexpr int NV = GROUPED ? 68 : 14;  // compiler-visible VALU of one 32-column unpack (the transpose is an asm block)
      constexpr int VPM = (NV + 2 * MT - 1) / (2 * MT);
      // ---- column half 0 of step i ----
      unpack_w(wr[u][1], scr[GROUPED ? u : 0][1], 2 * (st_begin + i) + kg < KS, 1, cur);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur.a[0], x[mt], acc[mt][0], 0, 0, 0);
        acc[mt][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur.a[1], x[mt], acc[mt][1], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 2 * MT; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);  // VALU
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- column half 1 of step i ----
      unpack_w(wr[un][0], scr[GROUPED ? un : 0][0], 2 * (st_begin + i + 1) + kg < KS, 0, cur);
      if constexpr (!(QQQ_PANEL_ABLATE & 8)) {
        load_w(i + PFS, 0, wr[u]);
        if constexpr (GROUPED) load_sc(i + PFS, scr[u]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur.a[2], x[mt], acc[mt][2], 0, 0, 0);
        acc[mt][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur.a[3], x[mt], acc[mt][3], 0, 0, 0);
        if constexpr (!(QQQ_PANEL_ABLATE & 16)) read_x(i + 1, kg, mt);  // the next step's fragment, in place
      }
#pragma unroll
      for (int q = 0; q < 2 * MT; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // 1 MFMA
        if (q % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 LDS read (behind its fragment's last use)
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);                   // VALU
        if (q == 2 * MT - 1) __builtin_amdgcn_sched_group_barrier(0x020, GROUPED ? 4 : 2, 0);  // VMEM reads (ring refills)
      }
      __builtin_amdgcn_sched_barrier(0);
      // RELAX: stage j's image is written during stage j-3 and read during stage j-1; its buffer held stage j-4, read
      // during stage j-5: a barrier at the end of every odd stage separates each of these pairs.  (PFS even and the loop
      // counter a multiple of PFS: the parity of i is the parity of u, a compile-time property.)
      if constexpr (!(QQQ_PANEL_ABLATE & 1))
        if (!RELAX || (u & 1)) __syncthreads();
      return;
    }
#pragma unroll
    for (int t = 0; t < SPW; ++t) {
      // the step after (i, t): (i, t+1) or (i+1, 0)
      const bool same = (t + 1 < SPW);
      const int ni = same ? i : i + 1, nu = same ? u : (u + 1) % PFS, nt = same ? t + 1 : 0;
      const int ntk = (KG == 2) ? kg : nt;
#pragma unroll
      for (int hf = 0; hf < HW; ++hf)
        unpack_w(wr[nu * SPW + nt][hf], scr[GROUPED ? nu : 0][hf], 2 * (st_begin + ni) + ntk < KS, hf, nxt);
      if constexpr (!(QQQ_PANEL_ABLATE & 8)) {
        load_w(ni + PFS, nt, wr[nu * SPW + nt]);
        if constexpr (GROUPED)
          if (nt == SPW - 1) load_sc(ni + PFS, scr[nu]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int q = 0; q < 2 * HW; ++q)
          acc[mt][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur.a[q], x[mt], acc[mt][q], 0, 0, 0);
        if constexpr (!(QQQ_PANEL_ABLATE & 16)) read_x(ni, ntk, mt);  // the next step's fragment, in place
      }
      // issue order inside this region: 1 MFMA, then its share of the VALU / LDS-read / VMEM work
      constexpr int NVALU = (GROUPED ? 68 : 14) * HW;  // compiler-visible VALU of one step's unpack (the transpose is an asm block)
      constexpr int NMF = 2 * HW * MT;
      constexpr int VPM = (NVALU + NMF - 1) / NMF;
#pragma unroll
      for (int q = 0; q < NMF; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           // 1 MFMA
        if (q % (2 * HW) == 2 * HW - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 LDS read (behind its fragment's last use)
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);                         // VALU
        if (q == NMF - 1) __builtin_amdgcn_sched_group_barrier(0x020, 2 * HW, 0);    // VMEM reads (ring refills)
      }
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
    if constexpr (!(QQQ_PANEL_ABLATE & 1))
      if (!RELAX || (u & 1)) __syncthreads();  // stage i+2 is in LDS for everybody (RELAX: see the 64-column path above)
  };

  if (nst > 0) {
    // ---- prologue: stages 0 and 1 go straight into LDS.  Everything that does not need their staging registers is
    // issued BEFORE the wait for them (loads return in order, the two are the oldest): the first weight stages'
    // HBM latency overlaps the activations' instead of following it ----
    load_x(0, xr[0]);
    load_x(1, xr[1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < PFS; ++j) {
#pragma unroll
      for (int t = 0; t < SPW; ++t) load_w(j, t, wr[j * SPW + t]);
      if constexpr (GROUPED) load_sc(j, scr[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 2; j < 2 + XL; ++j)
      if (j % XL >= 2 && (!RELAX || j == 2)) load_x(j, xr[j % XL]);  // (RELAX: only stage 2 early; the ring is filled below)
    __builtin_amdgcn_sched_barrier(0);
    store_x(0, xr[0]);
    store_x(1, xr[1]);
    if constexpr (RELAX) {  // a third stage goes straight into LDS, then the ring takes stages 3 .. 3 + XL - 1
      if (XL < 3) load_x(2, xr[2 % XL]);
      store_x(2, xr[2 % XL]);
#pragma unroll
      for (int j = LA; j < LA + XL; ++j) load_x(j, xr[j % XL]);
    } else {
#pragma unroll
      for (int j = 2; j < 2 + XL; ++j)
        if (j % XL < 2) load_x(j, xr[j % XL]);
    }
    __syncthreads();
    {  // operands of the first step; its ring slot is refilled like any other
      const int tk0 = (KG == 2) ? kg : 0;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) read_x(0, tk0, mt);
      if constexpr (HW == 2) {
        unpack_w(wr[0][0], scr[0][0], 2 * st_begin + tk0 < KS, 0, cur);  // half 1 and the refill: inside stage 0
      } else {
        unpack_w(wr[0][0], scr[0][0], 2 * st_begin + tk0 < KS, 0, cur);
        load_w(PFS, 0, wr[0]);
        if constexpr (GROUPED)
          if (SPW == 1) load_sc(PFS, scr[0]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // Two LDS buffers: stage 0 re-fills buffer 0 (with stage 2) at its very top, and the fragment reads just above are
    // the only reads of a buffer that no stage-end barrier separates from its next write -- a wave delayed behind the
    // barrier (two workgroups sharing a CU) would otherwise read stage 2 rows for its first step.
    if constexpr (NBUF == 2 || RELAX) __syncthreads();
    QQQ_TR(1);
#ifdef QQQ_PANEL_PRIO
    // measurement: static priority for the later-dispatched half of the waves (the arbitration loser of every SIMD pair)
    if (__builtin_amdgcn_readfirstlane(tid) >= NT / 2) __builtin_amdgcn_s_setprio(1);
#endif
    // ---- steady state: PFS stages per iteration (ring slots are compile-time registers), branch-free ----
    int i0 = 0;
    for (; i0 + PFS <= nst; i0 += PFS) {
#pragma unroll
      for (int u = 0; u < PFS; ++u) stage(i0 + u, u);
    }
    // ---- ragged tail (< PFS stages) ----
#pragma unroll
    for (int u = 0; u < PFS - 1; ++u)
      if (i0 + u < nst) stage(i0 + u, u);
  }
  __syncthreads();

  QQQ_TR(2);
  // split-K arrival ticket: taken here, ahead of the k-group meet, so the atomic's round trip (agent scope, ~1-2 us) runs
  // under the LDS exchange below; it is consumed where the slot is chosen
  const int tile = mblk * gridDim.x + strip;
  int arrival = 0;
  // (not for the 64-column shapes: with the ticket's register live across the meet hipcc schedules their main loop ~2 % slower
  //  at M=4096 -- measured, four builds side by side -- and they run with 1-2 slices, where the ticket is a small part)
  constexpr bool HOIST = (HW == 1);
  if (HOIST && ksplit > 1 && tid == 0) arrival = __hip_atomic_fetch_add(tickets + 2 * (size_t)tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- k-groups meet in LDS (KG == 2).  With MT >= 2 the tile is then FINISHED BY BOTH groups: group kg keeps the
  // m-tiles [kg*MT/2, +MT/2) -- it deposits the other half in LDS, adds the partner's deposit to its own half -- so
  // the split-K hand-off and the epilogue below are spread over all waves (and 64, not 128, live accumulators per
  // wave with HW == 2).  MT == 1: group 1 deposits, group 0 finishes alone.
  constexpr bool SPLIT = (KG == 2) && (MT >= 2);
  constexpr int MTO = SPLIT ? MT / 2 : MT;  // m-tiles this wave finishes
  constexpr int NQ = 2 * HW;
  v4i fin[MTO][NQ];
  const int mb = SPLIT ? kg * MTO : 0;      // first of them
  const bool finisher = SPLIT || kg == 0;
  if constexpr (SPLIT) {
    v4i* red = reinterpret_cast<v4i*>(smem);  // [kg][wn][MTO][NQ][64 lanes]
    v4i* mine = red + (size_t)((kg * WN + wn) * MTO) * NQ * 64;
    const v4i* theirs = red + (size_t)(((kg ^ 1) * WN + wn) * MTO) * NQ * 64;
    if (kg == 0) {
#pragma unroll
      for (int j = 0; j < MTO; ++j)
#pragma unroll
        for (int q = 0; q < NQ; ++q) mine[(j * NQ + q) * 64 + lane] = acc[MTO + j][q];
    } else {
#pragma unroll
      for (int j = 0; j < MTO; ++j)
#pragma unroll
        for (int q = 0; q < NQ; ++q) mine[(j * NQ + q) * 64 + lane] = acc[j][q];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int j = 0; j < MTO; ++j)
#pragma unroll
        for (int q = 0; q < NQ; ++q) fin[j][q] = acc[j][q] + theirs[(j * NQ + q) * 64 + lane];
    } else {
#pragma unroll
      for (int j = 0; j < MTO; ++j)
#pragma unroll
        for (int q = 0; q < NQ; ++q) fin[j][q] = acc[MTO + j][q] + theirs[(j * NQ + q) * 64 + lane];
    }
    __syncthreads();
  } else {
    if constexpr (KG == 2) {
      v4i* red = reinterpret_cast<v4i*>(smem);
      if (kg == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < NQ; ++q) red[((wn * MT + mt) * NQ + q) * 64 + lane] = acc[mt][q];
      }
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < NQ; ++q) fin[mt][q] = acc[mt][q] + ((kg == 0) ? red[((wn * MT + mt) * NQ + q) * 64 + lane] : (v4i){0, 0, 0, 0});
      __syncthreads();
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < NQ; ++q) fin[mt][q] = acc[mt][q];
    }
  }

  QQQ_TR(3);
  // ---- epilogue scales, fetched ahead of the split-K hand-off (32-column shapes: registers to spare): one round trip
  // less behind the fold.  NT is a multiple of BN / 8, so a thread's 8 columns are the same in every pass.
  constexpr int EP_ITEMS = ROWS * (BN / 8), EP_PASSES = (EP_ITEMS + NT - 1) / NT;
  constexpr bool EP_PRE = (HW == 1) && (NT % (BN / 8) == 0);
  float a_pre[EP_PRE ? EP_PASSES : 1];
  float2 s2_pre[4];
  if constexpr (EP_PRE) {
    const int n = strip * BN + (tid % (BN / 8)) * 8;
    const int nc = (n < N) ? n : 0;
    const int i0 = s2_stored_index(nc), i1 = s2_stored_index(nc + 4);
    s2_pre[0] = *reinterpret_cast<const float2*>(s2 + i0);
    s2_pre[1] = *reinterpret_cast<const float2*>(s2 + i0 + 8);
    s2_pre[2] = *reinterpret_cast<const float2*>(s2 + i1);
    s2_pre[3] = *reinterpret_cast<const float2*>(s2 + i1 + 8);
#pragma unroll
    for (int ps = 0; ps < EP_PASSES; ++ps) {
      const int m = mbase + (tid + ps * NT) / (BN / 8);
      a_pre[ps] = s1[m < M ? m : M - 1];
    }
  }

  // ---- in-launch split-K: slot = arrival index; the last arrival folds every slot and runs the epilogue ----
  // slot image: wave wn's m-tile mt, operand q at ((wn*MT + mt)*NQ + q) KiB, lane-linear inside
  if (ksplit > 1) {
    int* tk = tickets + 2 * (size_t)tile;  // [0] arrivals, [1] completed deposits; both zero again on exit
    if (tid == 0) xch = HOIST ? arrival : __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane(xch);
    QQQ_TR(4);
    QQQ_TRV(8, t);
    const size_t slot_ints = (size_t)ROWS * BN;
    const size_t tile_ints = slot_ints * (size_t)(ksplit - 1);
    const size_t wave_ints = ((size_t)wn * MT + mb) * NQ * 256;
    if (t < ksplit - 1) {
      if (finisher) {
        const unsigned long long sbv = reinterpret_cast<unsigned long long>(C + (size_t)tile * tile_ints + (size_t)t * slot_ints + wave_ints);
        // wave-uniform by construction; pinned to SGPRs for the "s" operand of the stores below
        const unsigned char* sb = reinterpret_cast<const unsigned char*>(
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sbv >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)sbv));
        const unsigned voff = lane * 16;
#pragma unroll
        for (int j = 0; j < MTO; ++j)
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const unsigned char* p = sb + (j * NQ + q) * 1024;
            // (s_nop 1 inside the string: hipcc does not know this is a 16-byte store whose data registers are read late, and
            //  would let its next VALU instruction overwrite them -- garbage in the slot under load; s_nop 4 in front: the
            //  scalar base may come straight from a v_readfirstlane, 5 wait states ahead of a VMEM read)
            asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(fin[j][q]), "s"(p) : "memory");
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();  // every wave's part of the deposit has reached memory
      QQQ_TR(5);
      if (tid == 0) qqq_publish_add(tk + 1, hflags);
      return;
    }
    if (tid == 0) {  // everybody waited for has arrived already (is depositing): short, and bounded as a matter of principle
      int spin = 0;
      while (__hip_atomic_load(tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ksplit - 1) {
        // a depositor that never completes (pre-empted for seconds, a debugger) must not end in a silently wrong D:
        // the launch is aborted and the host sees the error at its next synchronisation
        if (++spin > QQQ_SPIN_LIMIT) __builtin_trap();
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    if (qqq_formal_acquire(hflags)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (off by default, ~3 us per finisher: qqq_common.hip.h)
    QQQ_TR(5);
    // No acquire fence here: at agent scope it is a `buffer_inv sc1` over the whole L2, measured at ~3 us of the finisher's
    // critical path (tools/trace_panel.py).  The deposits are read with agent-scope loads instead (`load16_agent`, sc1),
    // issued behind the barrier above.
    if (finisher) {
      // FB slots in flight at a time: the deposits come back from the fabric (they were written through from other
      // XCDs), one round trip each if folded slot by slot -- 3 x ~2 us at 4 slices.  The slot index of a batch's
      // surplus loads is clamped (a valid slot, read twice), only the add is skipped.
      constexpr int FB = (MTO * NQ <= 8) ? 3 : 1;  // 96 registers of deposits at most (the 64-column shapes sit at the register limit)
      for (int s0 = 0; s0 < ksplit - 1; s0 += FB) {
        v4i dep[FB][MTO][NQ];
#pragma unroll
        for (int b = 0; b < FB; ++b) {
          const int sl = min(s0 + b, ksplit - 2);
          const __amdgpu_buffer_rsrc_t view = agent_view(C + (size_t)tile * tile_ints + (size_t)sl * slot_ints + wave_ints);
#pragma unroll
          for (int j = 0; j < MTO; ++j)
#pragma unroll
            for (int q = 0; q < NQ; ++q) dep[b][j][q] = load16_agent(view, (unsigned)((j * NQ + q) * 1024 + lane * 16));
        }
#pragma unroll
        for (int b = 0; b < FB; ++b)
          if (s0 + b < ksplit - 1) {
#pragma unroll
            for (int j = 0; j < MTO; ++j)
#pragma unroll
              for (int q = 0; q < NQ; ++q) fin[j][q] += dep[b][j][q];
          }
      }
    }
    QQQ_TR(6);
    if (tid < 2) __hip_atomic_store(tk + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // workspace zero on return
  }

  // ---- epilogue: int32 tile -> LDS (row-major, skewed rows) -> 8 consecutive n per thread -> 16-byte stores ----
  // D lane ln of the MFMA holds token j = ln & 15, rows 4*(ln >> 4) + r -> c' = ln >> 4, jt = r:
  //   column inside the strip  nl = 64*gl + 16*jt + 8*b + 4*(half + hf) + c'
  int* ep = reinterpret_cast<int*>(smem);
  if (finisher) {
    const int j = lane & 15, cp = lane >> 4;
#pragma unroll
    for (int jm = 0; jm < MTO; ++jm)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ep[(16 * (mb + jm) + j) * EP_STRIDE + 64 * gl + 16 * r + 8 * (q & 1) + 4 * (half + (q >> 1)) + cp] = fin[jm][q][r];
  }
  __syncthreads();
  // (unrolled: the s1 / s2 loads of every pass are issued together -- pass by pass each paid its own round trip, ~0.7 us)
#pragma unroll
  for (int ps = 0; ps < EP_PASSES; ++ps) {
    const int it = tid + ps * NT;
    const int row = it / (BN / 8), c8 = it % (BN / 8);
    const int m = mbase + row, n = strip * BN + c8 * 8;
    if (it < EP_ITEMS && m < M && n < N) {
      const v4i lo = *reinterpret_cast<const v4i*>(ep + row * EP_STRIDE + c8 * 8);
      const v4i hi4 = *reinterpret_cast<const v4i*>(ep + row * EP_STRIDE + c8 * 8 + 4);
      h4 o0, o1;
      if constexpr (EP_PRE) {
        o0 = epilogue_vals4(lo[0], lo[1], lo[2], lo[3], a_pre[ps], s2_pre[0], s2_pre[1]);
        o1 = epilogue_vals4(hi4[0], hi4[1], hi4[2], hi4[3], a_pre[ps], s2_pre[2], s2_pre[3]);
      } else {
        const float a_s = s1[m];
        o0 = epilogue_vals4(lo[0], lo[1], lo[2], lo[3], n, a_s, s2);
        o1 = epilogue_vals4(hi4[0], hi4[1], hi4[2], hi4[3], n + 4, a_s, s2);
      }
      h8 o = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
      if (bias) o = o + *reinterpret_cast<const h8*>(bias + n);  // fp16 add after the fp16 round
      *reinterpret_cast<h8*>(D + (size_t)m * N + n) = o;
      if (acc_out) {
        *reinterpret_cast<v4i*>(acc_out + (size_t)m * N + n) = lo;
        *reinterpret_cast<v4i*>(acc_out + (size_t)m * N + n + 4) = hi4;
      }
    }
  }
#ifdef QQQ_PANEL_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  QQQ_TR(7);
  { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); QQQ_TRV(9, xcc & 15); }
#endif
}

#endif  // QQQ_AMD_QQQ_PANEL_HIP_H_
