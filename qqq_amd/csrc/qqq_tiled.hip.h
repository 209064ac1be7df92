// qqq_tiled.hip.h -- "tiled" kernel (m > 128, LDS-DMA staged 32x32x32 MFMA tiles, in-launch split-K)
// Part of the single translation unit qqq_w4a8.hip (see its header comment for the design).
#ifndef QQQ_AMD_QQQ_TILED_HIP_H_
#define QQQ_AMD_QQQ_TILED_HIP_H_

// ------------------------------------------------------------------------------------------
// "tiled" kernel: large m, LDS-staged, MFMA 32x32x32
// ------------------------------------------------------------------------------------------
//
// Workgroup tile = BM tokens x 256 weight columns (4 column groups ng0..ng0+3), BK = 128.
// MFMA A operand lane l = (i = l & 31, h = l >> 5): i = 8*g + c <-> columns
//     n = n0 + 64*g + 16*jt + 8*b + c,  k = 32*t + 16*h + [0,16)  (k-tile 2*t + h of the stage).
// MFMA B operand lane l = (j = l & 31, h): token m0 + 32*mtile + j, same k.
// MFMA D lane l: column j = l & 31 (token), rows i = (r & 3) + 8*(r >> 2) + 4*(l >> 5)
//     -> g = r >> 2, c = 4*h + (r & 3): 4 consecutive n per (jt, b, g).
// Waves: WM x WN; wave (wm, wn) owns tokens [32*MTW*wm, +32*MTW) and jt in [JW*wn, +JW).
//
// LDS stage: W region 8 k-tiles x 2048 B (raw packed words; 16-byte chunk (c, kq) of block
// (kt, g) is stored at chunk position 4*c + (kq ^ g): the four 16-lane groups of a
// ds_read_b128 then hit 16 distinct bank quads), X region BM rows x 128 B (chunk position
// p ^ ((row >> 1) & 7)).  With global_load_lds the LDS image is lane-linear, so the swizzle is
// applied to the per-lane SOURCE address; the register-staged variant writes the same image.

template <int BM, int MTW, int JW, int NB, bool GROUPED, int NS>
__global__ __launch_bounds__((BM / (32 * MTW)) * (4 / JW) * (2 / NB) * 64) void qqq_tiled_kernel(
    const int8_t* __restrict__ A, const unsigned char* __restrict__ B, int32_t* __restrict__ C,
    _Float16* __restrict__ D, const float* __restrict__ s1, const float* __restrict__ s2,
    const _Float16* __restrict__ s3, int32_t* __restrict__ acc_out,
    const _Float16* __restrict__ bias, const int M, const int N, const int K, const int ksplit_hf,
    const int tiles_m, const int tiles_n, int* __restrict__ tickets, const int nslots, const int PW) {
  // (hand-off switches ride in the upper half of the K-split argument -- tune.fused bits 2 / 3: 1 = the formal agent-scope ACQUIRE
  // fence in front of the fold, 2 = agent-scope RELEASE on the depositor's completion count; see qqq_common.hip.h)
  const int ksplit = ksplit_hf & 0xffff, hflags = ksplit_hf >> 16;
  // wave tile: MTW m-tiles of 32 tokens x JW column tiles (jt) x NB column halves (b).  NB == 1: the two
  // b halves of a packed word go to two different waves (per-group mode: every weight is re-quantised
  // by exactly one wave of the workgroup).
  static_assert(NB == 1 || NB == 2, "NB");
  static_assert(JW == 1 || JW == 2 || JW == 4, "JW");
  constexpr int WM = BM / (32 * MTW);
  constexpr int WN = (4 / JW) * (2 / NB);
  constexpr int NT = WM * WN * 64;
  constexpr int W_BYTES = 8 * 2048;
  constexpr int X_BYTES = BM * 128;
  // NS == 0: register-staged, 2 LDS buffers (simple reference variant)
  // NS >= 2: LDS-DMA (global_load_lds) ring of NS stages, NS-1 stages of loads in flight
  // NS == 5: LDS-DMA ring of 3 stages driven by the staggered two-group ("ping-pong") schedule
  constexpr bool GLDS = NS > 0;
  // NS == 6: LDS-DMA ring of 3 stages, ONE barrier per 128-k block placed two k-steps before the stage
  //          switch, fragment pipeline (W reads 2 steps ahead, unpack 1 step ahead) running across it
  constexpr bool PINGPONG = (NS == 5);
  constexpr bool CONTPIPE = (NS == 6 || NS == 7);
  constexpr bool ILV = (NS == 7);  // NS == 7: as 6, with the LDS fragment reads and the DMA issue spread between the MFMAs too
  constexpr int NSTAGE = (PINGPONG || CONTPIPE) ? 3 : NS;
  constexpr int SC_BYTES = (GLDS && GROUPED) ? WM * WN * 512 : 0;  // per-wave slot of group scales
  constexpr int STAGE = W_BYTES + X_BYTES + SC_BYTES;
  constexpr int W_CHUNKS = W_BYTES / 16;  // 1024
  constexpr int X_CHUNKS = X_BYTES / 16;
  constexpr int WPT = W_CHUNKS / NT;  // chunks per thread
  constexpr int XPT = X_CHUNKS / NT;
  static_assert(W_CHUNKS % NT == 0 && X_CHUNKS % NT == 0, "tile/threads mismatch");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = (wave % WN) / (2 / NB);    // which jt set
  const int bsel = (wave % WN) % (2 / NB);  // which b half (NB == 1 only; 0 otherwise)

  // ---- XCD-aware tile order: block b runs on XCD b % 8; give each XCD a contiguous run of the
  // panel-major tile sequence (panels of PW strips x all m-tiles) so co-resident workgroups of one
  // XCD share weight strips / activation rows in that XCD's L2.  Speed only, never correctness.
  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  int lin;
  {
    const int q = ntiles >> 3, rr = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    lin = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  int tile_m, tile_n;
  {
    const int full = (tiles_n / PW) * PW * tiles_m;
    if (lin < full) {
      const int panel = lin / (PW * tiles_m), within = lin % (PW * tiles_m);
      tile_m = within / PW;
      tile_n = panel * PW + within % PW;
    } else {
      const int rem = lin - full, pw = tiles_n % PW;
      tile_m = rem / pw;
      tile_n = (tiles_n / PW) * PW + rem % pw;
    }
  }
  const int sp = blockIdx.y;
  const int m0 = tile_m * BM;
  const int ng0 = tile_n * 4;
  const int ngroups = N >> 6;

  const int NKB = K >> 7;  // 128-k blocks
  const int kb_begin = (int)(((long long)NKB * sp) / ksplit);
  const int kb_end = (int)(((long long)NKB * (sp + 1)) / ksplit);
  const size_t rowbytes = (size_t)N * 8;

  // ---- per-thread staging sources (offsets relative to the stage's first k-tile / k byte) ----
  unsigned wsrc[WPT], xsrc[XPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int cw = tid + i * NT;  // LDS chunk index inside the W region
    const int ktl = cw >> 7, gl = (cw >> 5) & 3, pos = cw & 31;
    const int cc = pos >> 2, kq = (pos & 3) ^ gl;
    int ngx = ng0 + gl;
    if (ngx >= ngroups) ngx = ngroups - 1;
    wsrc[i] = (unsigned)(ktl * rowbytes + (size_t)ngx * 512 + (4 * cc + kq) * 16);
  }
#pragma unroll
  for (int i = 0; i < XPT; ++i) {
    const int cx = tid + i * NT;
    const int row = cx >> 3, pos = cx & 7;
    const int chunk = pos ^ ((row >> 1) & 7);
    int grow = m0 + row;
    if (grow >= M) grow = M - 1;
    xsrc[i] = (unsigned)((size_t)(grow - m0) * K + chunk * 16);
  }
  const unsigned char* Abase = reinterpret_cast<const unsigned char*>(A) + (size_t)m0 * K;

  // ---- per-lane LDS read offsets ----
  const int li = lane & 31, h = lane >> 5;
  const int g = li >> 3, c = li & 7;
  // Which jt of the 16-byte chunk this lane consumes.  A lane reads 8 of the 16 bytes (JW <= 2); lanes
  // c < 4 take one half, lanes c >= 4 the OTHER half: the 32 lanes of a ds_read_b64 group then cover
  // all 64 banks exactly once (a fixed half would be a 2-way conflict).  This is only another
  // row<->column permutation of the MFMA tile (a sibling wave takes the complementary halves); the
  // epilogue and the group-scale addresses use the same lane-dependent jt.
  //   JW == 2: half = wn ^ (c >> 2), jt = 2*half + jj
  //   JW == 1: half = (wn >> 1) ^ (c >> 2), jt = 2*half + (wn & 1)   (8 bytes read, one dword used)
  const int half = (JW == 2) ? (wn ^ (c >> 2)) : (JW == 1) ? ((wn >> 1) ^ (c >> 2)) : 0;
  const int esel = (JW == 1) ? (wn & 1) : 0;
  const int jt0 = (JW == 4) ? 0 : 2 * half + esel;  // first jt of this lane
  unsigned wrd[4];  // + t*4096
#pragma unroll
  for (int kq = 0; kq < 4; ++kq)
    wrd[kq] = h * 2048 + g * 512 + (4 * c + (kq ^ g)) * 16 + ((JW == 4) ? 0 : half * 8);
  unsigned xrd[4];  // per k-step t; + mt*32*128
#pragma unroll
  for (int t = 0; t < 4; ++t)
    xrd[t] = W_BYTES + (wm * MTW * 32 + li) * 128 + (((2 * t + h) ^ ((li >> 1) & 7)) * 16);

  const _Float16* sptr = nullptr;  // register path (NS == 0)
  unsigned scsrc = 0;              // LDS-DMA path: byte offset of this lane's 16 B inside the group row
  unsigned scrd = 0;               // LDS byte offset (inside the stage) of this lane's 2*JW scales
  if constexpr (GROUPED) {
    int ngx = ng0 + g;
    if (ngx >= ngroups) ngx = ngroups - 1;
    sptr = s3 + (size_t)ngx * 64 + c * 8 + 2 * jt0;
    int ngl = ng0 + ((lane & 31) >> 3);
    if (ngl >= ngroups) ngl = ngroups - 1;
    scsrc = (unsigned)((ngl * 64 + (lane & 7) * 8) * 2);
    scrd = W_BYTES + X_BYTES + wave * 512 + (g * 64 + c * 8 + 2 * jt0) * 2;
  }

  v16i acc[MTW][JW][NB];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int jj = 0; jj < JW; ++jj)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][jj][b][r] = 0;

  v4u wreg[WPT], xreg[XPT];  // register staging (unused with GLDS)

  // part / nparts: issue only the part-th of nparts contiguous groups of this thread's DMA instructions
  // (compile-time constants at every call site); the default issues the whole stage.
  auto issue_loads = [&](const int kb, const int buf, const int part = 0, const int nparts = 1) {
    const unsigned char* wb = B + (size_t)(kb * 8) * rowbytes;
    const unsigned char* xb = Abase + (size_t)kb * 128;
    if constexpr (GLDS) {
      const unsigned st = lds_base + buf * STAGE + wave * 1024;
      const int lo = (WPT + XPT) * part / nparts, hi = (WPT + XPT) * (part + 1) / nparts;
#pragma unroll
      for (int i = 0; i < WPT; ++i)
        if (i >= lo && i < hi) glds16(wb + wsrc[i], st + i * (NT * 16));
#pragma unroll
      for (int i = 0; i < XPT; ++i)
        if (WPT + i >= lo && WPT + i < hi) glds16(xb + xsrc[i], st + W_BYTES + i * (NT * 16));
      if (part != nparts - 1) return;
      if constexpr (GROUPED) {
        // this wave's private copy of the tile's 256 group scales (512 B): lanes 0..31, 16 B each
        const unsigned scdst =
            __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE + W_BYTES + X_BYTES + wave * 512);
        if (lane < 32)  // exec-masked DMA: only 32 x 16 B are written
          glds16(reinterpret_cast<const unsigned char*>(s3 + (size_t)kb * N) + scsrc, scdst);
      }
    } else {
#pragma unroll
      for (int i = 0; i < WPT; ++i) wreg[i] = *reinterpret_cast<const v4u*>(wb + wsrc[i]);
#pragma unroll
      for (int i = 0; i < XPT; ++i) xreg[i] = *reinterpret_cast<const v4u*>(xb + xsrc[i]);
    }
  };
  auto commit_loads = [&](const int buf) {  // register-staged variant: write the LDS image
    if constexpr (!GLDS) {
      unsigned char* st = smem + buf * STAGE;
#pragma unroll
      for (int i = 0; i < WPT; ++i) *reinterpret_cast<v4u*>(st + (tid + i * NT) * 16) = wreg[i];
#pragma unroll
      for (int i = 0; i < XPT; ++i)
        *reinterpret_cast<v4u*>(st + W_BYTES + (tid + i * NT) * 16) = xreg[i];
    }
  };

  // per-group scales of this lane's (jt in [JW*wn, +JW), b) columns: 2*JW consecutive fp16
  typedef _Float16 hsc __attribute__((ext_vector_type(2 * JW)));
  hsc sc_cur = {}, sc_nxt = {};
  if constexpr (GROUPED)
    if (kb_begin < kb_end) sc_cur = *reinterpret_cast<const hsc*>(sptr + (size_t)kb_begin * N);

  // One k-step (32 k) of fragments: raw packed weight words [kq][jj] (only this wave's jt are read
  // from LDS) + the activation operands of this wave's m-tiles.
  struct Frag {
    unsigned wq[4][JW];
    v4i xop[MTW];
  };
  auto read_frag = [&](const unsigned char* st, const int t, Frag& f) {
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      const unsigned char* p = st + wrd[kq] + t * 4096;
      if constexpr (JW == 4) {
        const v4u v = *reinterpret_cast<const v4u*>(p);
        f.wq[kq][0] = v[0]; f.wq[kq][1] = v[1]; f.wq[kq][2] = v[2]; f.wq[kq][3] = v[3];
      } else if constexpr (JW == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        f.wq[kq][0] = v.x; f.wq[kq][1] = v.y;
      } else {
        const uint2 v = *reinterpret_cast<const uint2*>(p);  // conflict-free b64, one dword used
        f.wq[kq][0] = esel ? v.y : v.x;
      }
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
      f.xop[mt] = *reinterpret_cast<const v4i*>(st + xrd[t] + mt * (32 * 128));
  };
  // unpacked MFMA weight operands of one k-step
  struct Ops {
    v4i a[JW][NB];
  };
  auto unpack_frag = [&](const Frag& f, Ops& o) {
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      h2 sb0 = {(_Float16)0, (_Float16)0}, sb1 = sb0;
      if constexpr (GROUPED) {
        sb0 = (h2){sc_cur[2 * jj], sc_cur[2 * jj]};
        sb1 = (h2){sc_cur[2 * jj + 1], sc_cur[2 * jj + 1]};
      }
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        const unsigned q = f.wq[kq][jj];
        if constexpr (NB == 2) {
          int w0, w1;
          unpack_pair<GROUPED>(q, sb0, sb1, w0, w1);
          o.a[jj][0][kq] = w0;
          o.a[jj][1][kq] = w1;
        } else if constexpr (GROUPED) {  // one b half per wave (bsel is wave-uniform)
          o.a[jj][0][kq] = (int)dequant_group4(q >> (8 * bsel), bsel ? sb1 : sb0);
        } else {
          o.a[jj][0][kq] = (int)((q << (4 * bsel)) & QQQ_NIB_MASK);
        }
      }
    }
  };
  auto mfma_ops = [&](const Ops& o, const Frag& f) {
#pragma unroll
    for (int jj = 0; jj < JW; ++jj)
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
          acc[mt][jj][bi] = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.a[jj][bi], f.xop[mt], acc[mt][jj][bi], 0, 0, 0);
        }
  };
  auto mma_frag = [&](const Frag& f) {
    Ops o;
    unpack_frag(f, o);
    mfma_ops(o, f);
  };
  // software pipeline inside a stage: the LDS reads of k-step t+1 are issued before the MFMAs of
  // k-step t, so their latency hides under the matrix pipe instead of stalling the (in-order) wave
  auto compute_stage = [&](const int buf) {
    const unsigned char* st = smem + buf * STAGE;
    Frag f0 = {}, f1 = {};
    read_frag(st, 0, f0);
    read_frag(st, 1, f1);
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(f0);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(st, 2, f0);
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(f1);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(st, 3, f1);
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(f0);
    mma_frag(f1);
  };

  if constexpr (!GLDS) {
    // ---- register-staged variant: 2 LDS buffers, one barrier per 128-k block ----
    if (kb_begin < kb_end) {
      issue_loads(kb_begin, 0);
      commit_loads(0);
      __syncthreads();
    }
    for (int kb = kb_begin; kb < kb_end; ++kb) {
      const int buf = (kb - kb_begin) & 1;
      const bool more = (kb + 1 < kb_end);
      if (more) {
        issue_loads(kb + 1, buf ^ 1);
        if constexpr (GROUPED) sc_nxt = *reinterpret_cast<const hsc*>(sptr + (size_t)(kb + 1) * N);
      }
      compute_stage(buf);
      if (more) commit_loads(buf ^ 1);
      if constexpr (GROUPED) sc_cur = sc_nxt;
      __syncthreads();
    }
  } else {
    // ---- LDS-DMA ring: NS stages, NS-1 stages of loads in flight, one barrier per 128-k block.
    // The DMA loads are inline asm (invisible to hipcc's counters): we count them ourselves.  Every
    // thread issues exactly NL wave-instructions per stage, in stage order, so "stage j has landed"
    // == "at most NL * (number of younger stages issued) loads outstanding".
    constexpr int NL = WPT + XPT + (GROUPED ? 1 : 0);
    static_assert(NL * (NSTAGE - 2) <= 63, "vmcnt is a 6-bit counter");
    auto wait_younger = [&](const int younger) {  // wave-uniform
      if (younger <= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (younger == 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
      } else if (younger == 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL > 63 ? 63 : 2 * NL) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NL > 63 ? 63 : 3 * NL) : "memory");
      }
    };
    const int nkb = kb_end - kb_begin;
    if constexpr (CONTPIPE) {
      // ---- continuous fragment pipeline over a 3-stage DMA ring.
      // step u = 4*i + t:   ds_read W(u+2), X(u+1)  |  unpack W(u+1) -> ops  |  8 MFMA(u)
      // The only barrier of block i sits at the start of its step t=2: by then every wave has waited for
      // its DMA of stage i+1 (issued one block earlier), so from t=2 on stage i+1 may be read; stage i+2
      // is issued right behind that barrier into the buffer of stage i-1 (last read at (i-1, t=1)).
      // Reads past the last step are redirected to the last stage (harmless), never branched around, so
      // each step is one straight-line scheduling region with an enforced MFMA/VALU/DS interleave.
      unsigned wraw[2][4][JW];
      v4i xfr[2][MTW];
      Ops ops2[2];
      auto read_w = [&](const unsigned char* st, const int t, unsigned (&w)[4][JW]) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          const unsigned char* p = st + wrd[kq] + t * 4096;
          if constexpr (JW == 4) {
            const v4u v = *reinterpret_cast<const v4u*>(p);
            w[kq][0] = v[0]; w[kq][1] = v[1]; w[kq][2] = v[2]; w[kq][3] = v[3];
          } else if constexpr (JW == 2) {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
            w[kq][0] = v.x; w[kq][1] = v.y;
          } else {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
            w[kq][0] = esel ? v.y : v.x;
          }
        }
      };
      auto read_x = [&](const unsigned char* st, const int t, v4i (&x)[MTW]) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) x[mt] = *reinterpret_cast<const v4i*>(st + xrd[t] + mt * (32 * 128));
      };
      auto unpack_w = [&](const unsigned (&w)[4][JW], Ops& o) {
        Frag f;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
#pragma unroll
          for (int jj = 0; jj < JW; ++jj) f.wq[kq][jj] = w[kq][jj];
        unpack_frag(f, o);
      };
      if (nkb > 0) {
        issue_loads(kb_begin, 0);
        if (nkb > 1) issue_loads(kb_begin + 1, 1);
        wait_younger(nkb > 1 ? 1 : 0);
        __syncthreads();
        if constexpr (GROUPED) sc_cur = *reinterpret_cast<const hsc*>(smem + scrd);
        read_w(smem, 0, wraw[0]);
        read_x(smem, 0, xfr[0]);
        read_w(smem, 1, wraw[1]);
        unpack_w(wraw[0], ops2[0]);
      }
      for (int i = 0; i < nkb; ++i) {
        const unsigned char* st = smem + (i % 3) * STAGE;
        const bool has_next = (i + 1 < nkb);
        const unsigned char* stn = has_next ? smem + ((i + 1) % 3) * STAGE : st;  // redirect past-the-end reads
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t == 2 && has_next) {
            wait_younger(0);  // this wave's DMA of stage i+1 (the only one in flight) has landed
            __syncthreads();  // ... and everybody else's
          }
          if constexpr (!ILV) {
            // DMA issue of a stage spread over the 3 k-steps behind the barrier (2 of the wave's 6 instructions
            // each): a burst of all 8 waves x 6 KB right behind the barrier overruns the address pipe (~30
            // cycles per 1 KB instruction, measured with s_memtime stamps), and an in-order wave stuck in VMEM issue
            // cannot issue its MFMAs either.  -3 % cycles, 1-3 % time on real data.
            constexpr int SPREAD = 3;
            const int part = (t + 2) & 3;  // t=2 -> 0, t=3 -> 1, t=0 -> 2, t=1 -> 3
            const int stg = (t >= 2) ? i + 2 : i + 1;
            if (part < SPREAD && stg < nkb && stg >= 2) issue_loads(kb_begin + stg, stg % 3, part, SPREAD);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!ILV) {
            // fragment reads: W two steps ahead, X one step ahead
            read_w((t + 2 < 4) ? st : stn, (t + 2) & 3, wraw[t & 1]);
            read_x((t + 1 < 4) ? st : stn, (t + 1) & 3, xfr[(t + 1) & 1]);
            if constexpr (GROUPED)
              if (t == 3) sc_cur = *reinterpret_cast<const hsc*>(stn + scrd);  // scales of the block being unpacked
            __builtin_amdgcn_sched_barrier(0);
          } else if constexpr (GROUPED) {
            if (t == 3) sc_cur = *reinterpret_cast<const hsc*>(stn + scrd);  // scales of the block being unpacked
            __builtin_amdgcn_sched_barrier(0);
          }
          // hand-interleaved issue order: one MFMA of step u, then the unpack of one packed word of step
          // u+1 (hipcc otherwise issues the 8 MFMAs back to back and leaves the VALU work uncovered)
          constexpr int NM = MTW * JW * NB;  // MFMAs per step
          constexpr int NWD = 4 * JW;        // packed words per step
#pragma unroll
          for (int q = 0; q < (NM > NWD ? NM : NWD); ++q) {
            if (q < NM) {
              const int jj = q / (MTW * NB), mt = (q / NB) % MTW, bi = q % NB;
              acc[mt][jj][bi] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops2[t & 1].a[jj][bi], xfr[t & 1][mt],
                                                                      acc[mt][jj][bi], 0, 0, 0);
            }
            if constexpr (ILV) {
              // one LDS fragment read (W two steps ahead: 4 pieces, then X one step ahead: MTW pieces) or one DMA
              // instruction behind every MFMA instead of a burst in front of the step
              static_assert(!ILV || (JW == 2 && 4 + MTW + 2 <= MTW * JW * NB), "interleave slots");
              if (q < 4) {
                const unsigned char* p = ((t + 2 < 4) ? st : stn) + wrd[q] + ((t + 2) & 3) * 4096;
                const uint2 v = *reinterpret_cast<const uint2*>(p);
                wraw[t & 1][q][0] = v.x;
                wraw[t & 1][q][1] = v.y;
              } else if (q < 4 + MTW) {
                xfr[(t + 1) & 1][q - 4] =
                    *reinterpret_cast<const v4i*>(((t + 1 < 4) ? st : stn) + xrd[(t + 1) & 3] + (q - 4) * (32 * 128));
              } else if (q < 4 + MTW + 2) {
                const int part = (t + 2) & 3;  // t=2 -> 0, t=3 -> 1, t=0 -> 2: a third of the stage per k-step
                const int stg = (t >= 2) ? i + 2 : i + 1;
                if (part < 3 && stg < nkb && stg >= 2) issue_loads(kb_begin + stg, stg % 3, 2 * part + (q - 4 - MTW), 6);
              }
            }
            if (q < NWD) {
              const int kq = q & 3, jj = q >> 2;
              const unsigned wq_ = wraw[(t + 1) & 1][kq][jj];
              h2 sb0 = {(_Float16)0, (_Float16)0}, sb1 = sb0;
              if constexpr (GROUPED) {
                sb0 = (h2){sc_cur[2 * jj], sc_cur[2 * jj]};
                sb1 = (h2){sc_cur[2 * jj + 1], sc_cur[2 * jj + 1]};
              }
              // the empty asm pins the unpack HERE (LLVM would otherwise sink it to its use, one step later)
              if constexpr (NB == 2) {
                int w0, w1;
                unpack_pair<GROUPED>(wq_, sb0, sb1, w0, w1);
                asm volatile("" : "+v"(w0), "+v"(w1));
                ops2[(t + 1) & 1].a[jj][0][kq] = w0;
                ops2[(t + 1) & 1].a[jj][1][kq] = w1;
              } else if constexpr (GROUPED) {
                int w0 = (int)dequant_group4(wq_ >> (8 * bsel), bsel ? sb1 : sb0);
                asm volatile("" : "+v"(w0));
                ops2[(t + 1) & 1].a[jj][0][kq] = w0;
              } else {
                int w0 = (int)((wq_ << (4 * bsel)) & QQQ_NIB_MASK);
                asm volatile("" : "+v"(w0));
                ops2[(t + 1) & 1].a[jj][0][kq] = w0;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    } else if constexpr (!PINGPONG) {
      int issued = 0;  // stages issued so far (relative index)
      for (; issued < NSTAGE - 1 && issued < nkb; ++issued) issue_loads(kb_begin + issued, issued % NSTAGE);
      if (nkb > 0) {
        wait_younger(issued - 1);
        __syncthreads();
      }
      for (int i = 0; i < nkb; ++i) {
        // refill the buffer consumed in iteration i-1 (every wave has passed that iteration's barrier)
        if (issued < nkb) {
          issue_loads(kb_begin + issued, issued % NSTAGE);
          ++issued;
        }
        if constexpr (GROUPED)
          sc_cur = *reinterpret_cast<const hsc*>(smem + (i % NSTAGE) * STAGE + scrd);
        compute_stage(i % NSTAGE);
        if (i + 1 < nkb) {
          wait_younger(issued - (i + 2));  // stage i+1 must have landed; stages i+2.. may stay in flight
          __syncthreads();
        }
      }
    } else {
      // ---- staggered two-group schedule ("ping-pong").  Every k-step is cut in a LOAD phase (LDS
      // fragment reads for the NEXT k-step, int4 unpack of the current one, DMA issue / landing waits)
      // and an MFMA phase (8 matrix instructions), each closed by a workgroup barrier.  The second half
      // of the waves runs one phase behind the first, so on every SIMD one wave's MFMA burst overlaps
      // its partner's LOAD phase instead of both stalling on LDS at the same time.
      //   DMA of stage j is issued in LOAD(j-2, t=0) into buffer j % 3 (last read two barriers ago);
      //   its landing is awaited by every wave in LOAD(j-1, t=2); first read is in LOAD(j-1, t=3).
      const bool late = wave >= (WM * WN) / 2;  // wave-uniform
      auto phase_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);  // register-only MFMAs would otherwise drift across it
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      Frag fr[2] = {};
      Ops ops;
      if (nkb > 0) {
        issue_loads(kb_begin, 0);
        if (nkb > 1) issue_loads(kb_begin + 1, 1);
        wait_younger(nkb > 1 ? 1 : 0);
        __syncthreads();
        read_frag(smem, 0, fr[0]);
        if (late) phase_barrier();
      }
      for (int i = 0; i < nkb; ++i) {
        const unsigned char* st = smem + (i % 3) * STAGE;
        const unsigned char* stn = smem + ((i + 1) % 3) * STAGE;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // ---------------- LOAD phase ----------------
          if (t == 0) {
            if (i + 2 < nkb) issue_loads(kb_begin + i + 2, (i + 2) % 3);
            if constexpr (GROUPED) sc_cur = *reinterpret_cast<const hsc*>(st + scrd);
          }
          if (t == 2 && i + 1 < nkb) wait_younger(i + 2 < nkb ? 1 : 0);
          if (t < 3) {
            read_frag(st, t + 1, fr[(t + 1) & 1]);
          } else if (i + 1 < nkb) {
            read_frag(stn, 0, fr[0]);
          }
          unpack_frag(fr[t & 1], ops);
          phase_barrier();
          // ---------------- MFMA phase ----------------
          __builtin_amdgcn_s_setprio(1);  // the partner wave on this SIMD is in its LOAD phase: MFMA issue first
          mfma_ops(ops, fr[t & 1]);
          __builtin_amdgcn_s_setprio(0);
          phase_barrier();
        }
      }
      if (nkb > 0 && !late) phase_barrier();  // both groups execute the same number of barriers
    }
  }

  // ---- in-launch split-K (nslots > 0): the K slices of one tile meet in `nslots` tile-sized int32 slots of C.
  // Workgroups take a ticket in ARRIVAL order.  Arrival t < ksplit-1 deposits its partial tile in slot
  // t % nslots -- adding what the slot already holds when it is the slot's (t / nslots + 1)-th deposit --
  // and leaves; the last arrival adds every slot to its accumulators and runs the normal epilogue.
  // Whoever is waited for has already arrived (is past its main loop), so the spins are short and cannot
  // deadlock whatever the dispatch order.  Slot images are lane-linear (16 B per lane, full lines); they
  // are written through (sc0 sc1) and read with agent-scope (sc1) loads, because the K slices of a tile
  // run on different XCDs (= different, mutually non-coherent L2s).  int32 adds commute: bit-exact.
  // tickets[(1 + nslots) * tile + {0: arrivals, 1 + s: deposits completed in slot s}], all zero again on exit.
  bool finish = (ksplit == 1);
  if (ksplit > 1 && nslots > 0) {
    constexpr int NQ4 = MTW * JW * NB * 4;
    __shared__ int xch;
    int* tk = tickets + (size_t)lin * (1 + nslots);
    if (tid == 0) xch = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int t = xch;
    const unsigned voff = lane * 16;  // lane offset inside a 1 KiB wave-row of a slot image
    auto slot_base = [&](const int s_) {  // wave-uniform: lets the accesses use the SGPR-base + VGPR-offset form
      return reinterpret_cast<const unsigned char*>(C + ((size_t)s_ * ntiles + lin) * ((size_t)BM * 256) +
                                                    (size_t)wave * NQ4 * 256);
    };
    auto wait_done = [&](const int s_, const int want) {
      if (tid == 0) {
        int spin = 0;
        while (__hip_atomic_load(tk + 1 + s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          if (++spin > QQQ_SPIN_LIMIT) __builtin_trap();  // never a silently wrong D: abort the launch (see the panel kernel)
          __builtin_amdgcn_s_sleep(4);
        }
      }
      __syncthreads();  // no acquire fence (an L2-wide invalidate at agent scope): add_slot reads with agent-scope loads
      if (qqq_formal_acquire(hflags)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (off by default: qqq_common.hip.h)
    };
    auto add_slot = [&](const int s_) {
      const __amdgpu_buffer_rsrc_t view = agent_view(slot_base(s_));
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) {
#pragma unroll
          for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const v4i v = load16_agent(view, (unsigned)((((mt * JW + jj) * NB + bi) * 4 + gq) * 1024) + voff);
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[mt][jj][bi][4 * gq + r] += v[r];
            }
          __builtin_amdgcn_sched_barrier(0);  // at most 4*NB loads in flight: no spill next to 128 accumulators
        }
    };
    // one loop for both roles (a depositor folds at most its own slot, the last arrival every used slot)
    const bool last = (t == ksplit - 1);
    const int slot = last ? 0 : t % nslots, gen = last ? 0 : t / nslots;
    const int used = (ksplit - 1 < nslots) ? ksplit - 1 : nslots;
    const int nfold = last ? used : (gen > 0 ? 1 : 0);
    for (int i = 0; i < nfold; ++i) {
      const int s_ = slot + i;
      wait_done(s_, last ? (ksplit - 1 - s_ + nslots - 1) / nslots : gen);
      add_slot(s_);
    }
    if (!last) {
      // scalar base + one 32-bit lane offset: the 4*MTW*JW*NB stores share a single address VGPR
      const unsigned char* sbase = slot_base(slot);
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int jj = 0; jj < JW; ++jj)
#pragma unroll
          for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const v4i v = {acc[mt][jj][bi][4 * gq + 0], acc[mt][jj][bi][4 * gq + 1], acc[mt][jj][bi][4 * gq + 2],
                             acc[mt][jj][bi][4 * gq + 3]};
              const unsigned char* sb = sbase + (((mt * JW + jj) * NB + bi) * 4 + gq) * 1024;
              // s_nop 1 behind: a 16-byte store reads its data registers late and hipcc would let its next VALU overwrite them;
              // s_nop 4 in front: the scalar base may come straight from a v_readfirstlane (5 wait states before a VMEM read)
              asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sb) : "memory");
            }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // every wave's part of the deposit has reached memory
      if (tid == 0) {
        if (hflags & 2) __hip_atomic_store(tk + 1 + slot, gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(tk + 1 + slot, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    if (tid <= used) __hip_atomic_store(tk + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // workspace zero on return
    finish = true;
  }

  // ---- epilogue ----
  int mrow[MTW];
  float a_s[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    mrow[mt] = m0 + (wm * MTW + mt) * 32 + li;
    a_s[mt] = (mrow[mt] < M && finish) ? s1[mrow[mt]] : 0.f;
  }
  const int n_lane = ng0 * 64 + 4 * h;  // + 64*g' + 16*jt + 8*b  (g' = r >> 2), + (r & 3)
  if (finish) {
    // fp16 tile -> LDS (row-major, 16-byte chunks XOR-swizzled by the row so that the 8-byte writes of a
    // lane column and the 16-byte reads of a row are both conflict-light) -> full 128-byte-line stores.
    // Straight-from-register stores would be 8 bytes per lane scattered over 32 rows (measured: 2.6x write
    // amplification at the fabric, store-issue bound tail).
    __syncthreads();  // every wave is done reading the operand ring
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      // D-tile rows of this lane have c = 4*h + (r & 3): c >> 2 == h, so the lane's jt is uniform over r
      const int jt = (JW == 4) ? jj : (JW == 2) ? 2 * (wn ^ h) + jj : 2 * ((wn >> 1) ^ h) + (wn & 1);
#pragma unroll
      for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int nl = 4 * h + 64 * gq + 16 * jt + 8 * ((NB == 2) ? bi : bsel);  // column inside the tile
          const int n = ng0 * 64 + nl;
          const int ncl = (n < N) ? n : 0;  // scales of a dropped column: any legal address
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            const int ml = (wm * MTW + mt) * 32 + li;
            const int v0 = acc[mt][jj][bi][4 * gq + 0], v1 = acc[mt][jj][bi][4 * gq + 1];
            const int v2 = acc[mt][jj][bi][4 * gq + 2], v3 = acc[mt][jj][bi][4 * gq + 3];
            const h4 o = epilogue_vals4(v0, v1, v2, v3, ncl, a_s[mt], s2);
            *reinterpret_cast<h4*>(smem + ml * 512 + (((nl >> 3) ^ (ml & 31)) << 4) + ((nl & 7) << 1)) = o;
            if (acc_out && n < N && mrow[mt] < M) {
              v4i a = {v0, v1, v2, v3};
              *reinterpret_cast<v4i*>(acc_out + (size_t)mrow[mt] * N + n) = a;
            }
          }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < BM * 32; idx += NT) {
      const int ml = idx >> 5, j = idx & 31;
      const int m = m0 + ml, n = ng0 * 64 + j * 8;
      if (m < M && n < N) {
        h8 v = *reinterpret_cast<const h8*>(smem + ml * 512 + ((j ^ (ml & 31)) << 4));
        if (bias) v = v + *reinterpret_cast<const h8*>(bias + n);  // fp16 add after the fp16 round
        *reinterpret_cast<h8*>(D + (size_t)m * N + n) = v;
      }
    }
    return;
  }
  // split-K: int32 partial sums straight from the accumulators into slab sp of C
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    const int jt = (JW == 4) ? jj : (JW == 2) ? 2 * (wn ^ h) + jj : 2 * ((wn >> 1) ^ h) + (wn & 1);
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = n_lane + 64 * gq + 16 * jt + 8 * ((NB == 2) ? bi : bsel);
        if (n >= N) continue;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const int m = mrow[mt];
          if (m >= M) continue;
          v4i v = {acc[mt][jj][bi][4 * gq + 0], acc[mt][jj][bi][4 * gq + 1], acc[mt][jj][bi][4 * gq + 2],
                   acc[mt][jj][bi][4 * gq + 3]};
          *reinterpret_cast<v4i*>(C + ((size_t)sp * M + m) * N + n) = v;
        }
      }
  }
}


#endif  // QQQ_AMD_QQQ_TILED_HIP_H_
