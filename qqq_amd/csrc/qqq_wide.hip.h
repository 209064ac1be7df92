// qqq_wide.hip.h -- "wide" kernel (round 3): 256 tokens x 256 columns per workgroup, FOUR waves -- one per SIMD, 512 registers
// each (256 int32 accumulators in the accumulation half of the unified register file).  Large m.
// Part of the single translation unit qqq_w4a8.hip (see its header comment for the design).
#ifndef QQQ_AMD_QQQ_WIDE_HIP_H_
#define QQQ_AMD_QQQ_WIDE_HIP_H_

// ------------------------------------------------------------------------------------------
// Why this shape.  In the panel kernel's large-m shape (128 tokens x 256 columns, 8 waves = 4 column waves x 2 k-groups) a
// wave owns 64 columns x 128 tokens: every unpacked weight operand (and in the per-group mode every RE-QUANTISED one) feeds
// 8 m-tiles, every activation fragment read from LDS feeds 4 MFMAs, and the 128 accumulators fill half of the 256 registers
// two waves per SIMD leave each wave.  The per-SIMD register file holds 512 per lane: ONE wave per SIMD can keep 256
// accumulators (16 m-tiles x 4 column sets) next to 256 working registers.  A wave then owns 64 columns x 256 tokens:
//   * unpack / re-quantise VALU per MFMA halves (per-channel 1.5 -> 0.75, per-group 5.1 -> 2.4 -- the per-group mode was
//     VALU-issue bound at 128 tokens per weight operand: "re-quantise once, multiply many");
//   * 256 x 256 tiles: the L2 <-> fabric traffic of the tiled kernel (2 rounds x 8 XCDs x (4 m-tiles + 8 strips)), not the
//     128-row m-blocks' 4 rounds (741 MB per launch at M=4096 against 1137, profiles/r03_pmc_wide_m4096.txt);
//   * no k-group meet at the end, one barrier per 128-k stage between FOUR waves.
// The price: nothing hides a wave's own stalls (no partner on the SIMD) and an in-order wave issues one instruction per ~4
// cycles, i.e. a 16-cycle MFMA leaves room for at most three others.  (Round 6 MEASURED it -- tools/mfma_issue_bench.hip: ~5.2 cycles
// per instruction of any kind, TWO free behind an MFMA, nothing given back by an empty slot -- and rebuilt the loop against that: see
// QQQ_WIDE_BALANCE / QQQ_WIDE_CURSORS / QQQ_WIDE_DWORD below; with DWORD the weights arrive as one-word loads and the quad transpose
// described next is gone from the shipped loop.)  So:
//   * the accumulators are updated IN PLACE by inline-asm MFMAs ("+a"): with the builtin hipcc selects the untied form in the
//     accumulation registers and, all 256 of them live, bounces accumulators through VGPRs and scratch;
//   * the issue order is pinned slot by slot (sched_barrier): slot k of a 64-k step = the MFMA of (m-tile k / 4, column set
//     k % 4) + its share of everything else -- the NEXT step's weights (HBM -> VGPR ring, RS steps ahead) are transposed
//     (4 x 4 quad transpose as four 3-instruction pieces) and unpacked into a second operand set over the whole step, every
//     activation fragment is re-read from LDS for the next step right behind its fourth MFMA, and the activations are staged
//     by LDS-DMA, one 16-byte chunk per lane every 16 slots (see "LDS-DMA staging" below; at first global -> VGPR -> ds_write).
//     Everything is spread EVENLY: the four waves of a workgroup run in lock step (one barrier per stage), so anything issued
//     in a burst hits the LDS / the vector-memory issue four times at once (the first version staged in a burst and ran 7-9 %
//     slower, profiles/r03_wide_uniform_schedule.txt).
// MT = 16: 256-token tiles; MT = 8: 128-token tiles (128 accumulators); HW = 1: 128-column tiles, a wave owns 32 columns.
//
// grid = tiles_m * tiles_n * ksplit (XCD-aware order as in the tiled kernel), block = 256.  The host picks the shape that
// fills about a round of the chip (and at most two K slices).  K % 128 == 0 (the host sends other K to the panel kernel).
// Addresses: wave-uniform buffer descriptors + 32-bit lane offsets + scalar step offsets (buffer_load ... v_off, s[rsrc],
// s_off): no 64-bit per-lane address arithmetic on the issue port the MFMAs share.
// ------------------------------------------------------------------------------------------

// Measurement only (tools/ablate_wide.sh): -DQQQ_WIDE_ABLATE=<bits> removes parts of the steady-state loop -- 1 stage-end
// barrier, 2 activation staging, 4 transpose + unpack VALU, 8 weight-ring refill, 16 LDS fragment reads.  Results are wrong
// by construction; never defined in a shipped build.
#ifndef QQQ_WIDE_ABLATE
#define QQQ_WIDE_ABLATE 0
#endif
// Phase clocks of the measurement build (-DQQQ_PANEL_TRACE, see qqq_panel.hip.h), written by WAVE 0 of the workgroup -- all 64
// lanes store the same word: a lane-0 guard is a divergent branch, and behind one hipcc no longer treats the tile walk's
// scalars as scalars (an "s" asm operand then fails to compile).  `wn` is the wave index (wave-uniform).
#ifdef QQQ_PANEL_TRACE
#define QQQ_WTRV(i, v)                                                                                   \
  do {                                                                                                   \
    if (wn == 0 && qqq_trace_buf) qqq_trace_buf[(size_t)blockIdx.x * 16 + (i)] = (unsigned long long)(v); \
  } while (0)
#define QQQ_WTR(i) QQQ_WTRV(i, wall_clock64())
#else
#define QQQ_WTR(i) do {} while (0)
#define QQQ_WTRV(i, v) do {} while (0)
#endif
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order: compile-time indices for hand-placed code
template <int... S, class F>
__device__ __forceinline__ void qqq_static_for(std::integer_sequence<int, S...>, F&& f) {
  (f(std::integral_constant<int, S>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void qqq_static_for(F&& f) {
  qqq_static_for(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wide_view(const void* base_uniform) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base_uniform), 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ v4u wide_load16(__amdgpu_buffer_rsrc_t view, const unsigned voff, const unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(view, voff, soff, 0);
}

// Measurement / tuning: QQQ_WIDE_SLOTMAP=1 keeps the unpack items of the per-channel 256-token shape out of the slots that
// already carry a memory instruction (fragment re-read, staging write / reload, ring refill): a memory instruction takes
// more than one issue slot, and with a VALU item behind it the slot overruns its MFMA's 16 cycles.
#ifndef QQQ_WIDE_FLUSH_AUX
#define QQQ_WIDE_FLUSH_AUX 0  // cache policy of the tile walk's D stores (buffer aux bits: 1 sc0, 2 nt, 16 sc1); measurement builds set it
#endif
// QQQ_WIDE_STAGGER = d > 0 (round 6): the four waves of a workgroup run d issue slots apart instead of in lock step.  Wave w takes the stage barrier d (3 - w)
// slots INTO the next stage (wave 3 in front of its first MFMA, wave 0 behind its 3 d-th): all four meet at the same moment, so wave w runs d (3 - w) slots
// ahead of wave 3 from then on -- and no two waves hand the LDS / the vector-memory unit the same instruction in the same cycle.  0: one common barrier.
#ifndef QQQ_WIDE_STAGGER
#define QQQ_WIDE_STAGGER 0
#endif
#ifndef QQQ_WIDE_SLOTMAP
#define QQQ_WIDE_SLOTMAP 5  // bit 0: per-channel, bit 1: per-group, bit 2: the 128-token shape too
#endif
// The static roles of the NSLOT = 2 * HW * MT issue slots of a 64-k step (slot k: m-tile k / (2 HW), column set k % (2 HW)).
// HW = 2 (64 columns per wave): fragment re-read behind every fourth MFMA, ring refill in slots 2 and 6, the LDS-DMA's M0 in
// slot 8 and the DMA itself in slot 9 of every 16, group scales in slot 1 of a stage's second step; unpack items in the rest.
// HW = 1 (32 columns per wave, 256 x 128 tiles): every ODD slot re-reads a fragment, so everything else sits in even ones.
__host__ __device__ constexpr bool wide_frag_slot(int hw, int k) { return k % (2 * hw) == 2 * hw - 1; }
__host__ __device__ constexpr bool wide_refill_slot(int hw, int k, int hf) { return hf < hw && k == 2 + 4 * hf; }
// QQQ_WIDE_DWORD (round 6): the packed weights of a step come in as FOUR 4-byte loads per lane and 32-column half instead of one 16-byte load + a 4 x 4 quad transpose:
// MFMA lane (h, cq, jt) needs the words jt of the pieces kq = 0 .. 3 of its chunk, i.e. the dwords at chunk + 16 kq + 4 jt -- load kq fetches exactly that (a quad of lanes
// takes 16 contiguous bytes; the four loads of a half touch the same lines, the first brings them into the L1).  13 issue slots' worth of transpose (4 VCC writes, 8 DPP
// selects, in a loop that is issue-bound wherever it is not power-bound: per-group everywhere, per-channel in the 128-token / 128-column shapes) against 3 more loads.
// The loads of a step sit one per slot (wide_dw_load_index); ring of 4 steps only (8 steps x 8 loads would pass vmcnt's 63).
// Measured (profiles/r06b_wide_dword_loads.txt, interleaved A/B): 4096 tokens 427 - 435 -> 422 - 427 us, 1024 tokens (128 x 256 tiles) 123.5 - 124.5 -> 119.1, 320 ... 768 -1 ... -2 %,
// per-group 4096 -2 %, 512 -2.5 %; Llama-2-7B layers level ... -2.5 %; no point slower.  Loop: 1.38 -> 1.09 extras per MFMA per-channel, 3.2 -> 2.9 per-group, 2.5 -> 1.8 in the
// 128-token shape.  On.
#ifndef QQQ_WIDE_DWORD
#define QQQ_WIDE_DWORD 1
#endif
__host__ __device__ constexpr bool wide_dw(int mode) { return QQQ_WIDE_DWORD != 0 && mode != 2; }
__host__ __device__ constexpr int wide_dw_load_index(int mt, int hw, int k) {  // DWORD: the load 4 hf + kq that slot k issues; -1: none
  if (hw == 2) {
    if (mt == 8) {  // (128-token tiles, 32 slots: slot 30 is the stage's wait + barrier, and the ring cursor needs a plain slot behind the last load)
      constexpr int at[8] = {2, 5, 6, 10, 14, 18, 22, 26};
      for (int i = 0; i < 8; ++i)
        if (at[i] == k) return i;
      return -1;
    }
    return (k % 4 == 2 && k <= 30) ? (k - 2) / 4 : -1;
  }
  return k == 2 ? 0 : k == 8 ? 1 : k == 10 ? 2 : k == 16 ? 3 : -1;
}
// the ring load(s) of slot k in any mode: >= 0 an index (packed: the half hf; DWORD: 4 hf + kq; expanded: the column set), -1 none
__host__ __device__ constexpr int wide_ring_load(int mode, int mt, int hw, int k) {
  if (mode == 2) return k % ((2 * hw * mt) / (2 * hw)) == 2 ? k / ((2 * hw * mt) / (2 * hw)) : -1;
  if (wide_dw(mode)) return wide_dw_load_index(mt, hw, k);
  return wide_refill_slot(hw, k, 0) ? 0 : wide_refill_slot(hw, k, 1) ? 1 : -1;
}
__host__ __device__ constexpr int wide_ring_last_slot(int mode, int mt, int hw, int hf) {  // the slot that issues the LAST load of half hf of a step (packed modes)
  if (!wide_dw(mode)) return 2 + 4 * hf;
  int last = 0;
  for (int k = 0; k < 2 * hw * mt; ++k)
    if (wide_dw_load_index(mt, hw, k) == 4 * hf + 3) last = k;
  return last;
}
__host__ __device__ constexpr int wide_dma_period(int mt, int hw) { return (2 * hw * mt) / (mt / 4); }  // slots per DMA chunk: 16 / 8
__host__ __device__ constexpr bool wide_m0_slot(int mt, int hw, int k) { return k % wide_dma_period(mt, hw) == (hw == 2 ? 8 : 4); }
__host__ __device__ constexpr bool wide_dma_slot(int mt, int hw, int k) { return k % wide_dma_period(mt, hw) == (hw == 2 ? 9 : 6); }
__host__ __device__ constexpr int wide_scale_slot(int hw) { return hw == 2 ? 1 : 0; }
__host__ __device__ constexpr bool wide_item_slot(int hw, int k) {  // slots of a step that take an unpack item
  if (hw == 1) return k % 2 == 0 && k != 2 && k % 8 != 6;
  return (k % 4 == 0) || (k % 4 == 2 && k != 2 && k != 6) || (k % 8 == 5);
}
__host__ __device__ constexpr int wide_item_slots_before(int hw, int k) {  // number of item slots in [0, k)
  int n = 0;
  for (int j = 0; j < k; ++j) n += wide_item_slot(hw, j) ? 1 : 0;
  return n;
}

// QQQ_WIDE_BALANCE (round 6): the issue model of a wave that is alone on its SIMD, measured (tools/mfma_issue_bench.hip, profiles/r06_mfma_issue_model.txt): every
// instruction -- VALU, SALU, s_nop, a wait that does not block -- takes the wave ~5.2 cycles of issue; behind a 16-cycle MFMA TWO of them are free, the third costs its
// full 4 - 5 cycles, and an empty slot gives nothing back (3 VALU behind every other MFMA: 18.5 cycles per MFMA, 1.5 behind every one: 16.5).  A ds_read_b128 takes ~12
// cycles (alone in a slot: free; with two VALU next to it: + 13), an LDS-DMA ~17 (+ 6 whatever shares its slot).  The quad transpose's pieces were 3-instruction blocks
// (s_mov_b64 vcc + two DPP selects; with hipcc's hazard s_nop and the ring wait in front: 4 - 5 instructions in ONE slot, eight times per 64-k step).  So: every unpack
// item is ONE instruction (the transpose's thirteen per half: the ring wait, then per piece the VCC write and its two selects, in order -- nothing else in the loop writes
// VCC, tests/test_code_object_cpu.py checks the compiled code), and the items are dealt to the slots by CAPACITY: 0 where the slot already carries a fragment re-read, an
// LDS-DMA or a ring refill, 1 next to the DMA's M0 write, 2 elsewhere (scaled up together where a shape has more items than that: per-group, 128-token and 128-column tiles).
#ifndef QQQ_WIDE_BALANCE
#define QQQ_WIDE_BALANCE 1
#endif
// QQQ_WIDE_CURSORS (round 6): the plain kernel's loop loads read through running scalar offsets (one s_add per load kind and stage, as the tile walk's cursors) instead
// of offsets worked out per step from the stage index with a clamp (s_add, s_min, s_add, s_mul in ONE issue slot, three times per stage).  The clamp ("past the end of
// the K slice: re-read its last stage") goes: the descriptors END where their tensors end, so a load past the end of K is either the next slice's / next row's data
// (in range, never used) or out of range for the buffer unit (returns zeros, never used).
#ifndef QQQ_WIDE_CURSORS
#define QQQ_WIDE_CURSORS 1
#endif
// Fixed duties that sit in the slot schedule next to the loads (BALANCE): the stage-end wait + barrier (step parity 1) and, with running cursors (QQQ_WIDE_CURSORS),
// one scalar add per load kind -- each in a slot of its own choosing, its capacity reduced accordingly.
__host__ __device__ constexpr int wide_w8_refill_index_(int mt, int hw, int k) { return k % ((2 * hw * mt) / (2 * hw)) == 2 ? k / ((2 * hw * mt) / (2 * hw)) : -1; }
__host__ __device__ constexpr bool wide_plain_slot(int mode, int mt, int hw, int k) {  // no load, no fragment re-read, no M0 write in this slot
  if (k < 0 || k >= 2 * hw * mt) return false;
  if (wide_frag_slot(hw, k) || wide_dma_slot(mt, hw, k) || wide_m0_slot(mt, hw, k)) return false;
  if (wide_ring_load(mode, mt, hw, k) >= 0) return false;
  if (mode == 1 && k == wide_scale_slot(hw)) return false;
  return true;
}
__host__ __device__ constexpr int wide_barrier_slot(int mode, int mt, int hw) {  // step parity 1: the last plain slot of the step
  for (int k = 2 * hw * mt - 1; k >= 0; --k)
    if (wide_plain_slot(mode, mt, hw, k)) return k;
  return 0;
}
__host__ __device__ constexpr int wide_plain_after(int mode, int mt, int hw, int k0, int skip) {  // the (skip + 1)-th plain slot behind k0 (not the barrier's), or the last one there is; -1: none
  int last = -1;
  for (int k = k0 + 1; k < 2 * hw * mt; ++k)
    if (wide_plain_slot(mode, mt, hw, k) && k != wide_barrier_slot(mode, mt, hw)) {
      last = k;
      if (skip-- == 0) return k;
    }
  return last;
}
// cursor increments: the ring's after the step's last refill (both parities); the LDS-DMA's in front of the stage's first chunk (parity 0: pre-increment, first plain
// slot of the step); the scales' (per-group) behind their load (parity 1)
__host__ __device__ constexpr int wide_ring_inc_slot(int mode, int mt, int hw) {
  int last = 0;
  for (int k = 0; k < 2 * hw * mt; ++k)
    if (wide_ring_load(mode, mt, hw, k) >= 0) last = k;
  return wide_plain_after(mode, mt, hw, last, 1);
}
__host__ __device__ constexpr int wide_dma_inc_slot(int, int, int) { return 0; }  // (slot 0 carries no load, no fragment re-read and no M0 write in any shape)
__host__ __device__ constexpr int wide_scale_inc_slot(int mode, int mt, int hw) { return wide_plain_after(mode, mt, hw, wide_scale_slot(hw), 2); }
// QQQ_WIDE_XWAIT (round 6): the fragment re-reads are inline asm as well, their lgkmcnt waits hand-placed.  hipcc put a wait behind 8 of the 16 re-reads of a step --
// INTO the slot that already carries the ds_read_b128 (~12 cycles of issue): MFMA + ds_read + wait = 22 cycles, 6 lost eight times per step.  Here: one wait per group of
// four m-tiles, in the last plain slot in front of the group's first MFMA of the NEXT step (group 0: in the tail of the step that issued the reads).  LDS reads return in
// issue order (nothing else in the loop counts in lgkmcnt), so "group j has landed" == "at most <reads issued since its last one> are outstanding"; tools/check_waits.py
// replays the compiled loop (every instruction against the reads still in flight).
// Built, replayed clean on every instantiation, GPU suite green -- and measured LEVEL (profiles/r06_wide_balanced_slots.txt: 4096 tokens 423.2 - 427.3 us without, 422.9 -
// 425.9 with): once the slots are balanced the launch sits on the part's power limit again, and issue cycles saved come back as clock lost.  Off; the compiler's waits stay.
#ifndef QQQ_WIDE_XWAIT
#define QQQ_WIDE_XWAIT 0
#endif
__host__ __device__ constexpr int wide_frag_slot_of(int hw, int m) { return 2 * hw * m + 2 * hw - 1; }  // the slot whose extras re-read x[m]
// the wait slot of group j (m-tiles 4 j .. 4 j + 3): as an ABSOLUTE slot on the two-step timeline (step 0 issues the reads, step 1 uses them); -1: no such group
__host__ __device__ constexpr int wide_xw_abs(int mode, int mt, int hw, int j) {
  const int nslot = 2 * hw * mt;
  if (4 * j >= mt) return -1;
  const int use = nslot + 2 * hw * 4 * j;  // first MFMA of the group in step 1
  for (int a = use - 2; a > wide_frag_slot_of(hw, 4 * j + 3); --a) {
    const int k = a % nslot;
    if (wide_plain_slot(mode, mt, hw, k) && k != wide_barrier_slot(mode, mt, hw)) return a;
  }
  return -1;
}
__host__ __device__ constexpr int wide_xw_count(int mode, int mt, int hw, int j) {  // re-reads issued behind the group's last one and in front of its wait
  const int nslot = 2 * hw * mt, a = wide_xw_abs(mode, mt, hw, j), last = wide_frag_slot_of(hw, 4 * j + 3);
  int n = 0;
  for (int step = 0; step < 2; ++step)
    for (int m = 0; m < mt; ++m) {
      const int at = step * nslot + wide_frag_slot_of(hw, m);
      if (at > last && at < a) ++n;
    }
  return n;
}
__host__ __device__ constexpr int wide_xw_count_at(int mode, int mt, int hw, int k) {  // the wait of slot k of a step: lgkmcnt(n) (the strictest of the groups that wait here); -1: none
  int n = -1;
  for (int j = 0; 4 * j < mt; ++j)
    if (wide_xw_abs(mode, mt, hw, j) >= 0 && wide_xw_abs(mode, mt, hw, j) % (2 * hw * mt) == k && (n < 0 || wide_xw_count(mode, mt, hw, j) < n)) n = wide_xw_count(mode, mt, hw, j);
  return n;
}
// QQQ_WIDE_CHAINPREP (round 6, tile walk): the offsets of a stage's loads are prepared one stage AHEAD, in plain slots of the previous stage's second step (two register
// sets, picked by the stage's compile-time parity), instead of at the stage's start -- where copies, cursor adds and the tile-end test stood in ONE slot: 13 instructions
// between two MFMAs, once per stage (tools/slot_load.py).  What is left at the boundary: the countdown, the seam test, the "last P stages" test.
// Measured (profiles/r06c_tile_walk_prepared_offsets.txt): level.  The walk kernels are out of SGPRs (106, two spilled), hipcc keeps the loop-carried set in VGPRs and the
// v_readfirstlane read-backs -- which must stand at the stage head, five wait states ahead of the first load -- cost what the boundary bookkeeping did.  Parity-clean, off.
#ifndef QQQ_WIDE_CHAINPREP
#define QQQ_WIDE_CHAINPREP 0
#endif
__host__ __device__ constexpr int wide_prep_slot(int mode, int mt, int hw, int j) { return wide_plain_after(mode, mt, hw, -1, j); }  // the j-th plain slot of a step (j = 0 .. 3)
__host__ __device__ constexpr int wide_bal_cap(int mode, int mt, int hw, int t, int k, int cursors) {  // cursors: 0 none, 1 the plain kernel's running cursors, 2 the tile walk's prepared offsets
  if (!wide_plain_slot(mode, mt, hw, k)) return wide_m0_slot(mt, hw, k) && !wide_frag_slot(hw, k) && !wide_dma_slot(mt, hw, k) ? 1 : 0;
  int c = 2;
  if (t == 1 && k == wide_barrier_slot(mode, mt, hw)) c -= 2;
  if (QQQ_WIDE_XWAIT != 0 && wide_xw_count_at(mode, mt, hw, k) >= 0) c -= 1;
  if (cursors == 1) {
    if (k == wide_ring_inc_slot(mode, mt, hw)) c -= 1;
    if (t == 0 && k == wide_dma_inc_slot(mode, mt, hw)) c -= 1;
    if (mode == 1 && t == 1 && k == wide_scale_inc_slot(mode, mt, hw)) c -= 1;
  }
  if (cursors == 2 && t == 1) {
    if (k == wide_prep_slot(mode, mt, hw, 0) || k == wide_prep_slot(mode, mt, hw, 1) || (mode == 1 && k == wide_prep_slot(mode, mt, hw, 3))) c -= 2;
    if (k == wide_prep_slot(mode, mt, hw, 2)) c -= 1;
  }
  return c < 0 ? 0 : c;
}
__host__ __device__ constexpr int wide_bal_head(int mode) { return wide_dw(mode) ? 1 : 13; }  // per 32-column half: the ring wait, then (unless DWORD) 4 x (VCC write, 2 selects)
__host__ __device__ constexpr int wide_bal_items(int mode) { return wide_bal_head(mode) + (mode == 1 ? 32 : 12); }  // ... then the unpack parts
__host__ __device__ constexpr int wide_bal_cost(int mode, int w) { return w < wide_bal_head(mode) ? 1 : (mode == 1 ? 2 : 1); }  // instructions of item w of a half
// number of items (of the hw * wide_bal_items(mode) of a step, in order) dealt to slots [0, k): item i goes to the first slot whose cumulative share of the capacity
// reaches the item's cumulative share of the instructions
__host__ __device__ constexpr int wide_bal_before(int mode, int mt, int hw, int t, int k, int cursors) {
  const int nslot = 2 * hw * mt, per = wide_bal_items(mode), ni = hw * per;
  int ct = 0, cb = 0, tc = 0;
  for (int j = 0; j < nslot; ++j) ct += wide_bal_cap(mode, mt, hw, t, j, cursors);
  for (int j = 0; j < k; ++j) cb += wide_bal_cap(mode, mt, hw, t, j, cursors);
  for (int i = 0; i < ni; ++i) tc += wide_bal_cost(mode, i % per);
  if (k >= nslot) return ni;
  int n = 0, e = 0;
  for (int i = 0; i < ni; ++i) {
    e += wide_bal_cost(mode, i % per);
    if ((long long)e * ct <= (long long)cb * tc) ++n;  // item ends inside the capacity of the slots before k
  }
  return n;
}

// (the dealing worked out ONCE per shape and step parity: clang's constant evaluator walks the loops above per call, and the slot bodies ask 2 x 64 x 8 times per instantiation)
template <int MODE, int MT, int HW, int T, int CURSORS>
struct WideBalTable {
  struct Tab {
    int before[2 * HW * MT + 1];
  };
  static constexpr Tab make() {
    Tab r{};
    constexpr int nslot = 2 * HW * MT, per = wide_bal_items(MODE), ni = HW * per;
    int cap[nslot] = {};
    int ct = 0, tc = 0;
    for (int j = 0; j < nslot; ++j) {
      cap[j] = wide_bal_cap(MODE, MT, HW, T, j, CURSORS);
      ct += cap[j];
    }
    for (int i = 0; i < ni; ++i) tc += wide_bal_cost(MODE, i % per);
    int cb = 0, n = 0, e = wide_bal_cost(MODE, 0);  // e: cumulative cost up to and including item n
    for (int k = 0; k < nslot; ++k) {
      while (n < ni && (long long)e * ct <= (long long)cb * tc) {
        ++n;
        if (n < ni) e += wide_bal_cost(MODE, n % per);
      }
      r.before[k] = n;
      cb += cap[k];
    }
    r.before[nslot] = ni;
    return r;
  }
  static constexpr Tab tab = make();
};

static_assert(WideBalTable<0, 16, 2, 0, 1>::tab.before[17] == wide_bal_before(0, 16, 2, 0, 17, 1) && WideBalTable<0, 16, 2, 1, 1>::tab.before[63] == wide_bal_before(0, 16, 2, 1, 63, 1) &&
                  WideBalTable<1, 8, 2, 1, 0>::tab.before[9] == wide_bal_before(1, 8, 2, 1, 9, 0) && WideBalTable<1, 16, 1, 0, 2>::tab.before[30] == wide_bal_before(1, 16, 1, 0, 30, 2),
              "the table is the function");

// LDS-DMA staging: the activations go global -> LDS directly (buffer_load_dwordx4 ... lds, one
// 16-byte chunk per lane and instruction, lane-linear destination M0 + 16 * lane; the row swizzle is applied on the SOURCE
// side).  The second ablation (profiles/r03_wide_ablation2.txt) put 10 % of the loop on the staging's reload + ds_write pair
// -- their issue cost, not their latency.  An LDS-DMA is invisible to hipcc's wait-count bookkeeping, and its counted waits
// for the VISIBLE loads would then over-wait (vmcnt counts every load in flight), so EVERY vector-memory load of the loop is
// inline asm and the waits are placed by hand from the static schedule below: loads complete in issue order, so "the load I
// need is done" == "at most <number of loads issued after it> are outstanding".
// Vector-memory loads slot k of a step (parity t = second step of its stage) issues, in the order the slot issues them.
// mode: 0 per-channel int4, 1 per-group int4 (re-quantised in the loop), 2 expanded int8 weights (W8: one 16-byte load per column set and step)
__host__ __device__ constexpr int wide_w8_refill_index(int mt, int hw, int k) {  // W8: the column set whose operand slot k refills, -1: none
  const int per = (2 * hw * mt) / (2 * hw);
  return k % per == 2 ? k / per : -1;
}
__host__ __device__ constexpr int wide_loads_in_slot(int mode, int mt, int hw, int t, int k) {
  int n = 0;
  if (mode == 1 && t == 1 && k == wide_scale_slot(hw) && !(QQQ_WIDE_ABLATE & 8)) n += 2;            // the group scales of stage i + P
  if (mode != 2 && wide_ring_load(mode, mt, hw, k) >= 0 && !(QQQ_WIDE_ABLATE & 8)) n += 1;  // weight-ring refill
  if (mode == 2 && wide_w8_refill_index(mt, hw, k) >= 0 && !(QQQ_WIDE_ABLATE & 8)) n += 1;
  if (wide_dma_slot(mt, hw, k) && !(QQQ_WIDE_ABLATE & 2)) n += 1;                                  // one activation chunk per lane
  return n;
}
__host__ __device__ constexpr int wide_loads_in_step(int mode, int mt, int hw, int t, int from, int to) {  // slots [from, to)
  int n = 0;
  for (int k = from; k < to && k < 2 * hw * mt; ++k) n += wide_loads_in_slot(mode, mt, hw, t, k);
  return n;
}
// Loads issued after slot k0 of a step with parity t0 and before slot k1 of the step `steps` later (>= 1).
__host__ __device__ constexpr int wide_loads_between(int mode, int mt, int hw, int t0, int k0, int steps, int k1) {
  int n = wide_loads_in_step(mode, mt, hw, t0, k0 + 1, 2 * hw * mt);
  for (int j = 1; j < steps; ++j) n += wide_loads_in_step(mode, mt, hw, (t0 + j) & 1, 0, 2 * hw * mt);
  return n + wide_loads_in_step(mode, mt, hw, (t0 + steps) & 1, 0, k1);
}
__host__ __device__ constexpr int wide_w8_last_refill_slot(int mt, int hw) {
  int last = 0;
  for (int k = 0; k < 2 * hw * mt; ++k) last = wide_w8_refill_index(mt, hw, k) >= 0 ? k : last;
  return last;
}
__host__ __device__ constexpr int wide_last_dma_slot(int mt, int hw) {
  int last = 0;
  for (int k = 0; k < 2 * hw * mt; ++k) last = wide_dma_slot(mt, hw, k) ? k : last;
  return last;
}

// CHAIN (round 4): the persistent tile walk.  grid = one workgroup per CU; a workgroup walks ITS run of tiles (the positions
// the one-tile-per-workgroup grid would have given this CU round after round, same XCD-aware order) without ever draining its
// pipeline: the loads that the plain kernel redirects past the end of K (LDS-DMA LA stages ahead, weight ring RS steps ahead,
// group scales P stages ahead) fetch the NEXT tile's first stages instead, so at the tile seam the next tile's operands are
// already in LDS / in the ring, its first step is already unpacked, and the only thing between the last MFMA of a tile and the
// first MFMA of the next is the flush of the accumulators: an LDS-free epilogue (register-index <-> lane-row transposition with
// the gfx950 row swaps, straight from the registers to D) -- the stage buffers stay untouched.  The flat stage sequence keeps
// rotating through the P LDS buffers / RS ring slots across seams (a tile need not be a multiple of P stages), so the seam
// exists once per stage position of the unrolled trip.  The counterpart of the reference's stripe walk
// (csrc/qqq_gemm.cu:261-338, :729-760, :792-812), without partial tiles: ksplit == 1 only.
//
// MODE 2 (round 6, SURVEY 8 f-3's opt-in load-time re-layout): `B` is the EXPANDED weight tensor W8 of qqq_expand_int8 -- the per-group weights re-quantised to
// int8 ONCE at load time (the bit-exact dequant_group4), stored in the MFMA operand order: [64-k step][64-column group][column set q = 2 hf + b][lane][16 bytes],
// lane (h, c, jt) holding k = 64 step + 16 h + 0..15 of column 64 ng + 16 jt + 8 b + 4 hf + c.  One 16-byte load per column set and step IS the operand: no
// transpose, no unpack, no re-quantiser, no group scales in the loop -- the MFMAs read the ring registers directly, and a ring slot is refilled (with step
// s - 1 + RS) during the step AFTER the one that consumed it: RS - 1 steps of lead.  s3 is not read.
template <int MODE, int MT, int P, int RS, int HW, bool CHAIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void qqq_wide_kernel(
    const int8_t* __restrict__ A, const unsigned char* __restrict__ B, int32_t* __restrict__ C, _Float16* __restrict__ D,
    const float* __restrict__ s1, const float* __restrict__ s2, const _Float16* __restrict__ s3,
    int32_t* __restrict__ acc_out, int* __restrict__ tickets, const _Float16* __restrict__ bias, const int M, const int N,
    const int K, const int tiles_m, const int tiles_n, const int PW, const int ksplit_hf) {
  // (hand-off switches ride in the upper half of the K-split argument -- tune.fused bits 2 / 3: 1 = the formal agent-scope ACQUIRE
  // fence in front of the fold, 2 = agent-scope RELEASE on the depositor's completion count; see qqq_common.hip.h)
  // (bits 24..29: `skew`, the 128-k stages the LAST K slice gets on top of an even share, as in the panel kernel)
  const int ksplit = ksplit_hf & 0xffff, hflags = (ksplit_hf >> 16) & 0xff, skew = (ksplit_hf >> 24) & 0x3f;
  constexpr bool GROUPED = MODE == 1, W8 = MODE == 2;
  static_assert(MODE >= 0 && MODE <= 2, "0 per-channel, 1 per-group, 2 expanded int8");
  static_assert(!W8 || RS >= 3, "W8: a slot is refilled one step after its use -- RS - 1 steps of lead, the wait count needs at least one whole step in between");
  constexpr int RL = W8 ? RS - 1 : RS;   // the ring refill of step s fetches step s + RL
  static_assert(MT == 16 || MT == 8, "m-tiles of 16 tokens per wave (= per workgroup): 256 or 128 tokens");
  static_assert(HW == 2 || HW == 1, "a wave owns both 32-column halves of a 64-column group, or (128-column tiles) one");
  constexpr int ROWS = 16 * MT;
  constexpr int NQ = 2 * HW;             // column sets (of 16) per wave
  constexpr int BN = 128 * HW;           // 4 waves x 32 HW columns
  constexpr int NT = 256;
  constexpr int XB = ROWS * 128;         // bytes of one activation stage (128 k)
  constexpr int XPT = XB / 16 / NT;      // 16-byte chunks per thread and stage (8 / 4)
  static_assert((2 * P) % RS == 0, "weight ring (in 64-k steps) must divide the unroll period");
  // P LDS stage buffers = unroll period in stages.  The LDS-DMA of stage i + LA (LA = P - 1) is ISSUED during stage i (its
  // buffer was last read in stage i - 1), is waited for at the end of stage i + LA - 2 -- P - 3 full stages of lead: one with
  // four buffers, three with six -- and published by that stage's barrier: the fragment reads of stage i + LA begin in the last
  // step of stage i + LA - 1.  (A stage of 128 tokens lasts 0.7 us: less than an HBM round trip under load.)
  static_assert(P >= 4 && P <= 6, "LDS stage buffers: one being read, one complete, P - 2 in flight");
  constexpr int LA = P - 1;
  constexpr int NSLOT = NQ * MT;         // issue slots (= MFMAs) of a 64-k step
  constexpr int EPR = HW == 2 ? 8 * MT : 16 * MT;  // rows per epilogue pass: half the tile, or (128-column tiles) all of it (int32 image: EPR x (BN + 4) x 4 B = 130 / 65 / 132 / 66 KiB)
  constexpr int EP_STRIDE = BN + 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // 64-column group of the strip
  QQQ_WTR(0);  // (measurement builds only, -DQQQ_PANEL_TRACE)

  // ---- XCD-aware tile order (speed only): block b runs on XCD b % 8; an XCD walks panels of PW strips x all m-tiles ----
  int tile_m, tile_n, tile_lin, sp;  // tile coordinates, tile index, K slice
  auto locate = [&](const int lin, int& tm, int& tn) {
    const int full = (tiles_n / PW) * PW * tiles_m;
    if (lin < full) {
      const int panel = lin / (PW * tiles_m), within = lin % (PW * tiles_m);
      tm = within / PW;
      tn = panel * PW + within % PW;
    } else {
      const int rem = lin - full, pw = tiles_n % PW;
      tm = rem / pw;
      tn = (tiles_n / PW) * PW + rem % pw;
    }
  };
  int ch_first = 0, ch_stride = 0, ch_tiles = 1;  // CHAIN: position of this workgroup's first tile in the XCD-ordered run, step, count
  {
    // (with a K split the slices of a tile take consecutive positions of the XCD's run: resident together, on one XCD)
    const int ntot = tiles_m * tiles_n * (CHAIN ? 1 : ksplit);
    const int bid = blockIdx.x;
    const int q = ntot >> 3, rr = ntot & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int lin2 = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    if constexpr (CHAIN) {
      // the XCD's run has cnt positions; this workgroup takes idx, idx + stride, ... (what the CU would have been handed round
      // after round by the one-tile-per-workgroup grid).  The host launches a multiple of 8 workgroups, at most as many as tiles.
      const int cnt = q + (xcd < rr ? 1 : 0);
      ch_stride = (int)(gridDim.x >> 3);
      ch_first = lin2;
      ch_tiles = idx < cnt ? (cnt - idx + ch_stride - 1) / ch_stride : 0;
      if (ch_tiles == 0) return;
      sp = 0;
      tile_lin = lin2;
      locate(lin2, tile_m, tile_n);
    } else if (ksplit == 2 && (hflags & 16) != 0 && ((tiles_m * tiles_n) & 3) == 0) {
      // (measurement, tune.fused bit 7: the two slices of a tile on NEIGHBOURING XCDs -- XCD pair p walks a quarter of the tiles, the even XCD their first K
      // halves, the odd one the second: an L2 then streams half of K of its activation panels and weight strips instead of all of it for half as many
      // tiles; the deposits cross the fabric)
      const int per = (tiles_m * tiles_n) >> 2;
      sp = xcd & 1;
      tile_lin = (xcd >> 1) * per + idx;
      locate(tile_lin, tile_m, tile_n);
    } else {
      const int lin = lin2 / ksplit;
      sp = lin2 - lin * ksplit;
      tile_lin = lin;
      locate(lin, tile_m, tile_n);
    }
  }
  // ---- K split: which XCD am I on?  Published into the tile's arrival word right away (a nibble per slice: valid bit + XCC id),
  // read back with the arrival ticket after the main loop.  Slices that find each other on ONE XCD hand their partial tiles over
  // through that XCD's L2 (plain write-back stores; the finisher's agent-scope loads miss L1 and hit the L2 line) instead of
  // writing them through to memory -- see the deposit below for why the decision is consistent on both ends.
  unsigned my_xcc = 0;
  if (!CHAIN && ksplit > 1) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 7u;
    if (tid == 0 && ksplit <= 6)
      __hip_atomic_fetch_or(tickets + 2 * (size_t)tile_lin, (int)((8u | my_xcc) << (8 + 4 * sp)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int mbase = tile_m * ROWS;
  const int ngroups = N >> 6;
  const int cb = tile_n * 4 + wn;        // this wave's block of 32 HW columns: a 64-column group, or (HW = 1) half cb & 1 of group cb >> 1
  int ng = HW == 2 ? cb : cb >> 1;
  const int whalf = HW == 2 ? 0 : (cb & 1);
  if (ng >= ngroups) ng = ngroups - 1;   // N % BN != 0: surplus waves of the last strip compute on clamped columns, store nothing
  const unsigned rowbytes = (unsigned)N * 8u;
  const unsigned wstep = W8 ? (unsigned)N * 64u : 4u * rowbytes;  // bytes of B per 64-k step: four k-tiles of packed int4, or (W8) N x 64 int8

  // K slice [st0, st0 + NST) in 128-k stages (K % 128 == 0); steps and stages below are relative to it
  // (uneven slices: the last one is `skew` stages longer, so that it arrives last and finds the other deposits complete)
  const int NSE = (K >> 7) - skew;
  const int st0 = (int)(((long long)NSE * sp) / ksplit);
  const int NST = (sp == ksplit - 1 ? (K >> 7) : (int)(((long long)NSE * (sp + 1)) / ksplit)) - st0;
  const int KS = 2 * NST;                // 64-k steps

  // ---- per-lane sources ----
  const int h = lane >> 4, cq = (lane >> 2) & 3, q4 = lane & 3;  // q4: kq as a load lane, jt as an MFMA lane
  // Every vector-memory load from here to the end of the main loop is inline asm (see "LDS-DMA staging" above): descriptors as plain
  // SGPR quads (base, stride 0, no bound, raw-buffer flags), 32-bit lane offset, scalar step / stage offset.
  auto descriptor = [](const void* base_uniform) {
    const unsigned long long a = (unsigned long long)base_uniform;  // (readfirstlane: an "s" operand must not reach the asm in VGPRs)
    return (v4u){(unsigned)__builtin_amdgcn_readfirstlane((int)a), (unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)), 0xffffffffu,
                 0x00020000u};
  };
  // (CHAIN: one descriptor over the whole tensor, the tile's column group goes into the scalar offset -- ng * 512 / ng * 128 bytes)
  v4u wdesc = descriptor(CHAIN ? (const void*)B : (const void*)(B + (size_t)ng * (W8 ? 4096 : 512)));
  constexpr bool CUR = !CHAIN && QQQ_WIDE_CURSORS != 0;
  if constexpr (CUR) {  // the descriptor ends with the tensor (packed: K / 16 rows of 8 N bytes; expanded: K x N bytes -- below 4 GiB, the host sends larger ones elsewhere)
    const unsigned long long total = W8 ? (unsigned long long)K * (unsigned)N : ((unsigned long long)K * (unsigned)N) >> 1;
    wdesc[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(total - (unsigned long long)ng * (W8 ? 4096u : 512u)));
  }
  const unsigned woff = W8 ? (unsigned)(lane * 16 + 2048 * whalf)                                // + step * wstep (scalar) + 1024 * q
                           : (unsigned)h * rowbytes + (unsigned)(cq * 64 + q4 * 16 + 256 * whalf);   // + step * wstep (scalar) + 256 * hf
  v4u sdesc = descriptor(GROUPED ? (CHAIN ? (const void*)s3 : (const void*)(s3 + (size_t)ng * 64)) : (const void*)B);
  if constexpr (CUR && GROUPED) sdesc[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(K >> 7) * (unsigned)N * 2u - (unsigned)ng * 128u));
  const unsigned soff_l = (unsigned)((cq * 8 + 2 * q4) * 2 + 64 * whalf);         // + stage * N * 2 (scalar) + 64 * hf
  // Activation staging by LDS-DMA: instruction q of wave wn fills the 1 KiB [rows 8 wn + 32 q .. + 8) x 128 bytes of the stage
  // image, lane l -> byte 16 l of it = (row l >> 3, slot l & 7).  The image keeps the XOR swizzle of the fragment reads
  // (16-byte piece p of a row sits in slot p ^ ((row >> 1) & 7)), so the lane FETCHES piece slot ^ ((row >> 1) & 7).
  // (CHAIN: the descriptor follows the LDS-DMA cursor from tile to tile; its size field ends at the tile's last valid row, so
  // rows past M are out of range for the buffer unit -- whatever lands in their LDS rows is computed on and never stored)
  v4u xdesc = descriptor(A + (size_t)mbase * K);
  if constexpr (CUR) {  // ends at the last valid row of A (rows of this tile past M are clamped below; a stage past K in the tensor's last row is out of range)
    const unsigned long long rest = (unsigned long long)(M - mbase) * (unsigned)K;
    xdesc[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rest < 0xffffffffull ? rest : 0xffffffffull));
  }
  const int xr0 = tid >> 3, xslot = tid & 7;
  unsigned xoff[XPT];
#pragma unroll
  for (int q = 0; q < XPT; ++q) {
    int row = xr0 + 32 * q;
    if (!CHAIN && mbase + row >= M) row = M - 1 - mbase;  // rows past M: re-read the last row (computed, never stored)
    xoff[q] = (unsigned)row * (unsigned)K + (unsigned)((xslot ^ ((xr0 >> 1) & 7)) * 16);  // (rows 32 apart: same swizzle)
  }
  const unsigned lds_wave = (unsigned)wn * 1024u;  // + buffer * XB + q * 4096: the LDS-DMA's M0

  // ---- CHAIN: what a tile contributes to the three load cursors (all wave-uniform) ----
  // A-descriptor words of its m-tile, scalar offsets of this wave's column group into B (ng * 512) and s3 (ng * 128 bytes)
  auto tile_ref = [&](const int tm, const int tn, unsigned& a_lo, unsigned& a_hi, unsigned& a_rec, unsigned& w_so, unsigned& s_so) {
    const unsigned long long a = (unsigned long long)(A + (size_t)tm * ROWS * (size_t)K);
    a_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    a_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32));
    const int rows = M - tm * ROWS < ROWS ? M - tm * ROWS : ROWS;
    a_rec = (unsigned)__builtin_amdgcn_readfirstlane(rows * K);
    const int cbw = tn * 4 + wn;
    int g = HW == 2 ? cbw : cbw >> 1;
    if (g >= ngroups) g = ngroups - 1;
    // (readfirstlane: these feed the cursors, which must stay scalar -- a cursor kept in a VGPR reaches its load through a
    // v_readfirstlane one instruction ahead of it, inside the 5 wait states a VALU-written SGPR needs before a vector-memory
    // instruction may read it; hipcc's hazard bookkeeping does not see through the inline-asm MFMA in between.  Seen as stale
    // offsets in the first load behind every such copy: column half 0 and the first LDS-DMA chunk of every stage.)
    w_so = (unsigned)__builtin_amdgcn_readfirstlane(g * (W8 ? 4096 : 512));
    s_so = (unsigned)__builtin_amdgcn_readfirstlane(g * 128);
  };
  // the tile being computed (its epilogue needs the coordinates) and the one after it (which the loads cross into over the last
  // P stages of the current one); a workgroup on its last tile crosses into that same tile again: loaded, never used
  int cur_tm = tile_m, cur_tn = tile_n, nx_tm = tile_m, nx_tn = tile_n, ch_pos = 0;  // ch_pos: index of the current tile in this workgroup's run
  unsigned cu_a_lo = 0, cu_a_hi = 0, cu_a_rec = 0, cu_w_so = 0, cu_s_so = 0;
  unsigned nx_a_lo = 0, nx_a_hi = 0, nx_a_rec = 0, nx_w_so = 0, nx_s_so = 0;
  // Where the loads of a stage read.  With `left` stages of the tile to go (this one included) the LDS-DMA fetches stage
  // NST - left + LA, the ring refills steps 2 (NST - left) + t + RS, the scale load stage NST - left + P.  While all of them
  // stay inside the tile (left > P: all but the last P stages) three running offsets advance by constants -- scalar adds, the
  // only bookkeeping the common stage pays.  In the last P stages the offsets are worked out from `left`, each load going to the
  // current tile or to the next one; the seam then restarts the running offsets at the new tile's stage 0.
  unsigned cx_so = 0, cr_so = 0, cc_so = 0;
  auto sgpr = [](const unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  auto cursors_at_stage0 = [&]() {
    cx_so = (unsigned)LA * 128u;
    cr_so = sgpr(cu_w_so + (unsigned)RL * wstep);
    cc_so = sgpr(cu_s_so + (unsigned)P * (unsigned)N * 2u);
    xdesc[0] = sgpr(cu_a_lo), xdesc[1] = sgpr(cu_a_hi), xdesc[2] = sgpr(cu_a_rec);
  };
  if constexpr (CHAIN) {
    tile_ref(tile_m, tile_n, cu_a_lo, cu_a_hi, cu_a_rec, cu_w_so, cu_s_so);
    if (ch_tiles > 1) locate(ch_first + ch_stride, nx_tm, nx_tn);
    tile_ref(nx_tm, nx_tn, nx_a_lo, nx_a_hi, nx_a_rec, nx_w_so, nx_s_so);
    cursors_at_stage0();
  }

  // stage / step indices past the end are redirected to the last one (loaded, never used): the loop stays branch-free
  // (M0 is reserved: hipcc never allocates it, and nothing else in this kernel uses it.  A write of M0 needs one wait state in
  // front of the LDS-DMA that reads it: in the loop the write sits one issue slot earlier, dma_m0 / dma_go.)
  auto dma_m0 = [&](auto bufc, auto qc) __attribute__((always_inline)) {
    constexpr unsigned dst = (unsigned)(decltype(bufc)::value * XB + decltype(qc)::value * 4096);
    (void)lds_wave;  // (odr-use: clang does not capture what only an asm operand of a generic lambda names)
    asm volatile("s_add_u32 m0, %0, %1" : : "s"(lds_wave), "n"(dst) : "scc");
  };
  // (the scalar offset is pinned to an SGPR: a value hipcc can fold to a literal is not a valid soffset operand)
  auto dma_go = [&](auto qc, unsigned so) __attribute__((always_inline)) {
    (void)xoff[0], (void)xdesc;
    so = __builtin_amdgcn_readfirstlane(so);
    asm("" : "+s"(so));
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(xoff[decltype(qc)::value]), "s"(xdesc), "s"(so) : "memory");
  };
  // (cursors: the running offset IS the operand -- no copy, and its last write is slots away: no s_nop in front of the load)
  auto dma_go_cur = [&](auto qc) __attribute__((always_inline)) {
    (void)xoff[0], (void)xdesc, (void)cx_so;
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(xoff[decltype(qc)::value]), "s"(xdesc), "s"(cx_so) : "memory");
  };
  auto dma_x = [&](auto bufc, auto qc, const unsigned so) __attribute__((always_inline)) {  // chunk q of a stage -> LDS buffer buf
    dma_m0(bufc, qc);
    asm volatile("s_nop 0");
    dma_go(qc, so);
  };
  auto dma_stage_so = [&](auto bufc, const unsigned so) __attribute__((always_inline)) {
    qqq_static_for<XPT>([&](auto qc) { dma_x(bufc, qc, so); });
  };
  auto dma_stage = [&](auto bufc, const int st_rel) __attribute__((always_inline)) {
    const int st = st_rel < NST ? st_rel : NST - 1;
    dma_stage_so(bufc, (unsigned)(st0 + st) * 128u);
  };
  auto asm_load_w = [&](v4u& dst, auto hfc, unsigned so) __attribute__((always_inline)) {
    (void)woff, (void)wdesc;
    so = __builtin_amdgcn_readfirstlane(so);
    asm("" : "+s"(so));
    if constexpr ((QQQ_W_NT & 8) != 0)  // (measurement builds: non-temporal weight refills)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4 nt" : "=v"(dst) : "v"(woff), "s"(wdesc), "s"(so), "n"((W8 ? 1024 : 256) * decltype(hfc)::value));
    else
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(dst) : "v"(woff), "s"(wdesc), "s"(so), "n"((W8 ? 1024 : 256) * decltype(hfc)::value));
  };
  auto asm_load_w_cur = [&](v4u& dst, auto hfc) __attribute__((always_inline)) {
    (void)woff, (void)wdesc, (void)cr_so;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(dst) : "v"(woff), "s"(wdesc), "s"(cr_so), "n"((W8 ? 1024 : 256) * decltype(hfc)::value));
  };
  constexpr int WRN = W8 ? NQ : HW;      // 16-byte loads per step and lane: one per 32-column half of packed int4, or (W8) one per column set
  auto load_w_so = [&](const unsigned so, v4u (&dst)[WRN]) __attribute__((always_inline)) {
    qqq_static_for<WRN>([&](auto hfc) { asm_load_w(dst[decltype(hfc)::value], hfc, so); });
  };
  auto load_w = [&](const int step_rel, v4u (&dst)[WRN]) __attribute__((always_inline)) {
    const int s = step_rel < KS ? step_rel : KS - 1;
    load_w_so((unsigned)(2 * st0 + s) * wstep, dst);
  };
  // (the scales of a stage are always fetched as two words -- HW = 1 needs only the first -- so that the load counts of the
  // static schedule do not depend on HW)
  auto load_sc_so = [&](unsigned so, unsigned (&dst)[2]) __attribute__((always_inline)) {  // (raw words: an h2 copy behind the asm would read early)
    so = __builtin_amdgcn_readfirstlane(so);
    asm("" : "+s"(so));
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst[0]) : "v"(soff_l), "s"(sdesc), "s"(so));
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(dst[1]) : "v"(soff_l), "s"(sdesc), "s"(so), "n"(HW == 2 ? 64 : 0));
  };
  auto load_sc_cur = [&](unsigned (&dst)[2]) __attribute__((always_inline)) {
    (void)cc_so;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst[0]) : "v"(soff_l), "s"(sdesc), "s"(cc_so));
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(dst[1]) : "v"(soff_l), "s"(sdesc), "s"(cc_so), "n"(HW == 2 ? 64 : 0));
  };
  auto load_sc = [&](const int st_rel, unsigned (&dst)[2]) __attribute__((always_inline)) {
    const int st = st_rel < NST ? st_rel : NST - 1;
    load_sc_so((unsigned)(st0 + st) * (unsigned)N * 2u, dst);
  };

  v4i acc[MT][NQ];  // [mt][2 * hf + b]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[mt][q] = (v4i){0, 0, 0, 0};

  v4u wr[RS][WRN];
  constexpr bool DW = wide_dw(MODE);
  static_assert(!DW || QQQ_WIDE_BALANCE != 0, "the dword loads are scheduled by the balanced slot plan");
  unsigned wq[DW ? RS : 1][HW][4];  // DWORD: the ring as words [step][half][kq] -- word jt (this lane's) of piece kq of the lane's chunk
  const unsigned woff_dw = (unsigned)h * rowbytes + (unsigned)(cq * 64 + q4 * 4 + 256 * whalf);  // + step * wstep (scalar) + 256 hf + 16 kq (immediate)
  auto asm_load_d = [&](unsigned& dst, auto ic, unsigned so, auto rawc) __attribute__((always_inline)) {  // load index 4 hf + kq; rawc: `so` is a cursor (no copy)
    constexpr int li = decltype(ic)::value;
    (void)woff_dw, (void)wdesc;
    if constexpr (!decltype(rawc)::value) {
      so = __builtin_amdgcn_readfirstlane(so);
      asm("" : "+s"(so));
    }
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(dst) : "v"(woff_dw), "s"(wdesc), "s"(so), "n"(256 * (li / 4) + 16 * (li % 4)));
  };
  auto load_d_step = [&](const unsigned so, unsigned (&dst)[HW][4]) __attribute__((always_inline)) {  // all of a step's words (prologue)
    qqq_static_for<4 * HW>([&](auto ic) { asm_load_d(dst[decltype(ic)::value / 4][decltype(ic)::value % 4], ic, so, std::false_type{}); });
  };
  unsigned scr[GROUPED ? P : 1][2];  // group scales (two fp16 each) of P stages, as loaded
  v4i x[MT];     // activation fragments of the current step; x[mt] is re-read for the next step right behind its last MFMA
  v4i aop[2][NQ]; // weight operands [set][2 * hf + b]: the current step's and the next step's
  const unsigned xrd = (unsigned)((lane & 15) * 128);  // + mt * 2048; chunk = (4 * t + h) ^ ((row >> 1) & 7), row = 16 * mt + j
  const int xsw = ((lane & 15) >> 1) & 7;
  const unsigned xrd_t[2] = {xrd + (unsigned)(((0 + h) ^ xsw) << 4), xrd + (unsigned)(((4 + h) ^ xsw) << 4)};

  auto read_x = [&](const int buf, const int t, const int mt) {
    x[mt] = *reinterpret_cast<const v4i*>(smem + buf * XB + xrd_t[t] + mt * 2048);
  };
  // the loop's re-reads as inline asm (QQQ_WIDE_XWAIT): LDS byte address (the dynamic LDS starts at 0, as the LDS-DMA's M0 presumes) = lane part + a 16-bit
  // immediate; the stage buffers span more than 64 KiB, so there is a second lane part 64 KiB up
  // (not in the tile walk: with asm-defined fragments in every stage's basic block hipcc's allocator gives up -- 167 - 227 spilled registers when tried)
  constexpr bool XW = QQQ_WIDE_XWAIT != 0 && QQQ_WIDE_BALANCE != 0 && !(QQQ_WIDE_ABLATE & 16) && !CHAIN;
  static_assert(!XW || (wide_xw_abs(MODE, MT, HW, 0) >= 0 && wide_xw_abs(MODE, MT, HW, MT / 4 - 1) >= 0), "every group of four m-tiles has a slot for its wait");
  const unsigned xrd_hi[2] = {xrd_t[0] + 65536u, xrd_t[1] + 65536u};
  auto read_x_asm = [&](auto bufc, auto tc, auto mtc) __attribute__((always_inline)) {
    constexpr int imm = decltype(bufc)::value * XB + decltype(mtc)::value * 2048, tt = decltype(tc)::value;
    (void)xrd_hi[0], (void)xrd_t[0], (void)x[0];
    if constexpr (imm >= 65536) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[decltype(mtc)::value]) : "v"(xrd_hi[tt]), "n"(imm - 65536));
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[decltype(mtc)::value]) : "v"(xrd_t[tt]), "n"(imm));
  };

  // ---- the unpack of one 32-column half (hf) of a step, cut into pieces that are placed one by one between the MFMAs ----
  // transpose: the two butterfly stages of quad_transpose4 (qqq_common.hip.h) as four 3-instruction pieces (lane masks kept
  // in SGPR pairs, one s_mov_b64 into VCC per piece); pieces sit in different issue slots, which also covers the two wait
  // states a DPP source needs behind the VALU that wrote it
  unsigned z[4], y[4];
  const unsigned long long km5 = 0x5555555555555555ull, kma = 0xaaaaaaaaaaaaaaaaull, km3 = 0x3333333333333333ull, kmc = 0xccccccccccccccccull;
  auto tr_piece = [&](auto pc, const v4u& w) {
    constexpr int pi = decltype(pc)::value;
    (void)y[0];  // (odr-use outside the discarded branches: clang does not capture from inside them)
    (void)z[0];
    if constexpr (pi == 0)       // z0 = even ? w0 : w1',  z2 = even ? w2 : w3'
      asm volatile("s_mov_b64 vcc, %6\n\t"
                   "v_cndmask_b32_dpp %0, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_cndmask_b32_dpp %1, %5, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                   : "=&v"(z[0]), "=&v"(z[2]) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "s"(km5) : "vcc");
    else if constexpr (pi == 1)  // z1 = odd ? w1 : w0',  z3 = odd ? w3 : w2'
      asm volatile("s_mov_b64 vcc, %6\n\t"
                   "v_cndmask_b32_dpp %0, %2, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_cndmask_b32_dpp %1, %4, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                   : "=&v"(z[1]), "=&v"(z[3]) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "s"(kma) : "vcc");
    else if constexpr (pi == 2)  // y0 = lo ? z0 : z2'',  y1 = lo ? z1 : z3''
      asm volatile("s_mov_b64 vcc, %6\n\t"
                   "v_cndmask_b32_dpp %0, %4, %2, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_cndmask_b32_dpp %1, %5, %3, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                   : "=&v"(y[0]), "=&v"(y[1]) : "v"(z[0]), "v"(z[1]), "v"(z[2]), "v"(z[3]), "s"(km3) : "vcc");
    else                         // y2 = hi ? z2 : z0'',  y3 = hi ? z3 : z1''
      asm volatile("s_mov_b64 vcc, %6\n\t"
                   "v_cndmask_b32_dpp %0, %2, %4, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_cndmask_b32_dpp %1, %3, %5, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                   : "=&v"(y[2]), "=&v"(y[3]) : "v"(z[0]), "v"(z[1]), "v"(z[2]), "v"(z[3]), "s"(kmc) : "vcc");
  };
  // the same four pieces one INSTRUCTION at a time (QQQ_WIDE_BALANCE): si = 0 the piece's VCC write, 1 / 2 its two selects.  VCC carries from one statement to the
  // next: nothing else in the steady-state loop writes it (the asm statements keep their order; the compiled loop is checked by tests/test_code_object_cpu.py)
  auto tr_single = [&](auto pc, auto sc, const v4u& w) {
    constexpr int pi = decltype(pc)::value, si = decltype(sc)::value;
    (void)y[0];
    (void)z[0];
    if constexpr (si == 0) {
      asm volatile("s_mov_b64 vcc, %0" : : "s"(pi == 0 ? km5 : pi == 1 ? kma : pi == 2 ? km3 : kmc) : "vcc");
    } else if constexpr (pi == 0) {
      if constexpr (si == 1) asm volatile("v_cndmask_b32_dpp %0, %2, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(z[0]) : "v"(w[0]), "v"(w[1]));
      else asm volatile("v_cndmask_b32_dpp %0, %2, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(z[2]) : "v"(w[2]), "v"(w[3]));
    } else if constexpr (pi == 1) {
      if constexpr (si == 1) asm volatile("v_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(z[1]) : "v"(w[0]), "v"(w[1]));
      else asm volatile("v_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(z[3]) : "v"(w[2]), "v"(w[3]));
    } else if constexpr (pi == 2) {
      if constexpr (si == 1) asm volatile("v_cndmask_b32_dpp %0, %2, %1, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(y[0]) : "v"(z[0]), "v"(z[2]));
      else asm volatile("v_cndmask_b32_dpp %0, %2, %1, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(y[1]) : "v"(z[1]), "v"(z[3]));
    } else {
      if constexpr (si == 1) asm volatile("v_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(y[2]) : "v"(z[0]), "v"(z[2]));
      else asm volatile("v_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(y[3]) : "v"(z[1]), "v"(z[3]));
    }
  };
  // per-channel: 12 VALU (and | shift, and per packed word); per-group: 8 x dequant_group4 in 4 two-instruction parts
  const unsigned nmask = QQQ_NIB_MASK;
  h2 sb[2];                    // per-group: the half's two group scales, broadcast
  unsigned gt0[2] = {0, 0}, gt1[2] = {0, 0};  // per-group: the two re-quantisations (b = 0, 1) of a packed word in flight
  h2 gha[2], ghb[2];
  constexpr int UPARTS = GROUPED ? 32 : 12;
  auto un_setup = [&](const h2 sc) {
    if constexpr (GROUPED) {
      sb[0] = (h2){sc[0], sc[0]};
      sb[1] = (h2){sc[1], sc[1]};
    }
  };
  auto un_part = [&](auto pc, auto hfc, v4i (&a)[NQ], const unsigned (&y)[4]) {  // y: the half's four words (the transpose's output, or -- DWORD -- the ring registers themselves)
    constexpr int pi = decltype(pc)::value, hf = decltype(hfc)::value;
    if constexpr (GROUPED) {
      // the two re-quantisations of a packed word (b = 0, 1) run in lock step: part p of b = 0, then part p of b = 1 -- a
      // v_pk_fma_f16's consumer (v_perm_b32) must not be the next VALU instruction (one wait state, which hipcc fills with an
      // s_nop: it does not count the inline-asm MFMA in between), and the sibling's part sits there for free
      constexpr int kq = pi / 8, part = (pi % 8) / 2, b = pi % 2;
      if constexpr (part == 0) {
        const unsigned qv = b ? (y[kq] >> 8) : y[kq];
        const unsigned magic = qqq_fp16_1024x2();
        gt0[b] = (qv & 0x000f000fu) | magic;  // {1024+p0, 1024+p4}
        gt1[b] = (qv & 0x00f000f0u) | magic;  // {1024+16*p1, 1024+16*p5}
      } else if constexpr (part == 1) {
        const h2 c_sub = {(_Float16)-1032.0f, (_Float16)-1032.0f};
        const h2 c_mul = {(_Float16)0.0625f, (_Float16)0.0625f};
        const h2 c_add = {(_Float16)-72.0f, (_Float16)-72.0f};
        gha[b] = __builtin_bit_cast(h2, gt0[b]) + c_sub;                                    // exact
        ghb[b] = __builtin_elementwise_fma(__builtin_bit_cast(h2, gt1[b]), c_mul, c_add);  // exact
      } else if constexpr (part == 2) {
        const h2 c_mag = {(_Float16)1152.0f, (_Float16)1152.0f};
        gha[b] = __builtin_elementwise_fma(gha[b], sb[b], c_mag);
        ghb[b] = __builtin_elementwise_fma(ghb[b], sb[b], c_mag);
      } else {
        int w = (int)(__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, ghb[b]), __builtin_bit_cast(unsigned, gha[b]), 0x06040200u) ^ 0x80808080u);
        // (per-group: no holds in the plain kernel -- the slots' sched_barriers keep a part where it was dealt, and an asm statement that so much as READS the result
        // of a v_pk_fma_f16 makes hipcc put an s_nop in front of it: 105 of them per trip when tried)
        if constexpr (CHAIN) asm volatile("" : "+v"(w));  // (see below)
        a[2 * hf + b][kq] = w;
      }
    } else {
      // (CHAIN: every stage is a basic block of its own -- a seam may follow -- and hipcc sinks an operand word that is only
      // used by the next stage's MFMAs into that block, in front of its first MFMA: 24 VALU instructions per stage with the
      // matrix pipe idle.  The empty asm holds each word where its slot put it.)
      constexpr int kq = pi / 3, part = pi % 3;
      if constexpr (part == 0) {
        int w = (int)(y[kq] & nmask);                                            // odd nibbles  -> 16*w4 of column n      (b = 0)
        if constexpr (QQQ_WIDE_BALANCE != 0) asm volatile("" : : "v"(w));  // (a USE only: behind an asm that DEFINES a register hipcc puts an s_nop in front of its next reader)
        else if constexpr (CHAIN) asm volatile("" : "+v"(w));
        a[2 * hf][kq] = w;
      } else if constexpr (part == 1) {
        gt0[0] = y[kq] << 4;
        if constexpr (QQQ_WIDE_BALANCE != 0) asm volatile("" : : "v"(gt0[0]));
        else if constexpr (CHAIN) asm volatile("" : "+v"(gt0[0]));
      } else {
        int w = (int)(gt0[0] & nmask);                                           // even nibbles -> 16*w4 of column n + 8  (b = 1)
        if constexpr (QQQ_WIDE_BALANCE != 0) asm volatile("" : : "v"(w));
        else if constexpr (CHAIN) asm volatile("" : "+v"(w));
        a[2 * hf + 1][kq] = w;
      }
    }
  };
  // behind the loop: the last step's re-reads are never used -- their registers must stay theirs until they have landed
  auto drain_x = [&]() __attribute__((always_inline)) {
    if constexpr (XW) {
      qqq_static_for<MT / 4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        (void)x[0];  // (odr-use: clang does not capture what only an asm operand of a generic lambda names)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[4 * j]), "+v"(x[4 * j + 1]), "+v"(x[4 * j + 2]), "+v"(x[4 * j + 3]));
      });
    }
  };
  auto mfma = [&](v4i& c, const auto& wa, const v4i& xb) {
    // inline asm: the accumulator is updated IN PLACE in the accumulation registers.  (The builtin selects the untied
    // form there, and with all 256 of them live hipcc's allocator bounces accumulators through VGPRs and scratch.)
    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c) : "v"(wa), "v"(xb));
  };

  // One 64-k step (stage i, half t of it; u = i % P and t compile-time).  Slot k of a step = MFMA (m-tile k / 4, column set k % 4) + its share of
  // everything else.  Step s unpacks step s + 1 into the other operand set (2 x (4 transpose pieces + UPARTS parts), evenly
  // over the 4 MT slots), re-reads fragment x[mt] for step s + 1 right behind its fourth MFMA, refills ring slot s % RS (read by
  // the unpack that ran during step s - 1) with step s + RS, and issues the LDS-DMA of one 16-byte chunk per lane of activation
  // stage i + LA every 16 slots.  The waits are hand-counted from the static schedule (wide_loads_between).
  unsigned st_xso = 0, st_swo[2] = {0, 0}, st_sco = 0;  // CHAIN: the scalar offsets of the current stage's loads (do_stage_chain)
  constexpr bool PREP = CHAIN && QQQ_WIDE_CHAINPREP != 0 && QQQ_WIDE_BALANCE != 0;
  unsigned pz_xso[2] = {0, 0}, pz_swo[2][2] = {{0, 0}, {0, 0}}, pz_sco[2] = {0, 0};  // PREP: two sets, [stage parity]: the stage reads one while the next one's is written
  // one set from the running cursors (common case: the next stage's loads stay inside the tile), cursors advanced; in four pieces for four slots
  auto prep_piece = [&](auto setc, auto jc) __attribute__((always_inline)) {
    constexpr int ns = decltype(setc)::value, j = decltype(jc)::value;
    (void)pz_xso[0], (void)pz_swo[0][0], (void)pz_sco[0], (void)cx_so, (void)cr_so, (void)cc_so;
    // (sgpr(): a no-op on a value that is scalar already; it only tells hipcc so where it cannot see it -- the cursors themselves are not pinned: an "s" operand
    // of a value hipcc believes to be in a VGPR does not compile)
    if constexpr (j == 0) {
      pz_xso[ns] = sgpr(cx_so);
      cx_so += 128u;
      asm volatile("" : "+s"(pz_xso[ns]));
    } else if constexpr (j == 1) {
      pz_swo[ns][0] = sgpr(cr_so);
      pz_swo[ns][1] = sgpr(cr_so + wstep);
      asm volatile("" : "+s"(pz_swo[ns][0]), "+s"(pz_swo[ns][1]));
    } else if constexpr (j == 2) {
      cr_so += 2u * wstep;
    } else if constexpr (GROUPED) {
      pz_sco[ns] = sgpr(cc_so);
      cc_so += (unsigned)N * 2u;
      asm volatile("" : "+s"(pz_sco[ns]));
    }
  };
  auto prep_set = [&](auto setc) __attribute__((always_inline)) { qqq_static_for<4>([&](auto jc) { prep_piece(setc, jc); }); };
  auto step = [&](const int i, auto uc, auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value, u = decltype(uc)::value;
    constexpr int cur = t, nxt = 1 - t;         // 2 P steps per trip: the step's parity is its t
    constexpr int sl = (2 * u + t) % RS, sn = (sl + 1) % RS;
    const int step_abs = 2 * i + t;
    constexpr int su = GROUPED ? ((t == 1) ? (u + 1) % P : u) : 0;  // scales of the NEXT step's stage
    const int st_x = i + LA < NST ? i + LA : NST - 1;
    // (CHAIN: i is not used, do_stage_chain says where the loads read; plain kernel with cursors: the running offsets themselves, advanced in slots of their own below)
    const unsigned cswo = PREP ? pz_swo[u & 1][t] : st_swo[t], csco = PREP ? pz_sco[u & 1] : st_sco;
    constexpr int BSLOT = wide_barrier_slot(MODE, MT, HW);
    static_assert(!CUR || (wide_ring_inc_slot(MODE, MT, HW) > 0 && (!GROUPED || wide_scale_inc_slot(MODE, MT, HW) > wide_scale_slot(HW))), "every cursor has a slot behind its last use");
    constexpr int NI = HW * (4 + UPARTS);
    auto slot = [&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value, mt = k / NQ, q = k % NQ;
      constexpr int SD = QQQ_WIDE_STAGGER <= 0 ? 1 : (HW == 2 ? QQQ_WIDE_STAGGER : 1);  // (128-column tiles: the first M0 slot is slot 4)
      if constexpr (QQQ_WIDE_STAGGER > 0 && t == 0 && k % SD == 0 && k / SD < 4) {
        // the barrier that ends the PREVIOUS stage, taken by wave 3 - k / d only (wave-uniform scalar branch around one s_barrier).  What it orders is a stage away
        // on both sides: the buffer this stage's LDS-DMA overwrites (first chunk: slot wide_dma_slot > 3 d) was last read a step before the previous stage ended,
        // the buffer it publishes is first read in this stage's second step; the wave's own share of that buffer has landed (vmcnt wait at the end of the
        // previous stage).  No LDS drain in front of it: the fragment reads in flight belong to this step.
        static_assert(3 * SD < (HW == 2 ? 8 : 4), "every wave passes the barrier before the stage's first LDS-DMA (M0 slot)");
        (void)wn;
        asm volatile("s_cmp_lg_u32 %0, %1\n\ts_cbranch_scc1 1f\n\ts_barrier\n1:" : : "s"(wn), "n"(3 - k / SD) : "scc");
      }
      if constexpr (W8) mfma(acc[mt][q], wr[sl][q], x[mt]);
      else mfma(acc[mt][q], aop[cur][q], x[mt]);
      if constexpr (QQQ_WIDE_BALANCE != 0) __builtin_amdgcn_sched_barrier(0);  // (the slot's other instructions BEHIND its MFMA: hipcc likes to hoist a free VALU instruction in front of it, i.e. into the previous slot)
      if constexpr (W8) {
        // the operands of step s + 1 (ring slot sn): fetched during step s + 1 - RL, the last of them at that step's last refill slot; landed once at
        // most the loads issued since are outstanding.  One wait per step, behind the step's last MFMA.
        if constexpr (k == NSLOT - 1) {
          constexpr int younger_all = wide_loads_between(MODE, MT, HW, 0, wide_w8_last_refill_slot(MT, HW), RL - 1, NSLOT);
          constexpr int younger = younger_all < 63 ? younger_all : 63;
          if constexpr (HW == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wr[sn][0]), "+v"(wr[sn][1]), "+v"(wr[sn][2]), "+v"(wr[sn][3]) : "n"(younger));
          else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wr[sn][0]), "+v"(wr[sn][1]) : "n"(younger));
        }
      } else if constexpr (!(QQQ_WIDE_ABLATE & 4) && QQQ_WIDE_BALANCE != 0) {
        // one-instruction items dealt by slot capacity (see QQQ_WIDE_BALANCE above)
        constexpr int PER = wide_bal_items(MODE);
        constexpr int FIX = CUR ? 1 : PREP ? 2 : 0;
        constexpr int lo = WideBalTable<MODE, MT, HW, t, FIX>::tab.before[k], hi = WideBalTable<MODE, MT, HW, t, FIX>::tab.before[k + 1];
        qqq_static_for<(hi - lo)>([&](auto jc) {
          constexpr int it = lo + decltype(jc)::value;
          constexpr int hf = it / PER, w_ = it % PER;
          if constexpr (w_ == 0) {
            // ring slot sn, half hf: loaded RS - 1 steps ago, its last load at slot wide_ring_last_slot (2 + 4 hf; DWORD: the half's fourth word); everything older (the
            // group scales of this stage among it) has landed once at most the loads issued since are outstanding (vmcnt is a 6-bit counter: more than 63 loads waits a little early, never late)
            constexpr int younger_all = wide_loads_between(MODE, MT, HW, (t + RS - 1) & 1, wide_ring_last_slot(MODE, MT, HW, hf), RS - 1, k);
            constexpr int younger = younger_all < 63 ? younger_all : 63;
            if constexpr (DW) {
              (void)wq[0][0][0];
              // (CHAIN, 32-column waves: the second scale word is fetched and never used; tied in here it stays allocated until it has landed -- see the packed branch below)
              if constexpr (GROUPED && CHAIN && HW == 1) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(wq[sn][hf][0]), "+v"(wq[sn][hf][1]), "+v"(wq[sn][hf][2]), "+v"(wq[sn][hf][3]), "+v"(scr[su][0]), "+v"(scr[su][1]) : "n"(younger));
              else if constexpr (GROUPED) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(wq[sn][hf][0]), "+v"(wq[sn][hf][1]), "+v"(wq[sn][hf][2]), "+v"(wq[sn][hf][3]), "+v"(scr[su][hf]) : "n"(younger));
              else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wq[sn][hf][0]), "+v"(wq[sn][hf][1]), "+v"(wq[sn][hf][2]), "+v"(wq[sn][hf][3]) : "n"(younger));
            } else if constexpr (GROUPED && CHAIN && HW == 1) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(wr[sn][hf]), "+v"(scr[su][0]), "+v"(scr[su][1]) : "n"(younger));
            else if constexpr (GROUPED) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wr[sn][hf]), "+v"(scr[su][hf]) : "n"(younger));
            else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wr[sn][hf]) : "n"(younger));
            un_setup(__builtin_bit_cast(h2, scr[su][hf]));
          } else if constexpr (DW) {
            un_part(std::integral_constant<int, w_ - 1>{}, std::integral_constant<int, hf>{}, aop[nxt], wq[sn][hf]);
          } else if constexpr (w_ < 13) {
            tr_single(std::integral_constant<int, (w_ - 1) / 3>{}, std::integral_constant<int, (w_ - 1) % 3>{}, wr[sn][hf]);
          } else {
            un_part(std::integral_constant<int, w_ - 13>{}, std::integral_constant<int, hf>{}, aop[nxt], y);
          }
        });
      } else if constexpr (!(QQQ_WIDE_ABLATE & 4)) {
        // MT == 16: the NI items go to the NE = 34 memory-free slots of the step (per-channel one each, per-group 2-3 each)
        constexpr bool MAPPED = HW == 1 || ((QQQ_WIDE_SLOTMAP & (GROUPED ? 2 : 1)) != 0 && (MT == 16 || (QQQ_WIDE_SLOTMAP & 4) != 0));
        constexpr int NE = wide_item_slots_before(HW, NSLOT), e = wide_item_slots_before(HW, k);
        constexpr bool here = wide_item_slot(HW, k);
        constexpr int lo = MAPPED ? (here ? (e * NI) / NE : 0) : (k * NI) / NSLOT;
        constexpr int hi = MAPPED ? (here ? ((e + 1) * NI) / NE : 0) : ((k + 1) * NI) / NSLOT;
        qqq_static_for<(hi - lo)>([&](auto jc) {
          constexpr int it = lo + decltype(jc)::value;
          constexpr int hf = it / (4 + UPARTS), w_ = it % (4 + UPARTS);
          if constexpr (w_ < 4) {
            if constexpr (w_ == 0) {
              // ring slot sn, half hf: loaded RS - 1 steps ago at slot 2 + 4 hf; everything older (the group scales of this
              // stage among it) has landed once at most the loads issued since are outstanding
              // (vmcnt is a 6-bit counter: a deeper ring than 63 loads waits a little early, never late)
              constexpr int younger_all = wide_loads_between(MODE, MT, HW, (t + RS - 1) & 1, 2 + 4 * hf, RS - 1, k);
              constexpr int younger = younger_all < 63 ? younger_all : 63;
              // (CHAIN, 32-column waves: the second scale word is fetched -- the load counts do not depend on HW -- and never
              // used; tied in here it stays allocated until it has landed.  The plain kernel keeps it live by using it behind its loop.)
              if constexpr (GROUPED && CHAIN && HW == 1) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(wr[sn][hf]), "+v"(scr[su][0]), "+v"(scr[su][1]) : "n"(younger));
              else if constexpr (GROUPED) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wr[sn][hf]), "+v"(scr[su][hf]) : "n"(younger));
              else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wr[sn][hf]) : "n"(younger));
              un_setup(__builtin_bit_cast(h2, scr[su][hf]));
            }
            tr_piece(std::integral_constant<int, w_>{}, wr[sn][hf]);
          } else {
            un_part(std::integral_constant<int, w_ - 4>{}, std::integral_constant<int, hf>{}, aop[nxt], y);
          }
        });
      } else if constexpr (k == 0) {
        qqq_static_for<NQ>([&](auto qc) {
          constexpr int qq = decltype(qc)::value;
          aop[nxt][qq] = (v4i){(int)wr[sn][qq / 2][0], (int)wr[sn][qq / 2][1], (int)wr[sn][qq / 2][2], (int)wr[sn][qq / 2][3]};
        });
      }
      if constexpr (XW) {
        if constexpr (wide_frag_slot(HW, k)) read_x_asm(std::integral_constant<int, (t == 0 ? (u % P) : ((u + 1) % P))>{}, std::integral_constant<int, (t == 0 ? 1 : 0)>{}, std::integral_constant<int, mt>{});
        constexpr int n = wide_xw_count_at(MODE, MT, HW, k);
        if constexpr (n >= 0) {  // the re-reads of the group(s) that wait here (issued a step ago; group 0: in this step) have landed
          // (not tied to the registers: their only readers are the MFMA statements, which keep their place behind this one -- and behind an asm that DEFINES a
          // register hipcc puts an s_nop in front of the next statement that reads it)
          asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(n));
        }
      } else if constexpr (wide_frag_slot(HW, k) && !(QQQ_WIDE_ABLATE & 16)) {
        read_x(t == 0 ? (u % P) : ((u + 1) % P), t == 0 ? 1 : 0, mt);
      }
      // (the order of the loads inside a slot is the order wide_loads_in_slot counts them in)
      if constexpr (GROUPED && t == 1 && k == wide_scale_slot(HW) && !(QQQ_WIDE_ABLATE & 8)) {
        if constexpr (CUR) load_sc_cur(scr[u]);
        else if constexpr (CHAIN) load_sc_so(csco, scr[u]);
        else load_sc(i + P, scr[u]);
      }
      if constexpr (PREP && t == 1) {  // the NEXT stage's offsets (the other set), piece by piece (wide_bal_cap leaves room for them)
        qqq_static_for<4>([&](auto jc) {
          if constexpr (k == wide_prep_slot(MODE, MT, HW, decltype(jc)::value)) prep_piece(std::integral_constant<int, (u + 1) & 1>{}, jc);
        });
      }
      if constexpr (CUR) {  // the cursors' scalar adds, one per slot (wide_bal_cap leaves room for them)
        (void)cx_so, (void)cr_so, (void)cc_so;
        if constexpr (k == wide_ring_inc_slot(MODE, MT, HW)) {
          cr_so += wstep;
          asm volatile("" : "+s"(cr_so));
        }
        if constexpr (t == 0 && k == wide_dma_inc_slot(MODE, MT, HW)) {
          cx_so += 128u;
          asm volatile("" : "+s"(cx_so));
        }
        if constexpr (GROUPED && t == 1 && k == wide_scale_inc_slot(MODE, MT, HW)) {
          cc_so += (unsigned)N * 2u;
          asm volatile("" : "+s"(cc_so));
        }
      }
      if constexpr (!(QQQ_WIDE_ABLATE & 8)) {  // ring refill, one 16-byte load per slot
        const int sw = step_abs + RL < KS ? step_abs + RL : KS - 1;
        const unsigned swo = CUR ? cr_so : CHAIN ? cswo : (unsigned)(2 * st0 + sw) * wstep;
        if constexpr (W8) {  // column set j of the slot the PREVIOUS step consumed
          constexpr int j = wide_w8_refill_index(MT, HW, k);
          if constexpr (j >= 0 && CUR) asm_load_w_cur(wr[(sl + RS - 1) % RS][j], std::integral_constant<int, j>{});
          else if constexpr (j >= 0) asm_load_w(wr[(sl + RS - 1) % RS][j], std::integral_constant<int, j>{}, swo);
        } else if constexpr (DW) {  // one word per slot: load 4 hf + kq of the step this ring slot holds next
          constexpr int li = wide_dw_load_index(MT, HW, k);
          if constexpr (li >= 0) {
            if constexpr (CUR) asm_load_d(wq[sl][li / 4][li % 4], std::integral_constant<int, li>{}, cr_so, std::true_type{});
            else asm_load_d(wq[sl][li / 4][li % 4], std::integral_constant<int, li>{}, swo, std::false_type{});
          }
        } else if constexpr (CUR) {
          if constexpr (wide_refill_slot(HW, k, 0)) asm_load_w_cur(wr[sl][0], std::integral_constant<int, 0>{});
          if constexpr (wide_refill_slot(HW, k, 1)) asm_load_w_cur(wr[sl][HW - 1], std::integral_constant<int, 1>{});
        } else {
          if constexpr (wide_refill_slot(HW, k, 0)) asm_load_w(wr[sl][0], std::integral_constant<int, 0>{}, swo);
          if constexpr (wide_refill_slot(HW, k, 1)) asm_load_w(wr[sl][HW - 1], std::integral_constant<int, 1>{}, swo);
        }
      }
      constexpr int DP = wide_dma_period(MT, HW);  // chunk (XPT / 2) t + k / DP of stage i + LA: M0, then the DMA
      if constexpr (wide_m0_slot(MT, HW, k) && !(QQQ_WIDE_ABLATE & 2))
        dma_m0(std::integral_constant<int, (u + LA) % P>{}, std::integral_constant<int, (XPT / 2) * t + k / DP>{});
      if constexpr (wide_dma_slot(MT, HW, k) && !(QQQ_WIDE_ABLATE & 2)) {
        if constexpr (CUR) dma_go_cur(std::integral_constant<int, (XPT / 2) * t + k / DP>{});
        else dma_go(std::integral_constant<int, (XPT / 2) * t + k / DP>{}, PREP ? pz_xso[u & 1] : CHAIN ? st_xso : (unsigned)(st0 + st_x) * 128u);
      }
      if constexpr (QQQ_WIDE_BALANCE != 0 && QQQ_WIDE_STAGGER == 0 && t == 1 && k == BSLOT) {
        // the stage's end, in a slot of its own (BALANCE): the LDS-DMA of stage i + 2, issued during stage i + 3 - P, is done when at most the loads issued since
        // its last chunk are outstanding; then the barrier.  A bare s_barrier: what it orders is this wave's share of that DMA (waited for here) and the fragment reads of
        // the buffer the next stage's DMA overwrites -- consumed by MFMAs a step ago; the reads in flight belong to the next stage and need no drain.
        constexpr int since = wide_loads_between(MODE, MT, HW, 1, wide_last_dma_slot(MT, HW), 2 * (P - 3), k + 1);
        static_assert(since < 64, "vmcnt is a 6-bit counter");
        if constexpr (!(QQQ_WIDE_ABLATE & 1)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"(since) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(since) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    qqq_static_for<NSLOT>(slot);
  };

  // ---- CHAIN: the tile seam.  The scales of a tile (token scales of its rows, channel scales and bias of its columns) sit in
  // one of two small LDS regions behind the stage buffers, written by LDS-DMA one seam earlier (region = parity of the tile's
  // index in the run: the region being overwritten was last read a whole tile ago, and every stage-end barrier since has
  // joined the waves), read here with plain ds_reads -- no vector-memory wait in the seam at all.  The flush itself: lane
  // (token j = lane & 15, c' = lane >> 4) holds, for column set q = (b, hf), register r <-> column 16 r + 8 b + 4 hf + c' of
  // its wave's 64; two v_permlane16_swap + two v_permlane32_swap per set exchange the register index r with the lane row c',
  // after which lane row rho holds columns 16 rho + 8 b + 4 hf + {0..3}: 16 consecutive columns of one token per lane
  // (HW = 1: two groups of four), converted and stored straight from the registers. ----
  constexpr int SCB = ROWS * 4 + BN * 6;  // bytes of a scale region: float s1[ROWS], float s2[BN] (stored order), fp16 bias[BN]
  constexpr int SC0 = P * XB;
  const int cej = lane & 15, cecp = lane >> 4;
  const int cegrp = HW == 2 ? wn : (wn >> 1), cehalf = wn & 1;
  auto scale_dma = [&](const int tm, const int tn, const int par) __attribute__((always_inline)) {
    const unsigned voff = (unsigned)((64 * wn + lane) * 4);
    const int rows = M - tm * ROWS < ROWS ? M - tm * ROWS : ROWS, cols = N - tn * BN < BN ? N - tn * BN : BN;
    auto one = [&](const void* base, const unsigned bytes, const unsigned dst) {  // 256 bytes per wave; past `bytes`: out of range
      v4u d = descriptor(base);
      d[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)bytes);
      const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)dst);
      // (s_nop 4: the descriptor may just have come out of a v_readlane -- 5 wait states before a vector-memory instruction reads it)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dword %0, %1, 0 offen lds" : : "v"(voff), "s"(d), "s"(m0v) : "memory");
    };
    const unsigned r0 = (unsigned)(SC0 + par * SCB + 256 * wn);
    if (wn < ROWS / 64) one(s1 + (size_t)tm * ROWS, (unsigned)rows * 4u, r0);
    if (wn < BN / 64) one(s2 + (size_t)tn * BN, (unsigned)cols * 4u, r0 + ROWS * 4);
    if (wn < BN / 128 && bias) one(bias + (size_t)tn * BN, (unsigned)cols * 2u, r0 + ROWS * 4 + BN * 4);
  };
  auto flush_tile = [&](const int tm, const int tn, const int par) __attribute__((always_inline)) {
    // (the MFMAs are inline asm: hipcc does not know that the accumulators it is about to read were written by the matrix pipe)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned char* sc = smem + SC0 + par * SCB;
    const float* s1t = reinterpret_cast<const float*>(sc);
    const float* s2t = reinterpret_cast<const float*>(sc + ROWS * 4);
    const _Float16* bt = reinterpret_cast<const _Float16*>(sc + ROWS * 4 + BN * 4);
    const _Float16 nzc = (_Float16)-0.0f;
    const int cl = 64 * cegrp + 16 * cecp;                   // this lane's 16 columns inside the strip, once transposed
    const bool cols_ok = tn * BN + 64 * cegrp < N;           // (N % 64 == 0: a wave's 64 columns are inside N or not at all)
    float2 dsa[NQ], dsb[NQ];
    h4 dbv[NQ];
    int dl[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      dl[q] = cl + 8 * (q & 1) + 4 * (HW == 2 ? (q >> 1) : cehalf);
      const int i0 = s2_stored_index(dl[q]);                 // (a tile starts on a multiple of 32 columns: the stored order is tile-local)
      dsa[q] = *reinterpret_cast<const float2*>(s2t + i0);
      dsb[q] = *reinterpret_cast<const float2*>(s2t + i0 + 8);
      dbv[q] = (h4){nzc, nzc, nzc, nzc};
    }
    if (bias) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) dbv[q] = *reinterpret_cast<const h4*>(bt + dl[q]);
    }
    v4i zero4 = {0, 0, 0, 0};
    asm volatile("" : "+v"(zero4));  // (a VGPR quad of zeros: the operand of the accumulator reset below)
    // Stores through buffer descriptors that END at the tile's last valid row: a row past M is out of range for the buffer unit
    // and dropped there -- no per-lane guard, no divergent branch anywhere in the tile walk (behind one, hipcc's structurizer
    // rebuilds the whole stage loop around exit flags and lane-mask branches: three taken branches per stage, and nothing
    // hides a fetch bubble when the wave is alone on its SIMD).  The row offset sits in the VECTOR offset, which the range
    // check certainly covers.
    const int rows = M - tm * ROWS < ROWS ? M - tm * ROWS : ROWS;
    const __amdgpu_buffer_rsrc_t dview = __builtin_amdgcn_make_buffer_rsrc(D + (size_t)tm * ROWS * N, 0, rows * N * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t aview = __builtin_amdgcn_make_buffer_rsrc(acc_out ? acc_out + (size_t)tm * ROWS * N : (int32_t*)D, 0, rows * N * 4, 0x00020000);
    const unsigned drow = (unsigned)(cej * N + tn * BN) * 2u;  // + 16 mt N * 2: this lane's token row, at the strip's first column
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float ts = s1t[16 * mt + cej];
      h4 o[NQ];
      // the 4 x 4 transposition between register index and lane row, two column sets per statement, in place: behind the
      // accumulator reads two wait states (VALU write -> row swap), every later swap has its two producers' worth of distance.
      // The accumulators are reset IN PLACE right behind their read: D = 0 x 0 + 0 (the tied operand keeps the register; a
      // fresh zero value would have to be coalesced with the loop-carried accumulator, and with all 256 accumulation registers
      // taken a failed coalescing is a spill)
#pragma unroll
      for (int q = 0; q < NQ; q += 2) {
        v4i ra = acc[mt][q], rb = acc[mt][q + 1];
        asm volatile("" : "+v"(ra), "+v"(rb));  // out of the accumulation registers, now (as whole quads: element by element hipcc moves the accumulators around instead)
        asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %1, 0" : "+a"(acc[mt][q]) : "v"(zero4));
        asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %1, 0" : "+a"(acc[mt][q + 1]) : "v"(zero4));
        int a0 = ra[0], a1 = ra[1], a2 = ra[2], a3 = ra[3], b0 = rb[0], b1 = rb[1], b2 = rb[2], b3 = rb[3];
        asm volatile("s_nop 1\n\t"
                     "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                     "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                     "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
                     "v_permlane32_swap_b32 %4, %6\n\tv_permlane32_swap_b32 %5, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
        if (acc_out && cols_ok) {  // (test hook, wave-uniform: the int32 accumulators, 4 consecutive columns per set)
          const unsigned ar = (drow + (unsigned)(16 * mt) * (unsigned)N * 2u) * 2u;
          __builtin_amdgcn_raw_buffer_store_b128((v4u){(unsigned)a0, (unsigned)a1, (unsigned)a2, (unsigned)a3}, aview, ar + (unsigned)dl[q] * 4u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128((v4u){(unsigned)b0, (unsigned)b1, (unsigned)b2, (unsigned)b3}, aview, ar + (unsigned)dl[q + 1] * 4u, 0, 0);
        }
        o[q] = epilogue_vals4(a0, a1, a2, a3, ts, dsa[q], dsb[q]) + dbv[q];  // (x + (-0.0) == x for every x)
        o[q + 1] = epilogue_vals4(b0, b1, b2, b3, ts, dsa[q + 1], dsb[q + 1]) + dbv[q + 1];
      }
      if (cols_ok) {  // (wave-uniform)
        const unsigned dr = drow + (unsigned)(16 * mt) * (unsigned)N * 2u;
        if constexpr (HW == 2) {  // column sets in column order: q = 0, 2, 1, 3
          const h8 lo8 = {o[0][0], o[0][1], o[0][2], o[0][3], o[2][0], o[2][1], o[2][2], o[2][3]};
          const h8 hi8 = {o[1][0], o[1][1], o[1][2], o[1][3], o[3][0], o[3][1], o[3][2], o[3][3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, lo8), dview, dr + (unsigned)cl * 2u, 0, QQQ_WIDE_FLUSH_AUX);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, hi8), dview, dr + (unsigned)cl * 2u + 16u, 0, QQQ_WIDE_FLUSH_AUX);
        } else {
          typedef unsigned v2u __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int q = 0; q < NQ; ++q) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o[q]), dview, dr + (unsigned)dl[q] * 2u, 0, QQQ_WIDE_FLUSH_AUX);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // m-tile by m-tile: the seam must not ask for more registers than the loop leaves free
    }
  };
  // flush the finished tile, then move on: the loads are already inside the next tile, whose scales go to the other region
  auto seam = [&]() __attribute__((always_inline)) {
    QQQ_WTRV(ch_pos < 7 ? 2 + 2 * ch_pos : 15, wall_clock64());
    flush_tile(cur_tm, cur_tn, ch_pos & 1);
    QQQ_WTRV(ch_pos < 7 ? 3 + 2 * ch_pos : 15, wall_clock64());
    ++ch_pos;
    cur_tm = nx_tm, cur_tn = nx_tn;
    cu_a_lo = nx_a_lo, cu_a_hi = nx_a_hi, cu_a_rec = nx_a_rec, cu_w_so = nx_w_so, cu_s_so = nx_s_so;
    if (ch_pos < ch_tiles) {
      scale_dma(cur_tm, cur_tn, ch_pos & 1);
      if (ch_pos + 1 < ch_tiles) locate(ch_first + (ch_pos + 1) * ch_stride, nx_tm, nx_tn);
      nx_tm = __builtin_amdgcn_readfirstlane(nx_tm), nx_tn = __builtin_amdgcn_readfirstlane(nx_tn);
      tile_ref(nx_tm, nx_tn, nx_a_lo, nx_a_hi, nx_a_rec, nx_w_so, nx_s_so);
    }
    cursors_at_stage0();
  };

  // ---- epilogue operands that do not depend on the accumulators: fetched here, 12 registers carried through the loop ----
  constexpr int RST = NT / (BN / 8);     // epilogue: rows a pass-step of the workgroup covers (8 / 16)
  const int c8 = tid % (BN / 8), er0 = tid / (BN / 8);  // this thread's 8 columns; rows er0 + RST * ps of a pass
  const int n = tile_n * BN + c8 * 8;
  float2 s2v[4] = {};
  const _Float16 nz = (_Float16)-0.0f;
  h8 bv = {nz, nz, nz, nz, nz, nz, nz, nz};
  if (!CHAIN && n < N) {  // (CHAIN: the seam reads the tile's scales from LDS; nothing is carried through the loop)
    const int i0s = s2_stored_index(n), i1s = s2_stored_index(n + 4);
    s2v[0] = *reinterpret_cast<const float2*>(s2 + i0s);
    s2v[1] = *reinterpret_cast<const float2*>(s2 + i0s + 8);
    s2v[2] = *reinterpret_cast<const float2*>(s2 + i1s);
    s2v[3] = *reinterpret_cast<const float2*>(s2 + i1s + 8);
    if (bias) bv = *reinterpret_cast<const h8*>(bias + n);
  }

  // ---- prologue: stages 0 .. LA - 1 by LDS-DMA, the weight ring, the scales; then EVERYTHING has landed (the loop's hand-counted
  // waits presuppose that nothing older than its own loads is in flight) ----
  if constexpr (CHAIN) {  // the same loads, read through the cursors (which end up LA stages / RS steps / P stages into the run)
    scale_dma(tile_m, tile_n, 0);
    qqq_static_for<LA>([&](auto jc) { dma_stage_so(jc, (unsigned)decltype(jc)::value * 128u); });
#pragma unroll
    for (int j = 0; j < RL; ++j) {
      if constexpr (DW) load_d_step(cu_w_so + (unsigned)j * wstep, wq[j]);
      else load_w_so(cu_w_so + (unsigned)j * wstep, wr[j]);
    }
    if constexpr (GROUPED) {
#pragma unroll
      for (int j = 0; j < P; ++j) load_sc_so(cu_s_so + (unsigned)j * (unsigned)N * 2u, scr[j]);
    }
  } else {
    qqq_static_for<LA>([&](auto jc) { dma_stage(jc, decltype(jc)::value); });
#pragma unroll
    for (int j = 0; j < RL; ++j) {
      if constexpr (DW) load_d_step((unsigned)(2 * st0 + (j < KS ? j : KS - 1)) * wstep, wq[j]);
      else load_w(j, wr[j]);
    }
    if constexpr (GROUPED) {
#pragma unroll
      for (int j = 0; j < P; ++j) load_sc(j, scr[j]);
    }
  }
  qqq_static_for<RL>([&](auto jc) {  // (the asm loads' results are tied to the wait: nothing may read them before it)
    constexpr int j = decltype(jc)::value;
    (void)wr[0];
    if constexpr (DW) {
      (void)wq[0][0][0];
      qqq_static_for<HW>([&](auto hfc) {
        constexpr int hf = decltype(hfc)::value;
        (void)wq[0][0][0];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(wq[j][hf][0]), "+v"(wq[j][hf][1]), "+v"(wq[j][hf][2]), "+v"(wq[j][hf][3]));
      });
    } else
    // (HW = 1: ONE operand -- the same variable tied twice gets two registers and a copy in front of the wait)
    if constexpr (W8 && HW == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wr[j][0]), "+v"(wr[j][1]), "+v"(wr[j][2]), "+v"(wr[j][3]));
    else if constexpr (W8 || HW == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wr[j][0]), "+v"(wr[j][1]));
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(wr[j][0]));
  });
  if constexpr (GROUPED) {
    qqq_static_for<P>([&](auto jc) {
      (void)scr[0];
      asm volatile("" : "+v"(scr[decltype(jc)::value][0]), "+v"(scr[decltype(jc)::value][1]));
    });
  }
  __syncthreads();
  if constexpr (XW) {
    // (asm as well, and drained here: a compiler-visible LDS read pending at the loop's entry makes hipcc wait for it INSIDE the loop body -- lgkmcnt(14) ... (0) in
    // front of the first step's MFMAs of EVERY trip, each of which also drains the asm re-reads in flight)
    qqq_static_for<MT>([&](auto mc) { read_x_asm(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, mc); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) read_x(0, 0, mt);
  }
  if constexpr (!W8) {
    qqq_static_for<HW>([&](auto hfc) {  // both halves of step 0 into operand set 0
      constexpr int hf = decltype(hfc)::value;
      un_setup(__builtin_bit_cast(h2, scr[0][hf]));
      if constexpr (DW) {
        (void)wq[0][0][0];
        qqq_static_for<UPARTS>([&](auto pc) { un_part(pc, hfc, aop[0], wq[0][hf]); });
      } else {
        qqq_static_for<4>([&](auto pc) { tr_piece(pc, wr[0][hf]); __builtin_amdgcn_sched_barrier(0); });
        qqq_static_for<UPARTS>([&](auto pc) { un_part(pc, hfc, aop[0], y); });
      }
    });
  }
  __builtin_amdgcn_sched_barrier(0);

  if constexpr (CUR) {  // the loop's loads of stage 0: LDS-DMA LA stages ahead, ring RL steps ahead, scales P stages ahead (relative to the slice's first stage)
    cx_so = sgpr((unsigned)(st0 + LA - 1) * 128u);  // (advanced in front of the stage's first chunk)
    cr_so = sgpr((unsigned)(2 * st0 + RL) * wstep);
    cc_so = sgpr((unsigned)(st0 + P) * (unsigned)N * 2u);
  }
  auto do_stage = [&](const int i, auto uc) __attribute__((always_inline)) {  // one 128-k stage: two steps and the barrier that publishes stage i + LA
    step(i, uc, std::integral_constant<int, 0>{});
    step(i, uc, std::integral_constant<int, 1>{});
    if constexpr (QQQ_WIDE_BALANCE != 0 && QQQ_WIDE_STAGGER == 0) return;  // (the stage-end wait and barrier sit in slot BSLOT of the second step)
    // the LDS-DMA of stage i + 2, issued during stage i + 3 - P: done when at most the loads issued since its last chunk
    // (the last DMA slot of that stage's second step) are outstanding, i.e. those of the P - 3 stages since
    constexpr int since = wide_loads_between(MODE, MT, HW, 1, wide_last_dma_slot(MT, HW), 2 * (P - 3), NSLOT);
    static_assert(since < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(since) : "memory");
    if constexpr (!(QQQ_WIDE_ABLATE & 1) && QQQ_WIDE_STAGGER == 0) __syncthreads();  // stage i + 2 is in LDS for everybody; buffer (i % P) is free
  };
  QQQ_WTR(1);
  if constexpr (CHAIN) {
    // ---- the flat stage sequence of this workgroup's whole run: the trip of P stages never ends, a seam follows whichever
    // stage completes a tile; after the last tile's flush the wave is done (s_endpgm waits for what is still in flight) ----
    static_assert(!CHAIN || P == 4, "the trip is written out for four stage buffers");
    int left = NST;
    auto do_stage_chain = [&](auto uc) __attribute__((always_inline)) {
      const unsigned sstep = (unsigned)N * 2u;
      constexpr int sset = decltype(uc)::value & 1;
      if constexpr (PREP) {
        if (__builtin_expect(left <= P, 0)) {  // the last P stages: the prepared set is overwritten -- the loads cross into the next tile one kind after the other
          const int done = NST - left;
          const bool xn = left <= LA;
          pz_xso[sset] = sgpr(xn ? (unsigned)(LA - left) * 128u : (unsigned)(done + LA) * 128u);
          xdesc[0] = sgpr(xn ? nx_a_lo : cu_a_lo), xdesc[1] = sgpr(xn ? nx_a_hi : cu_a_hi), xdesc[2] = sgpr(xn ? nx_a_rec : cu_a_rec);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int sn_ = 2 * done + t + RL;
            pz_swo[sset][t] = sgpr(sn_ < KS ? cu_w_so + (unsigned)sn_ * wstep : nx_w_so + (unsigned)(sn_ - KS) * wstep);
          }
          pz_sco[sset] = sgpr(nx_s_so + (unsigned)(P - left) * sstep);
          asm volatile("s_nop 4" : "+s"(pz_xso[sset]), "+s"(pz_swo[sset][0]), "+s"(pz_swo[sset][1]), "+s"(pz_sco[sset]), "+s"(xdesc));  // (settled here: see below)
        }
        // in SGPRs HERE, three MFMAs ahead of the first load that reads one: a set that reaches its stage through a loop-carried copy (hipcc keeps those in
        // VGPRs when SGPRs run short) would otherwise be read back by a v_readfirstlane right in front of the load, inside the 5 wait states (see below)
        pz_xso[sset] = sgpr(pz_xso[sset]), pz_swo[sset][0] = sgpr(pz_swo[sset][0]), pz_swo[sset][1] = sgpr(pz_swo[sset][1]), pz_sco[sset] = sgpr(pz_sco[sset]);
        asm volatile("" : "+s"(pz_xso[sset]), "+s"(pz_swo[sset][0]), "+s"(pz_swo[sset][1]), "+s"(pz_sco[sset]));
      } else
      if (__builtin_expect(left > P, 1)) {  // every load of this stage stays inside the tile
        // (sgpr(): a no-op on a value that is scalar already; it only tells hipcc so where it cannot see it)
        st_xso = sgpr(cx_so), st_swo[0] = sgpr(cr_so), st_swo[1] = sgpr(cr_so + wstep), st_sco = sgpr(cc_so);
        cx_so += 128u, cr_so += 2u * wstep, cc_so += sstep;
        asm volatile("" : "+s"(st_xso), "+s"(st_swo[0]), "+s"(st_swo[1]), "+s"(st_sco));
      } else {         // the last P stages: the loads cross into the next tile one kind after the other
        const int done = NST - left;                       // stages of this tile behind us
        const bool xn = left <= LA, cn = true;             // (left <= P: the scale load is always in the next tile)
        st_xso = sgpr(xn ? (unsigned)(LA - left) * 128u : (unsigned)(done + LA) * 128u);
        xdesc[0] = sgpr(xn ? nx_a_lo : cu_a_lo), xdesc[1] = sgpr(xn ? nx_a_hi : cu_a_hi), xdesc[2] = sgpr(xn ? nx_a_rec : cu_a_rec);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int sn_ = 2 * done + t + RL;               // the step the ring refill of step t fetches
          st_swo[t] = sgpr(sn_ < KS ? cu_w_so + (unsigned)sn_ * wstep : nx_w_so + (unsigned)(sn_ - KS) * wstep);
        }
        st_sco = sgpr(cn ? nx_s_so + (unsigned)(P - left) * sstep : 0u);
        // in SGPRs and settled HERE: a value that reaches its load through a VALU copy (v_readfirstlane / v_readlane) one slot
        // ahead of it is read inside the 5 wait states a VALU-written SGPR needs before a vector-memory instruction may use it
        // -- hipcc's own count is thrown off by the inline-asm MFMA in between (tools/check_vmem.py checks the compiled code)
        asm volatile("s_nop 4" : "+s"(st_xso), "+s"(st_swo[0]), "+s"(st_swo[1]), "+s"(st_sco), "+s"(xdesc));
      }
      step(0, uc, std::integral_constant<int, 0>{});
      step(0, uc, std::integral_constant<int, 1>{});
      if constexpr (QQQ_WIDE_BALANCE != 0 && QQQ_WIDE_STAGGER == 0) return;  // (the stage-end wait and barrier sit in slot BSLOT of the second step)
      constexpr int since = wide_loads_between(MODE, MT, HW, 1, wide_last_dma_slot(MT, HW), 2 * (P - 3), NSLOT);
      static_assert(since < 64, "vmcnt is a 6-bit counter");
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(since) : "memory");
      if constexpr (QQQ_WIDE_STAGGER == 0) __syncthreads();
    };
    if constexpr (PREP) prep_set(std::integral_constant<int, 0>{});  // the first stage's set (the cursors stand at the run's stage 0)
    bool more = true;
    // (the expectations put the seams and the tile-end offset code out of line: the common stage falls through from one MFMA
    // run into the next -- a taken branch is a fetch bubble nothing hides when the wave is alone on its SIMD)
#define QQQ_CHAIN_STAGE(U)                                    \
  if (__builtin_expect(more, 1)) {                            \
    do_stage_chain(std::integral_constant<int, U>{});         \
    if (__builtin_expect(--left == 0, 0)) {                   \
      seam();                                                 \
      if constexpr (PREP) prep_set(std::integral_constant<int, (U + 1) & 1>{}); /* the next tile's stage 0, from the cursors the seam restarted */ \
      left = NST;                                             \
      if (ch_pos == ch_tiles) {                               \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      \
        drain_x();                                            \
        more = false;                                         \
      }                                                       \
    }                                                         \
  }
    do {
      QQQ_CHAIN_STAGE(0)
      QQQ_CHAIN_STAGE(1)
      QQQ_CHAIN_STAGE(2)
      QQQ_CHAIN_STAGE(3)
    } while (more);
    return;
#undef QQQ_CHAIN_STAGE
  }
  // ---- steady state: P stages per iteration (ring slots and LDS buffers are compile-time), branch-free ----
  int i0 = 0;
  for (; i0 + P <= NST; i0 += P) {
#if defined(QQQ_PANEL_TRACE) && defined(QQQ_WIDE_TRACE_TRIPS)  // (perturbs the loop: an s_memtime per trip drains the LDS queue; without it the trace build's loop is the shipped one)
    if (i0 < 12 * P) QQQ_WTRV(4 + i0 / P, __builtin_amdgcn_s_memtime());  // shader clock at the top of the first 12 trips
#endif
    qqq_static_for<P>([&](auto uc) __attribute__((always_inline)) { do_stage(i0 + decltype(uc)::value, uc); });
  }
  qqq_static_for<P - 1>([&](auto uc) __attribute__((always_inline)) {  // ragged tail (< P stages)
    if (i0 + decltype(uc)::value < NST) do_stage(i0 + decltype(uc)::value, uc);
  });

  // the clamped LDS-DMAs of the last stages are still writing stage buffers: the epilogue image below lives in the same LDS.
  // The ring / scale registers are USED behind the wait: the last steps' refills are never consumed, and hipcc would hand the
  // destination of a dead asm load to the next value -- which the load then overwrites when it lands (seen: the transposed words of
  // column half 0 in the ragged tail).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  drain_x();
  if constexpr (DW) {
    qqq_static_for<RS * HW>([&](auto jc) {
      constexpr int j = decltype(jc)::value / HW, hf = decltype(jc)::value % HW;
      (void)wq[0][0][0];
      asm volatile("" : : "v"(wq[j][hf][0]), "v"(wq[j][hf][1]), "v"(wq[j][hf][2]), "v"(wq[j][hf][3]));
    });
  }
  qqq_static_for<RS>([&](auto jc) {
    (void)wr[0];
    if constexpr (DW) return;
    asm volatile("" : : "v"(wr[decltype(jc)::value][0]), "v"(wr[decltype(jc)::value][WRN - 1]));
    if constexpr (W8 && HW == 2) asm volatile("" : : "v"(wr[decltype(jc)::value][1]), "v"(wr[decltype(jc)::value][2]));
  });
  if constexpr (GROUPED) {
    qqq_static_for<P>([&](auto jc) {
      (void)scr[0];
      asm volatile("" : : "v"(scr[decltype(jc)::value][0]), "v"(scr[decltype(jc)::value][1]));
    });
  }
  // ---- epilogue: EPR rows at a time: int32 -> LDS (row-major, skewed rows) -> 8 consecutive n per thread -> 16-byte stores ----
  // D lane ln of the MFMA holds token j = ln & 15, rows 4 * (ln >> 4) + r -> c' = ln >> 4, jt = r:
  //   column inside the strip  nl = 64 * (column group of the wave) + 16 * jt + 8 * b + 4 * hf + c'
  QQQ_WTR(2);
  // (the MFMAs are inline asm: hipcc does not know that the accumulators it is about to read were written by the matrix pipe)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  int* ep = reinterpret_cast<int*>(smem);
  const int ej = lane & 15, ecp = lane >> 4;
  const int egrp = HW == 2 ? wn : (wn >> 1), ehalf = wn & 1;  // the wave's 64-column group of the strip (and, HW = 1, its half)
  constexpr int EP_ITEMS = EPR * (BN / 8), EP_PASSES = EP_ITEMS / NT;  // 16 rows per thread and pass
  // this wave's accumulators of EPR rows -> the row-major LDS image.  (always_inline: called from three places since the deposit
  // has two store policies, hipcc would otherwise leave it out of line and take the accumulators by ADDRESS -- i.e. keep all
  // 256 of them in scratch: 4 accumulation registers and 2900 scratch instructions in the first build)
  auto image = [&](const int pass) __attribute__((always_inline)) {
#pragma unroll
    for (int jm = 0; jm < EPR / 16; ++jm)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ep[(16 * jm + ej) * EP_STRIDE + 64 * egrp + 16 * r + 8 * (q & 1) + 4 * (HW == 2 ? (q >> 1) : ehalf) + ecp] = acc[pass * (EPR / 16) + jm][q][r];
  };

  // ---- in-launch split-K (ksplit > 1): arrival-order tickets as in the panel kernel.  A depositor sends its partial tile
  // through the SAME LDS transposition as the epilogue and writes it row-major ([ROWS][256] int32, full 1 KiB rows,
  // write-through) into slot `arrival` of the tile in the caller's reduce buffer C; the last arrival adds the slots to its
  // own image rows (agent-scope sc1 loads) on the way to the fp16 conversion.  Two ticket words per tile in `workspace`
  // (arrivals, completed deposits), zero again on exit. ----
  // ---- two slices, EXCHANGE (round 6; hflags & 8, tiles whose epilogue runs in two row halves): instead of one slice depositing its whole partial tile and
  // idling while the other folds 256 KiB and runs the whole epilogue, each slice deposits the row half the OTHER one will finish and finishes its own half
  // (slice sp owns rows [sp EPR, sp EPR + EPR)) -- both CUs busy, half the fold and half the epilogue on the critical path, even K slices (no skew).
  // A slice that waits for its partner's deposit must know the partner is RUNNING: HIP promises nothing about dispatch order or co-residency.  The arrival
  // word already says so: every slice publishes a valid nibble (XCC id) at kernel start.  The FIRST arrival decides -- partner's nibble valid: exchange (bit 16),
  // else classic (bit 17: it deposits everything and leaves, the partner folds) -- and records the decision with a second atomic; the second arrival reads it
  // (polling a moment if it arrives between the two atomics: the decider is past its loop and never waits in between).  An exchanging slice deposits before it
  // waits and waits only for a partner that has provably started and that itself deposits before it waits: no cycle, whatever else occupies the chip. ----
  constexpr unsigned XB_EXCH = 1u << 16, XB_CLASSIC = 1u << 17;
  const bool exch_on = ksplit == 2 && (hflags & 8) != 0 && ROWS / EPR == 2;
  int arrival = 0;
  bool exch = false;
  if (ksplit > 1) {
    int* tk = tickets + 2 * (size_t)tile_lin;
    int* xch = ep + EPR * EP_STRIDE;  // one word behind the image (a second __shared__ object would de-pipeline the main loop)
    if (tid == 0) {
      unsigned w = (unsigned)__hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (exch_on) {
        const unsigned arr = w & 0xffu;
        if (arr == 0) {  // first arrival: decide, record
          const unsigned dec = (((w >> (8 + 4 * (1 - sp))) & 8u) != 0) ? XB_EXCH : XB_CLASSIC;
          (void)__hip_atomic_fetch_or(tk, (int)dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w |= dec;
        } else {         // second arrival: the decision is there, or a moment away
          int spin = 0;
          while (!(w & (XB_EXCH | XB_CLASSIC))) {
            if (++spin > QQQ_SPIN_LIMIT) __builtin_trap();
            __builtin_amdgcn_s_sleep(1);
            w = (unsigned)__hip_atomic_load(tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          w = (w & ~0xffu) | arr;
        }
      }
      *xch = (int)w;
    }
    __syncthreads();
    const unsigned word = (unsigned)__builtin_amdgcn_readfirstlane(*xch);
    arrival = (int)(word & 0xffu);
    exch = exch_on && (word & XB_EXCH) != 0;
#ifndef QQQ_WIDE_TRACE_TRIPS
    QQQ_WTR(4);                                   // split K: ticket taken
    QQQ_WTRV(7, (arrival << 1) | (exch ? 1 : 0)); // ... arrival index, exchange decided
#endif
    const size_t slot_ints = (size_t)ROWS * BN;
    if (exch || arrival < ksplit - 1) {
      // Same XCD for every slice of this tile, as far as the arrival word shows at MY arrival?  Then the deposit may stay in
      // this XCD's L2.  The set of published nibbles only grows, and the finisher takes its ticket after every depositor: it
      // sees at least what I see.  So if I deposit L2-only, the finisher (and every other slice) is on my XCD and finds my
      // lines in the L2 we share; if any slice is elsewhere -- or has not even started -- I write through, and a write-through
      // store is right for a reader anywhere.  The finisher's loads are the same either way (agent scope: an L1 miss, an L2 hit
      // where the line is, memory otherwise).  hflags & 4 (tune.fused bit 4) forces the write-through path, for A/B timing.
      // (exchange: the one slot of the tile, each slice writing the row half it does not own; both nibbles are valid by then)
      bool local = ksplit <= 6 && !(hflags & 4);
      for (int j = 0; j < ksplit; ++j) local = local && ((word >> (8 + 4 * j)) & 15u) == (8u | my_xcc);
      const __amdgpu_buffer_rsrc_t sv = wide_view(C + ((size_t)tile_lin * (ksplit - 1) + (exch ? 0 : arrival)) * slot_ints);
      auto deposit = [&](auto auxc) __attribute__((always_inline)) {
        constexpr int aux = decltype(auxc)::value;
#pragma unroll
        for (int pass = 0; pass < ROWS / EPR; ++pass) {
          if (exch && pass == sp) continue;  // (wave-uniform) my own half stays here
          __syncthreads();
          image(pass);
          __syncthreads();
#pragma unroll
          for (int ps = 0; ps < EP_PASSES; ++ps) {
            const int row = er0 + RST * ps;
            const unsigned off = (unsigned)(((pass * EPR + row) * BN + c8 * 8) * 4);
            const v4i lo = *reinterpret_cast<const v4i*>(ep + row * EP_STRIDE + c8 * 8);
            const v4i hi4 = *reinterpret_cast<const v4i*>(ep + row * EP_STRIDE + c8 * 8 + 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, lo), sv, off, 0, aux);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, hi4), sv, off + 16, 0, aux);
          }
        }
      };
      if (local) deposit(std::integral_constant<int, 0>{});            // write-back: the line lives in this XCD's L2
      else deposit(std::integral_constant<int, /*sc0 sc1*/ 17>{});     // written through to memory
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // every wave's part of the deposit is where the finisher will look for it
      if (tid == 0) qqq_publish_add(tk + 1, hflags);
#ifndef QQQ_WIDE_TRACE_TRIPS
      QQQ_WTR(5);                                 // deposit drained and published
#endif
      if (!exch) return;
    }
  }

  // every token scale this thread needs, fetched up front (pass by pass each batch paid its own round trip) and pinned here:
  // left alone hipcc sinks every load to its use inside the guarded store below, one exposed round trip per row
  float a_s[ROWS / EPR][EP_PASSES];
#pragma unroll
  for (int pass = 0; pass < ROWS / EPR; ++pass)
#pragma unroll
    for (int ps = 0; ps < EP_PASSES; ++ps) {
      const int m = mbase + pass * EPR + er0 + RST * ps;
      a_s[pass][ps] = s1[m < M ? m : M - 1];
    }
  // interior tiles without the test hook: branch-free stores (a guard per row makes hipcc drain the memory queue per row)
  const bool interior = (mbase + ROWS <= M) && (tile_n * BN + BN <= N) && acc_out == nullptr;
  const bool fold = ksplit > 1;
  const int32_t* slots = C + (size_t)tile_lin * (size_t)(ksplit - 1) * ((size_t)ROWS * BN);
  // EP_PASSES / 4 rows at a time: the loads (image rows from LDS, and with a K split the deposits' rows from C) of all of
  // them are issued before the first conversion, branch-free; only the stores sit behind the edge guard
  constexpr int RB = EP_PASSES / 4;  // (a deeper batch spills in the fold path: 16 registers per row and slot in flight)
  auto out_rows = [&](const int pass, const int half, const bool guarded, const bool folding) {
    v4i lo[RB], hi4[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int row = er0 + RST * (half * RB + j);
      lo[j] = *reinterpret_cast<const v4i*>(ep + row * EP_STRIDE + c8 * 8);
      hi4[j] = *reinterpret_cast<const v4i*>(ep + row * EP_STRIDE + c8 * 8 + 4);
    }
    if (folding) {
      for (int sl = 0; sl < ksplit - 1; ++sl) {
        const __amdgpu_buffer_rsrc_t sv = agent_view(slots + (size_t)sl * ((size_t)ROWS * BN));
        v4i d0[RB], d1[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const unsigned off = (unsigned)(((pass * EPR + er0 + RST * (half * RB + j)) * BN + c8 * 8) * 4);
          d0[j] = load16_agent(sv, off);
          d1[j] = load16_agent(sv, off + 16);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          lo[j] += d0[j];
          hi4[j] += d1[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int ps = half * RB + j;
      const int m = mbase + pass * EPR + er0 + RST * ps;
      const h4 o0 = epilogue_vals4(lo[j][0], lo[j][1], lo[j][2], lo[j][3], a_s[pass][ps], s2v[0], s2v[1]);
      const h4 o1 = epilogue_vals4(hi4[j][0], hi4[j][1], hi4[j][2], hi4[j][3], a_s[pass][ps], s2v[2], s2v[3]);
      h8 o = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
      o = o + bv;  // fp16 add after the fp16 round; without a bias bv = -0.0: x + (-0.0) == x bit for bit for every x, +-0 included
      if (!guarded || (m < M && n < N)) {
        *reinterpret_cast<h8*>(D + (size_t)m * N + n) = o;
        if (guarded && acc_out) {
          *reinterpret_cast<v4i*>(acc_out + (size_t)m * N + n) = lo[j];
          *reinterpret_cast<v4i*>(acc_out + (size_t)m * N + n + 4) = hi4[j];
        }
      }
    }
  };
  const int first_pass = exch ? sp : 0;
#pragma unroll
  for (int pass = 0; pass < ROWS / EPR; ++pass) {
    if (exch && pass != sp) continue;  // (wave-uniform) exchange: the other half is the partner's
    __syncthreads();
    image(pass);
    if (pass == first_pass && fold) {
      // the last arrival: everybody it waits for has arrived already (is depositing) -- short, and bounded as a matter of
      // principle: a depositor that never completes must not end in a silently wrong D (the launch is aborted instead)
      // (exchange: both slices wait here for BOTH deposits -- their own is counted -- then count once more: whoever counts second knows
      // that nobody polls the words any longer and zeroes them)
      if (tid == 0) {
        int* tk = tickets + 2 * (size_t)tile_lin;
        int spin = 0;
        while (__hip_atomic_load(tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (exch ? 2 : ksplit - 1)) {
          if (++spin > QQQ_SPIN_LIMIT) __builtin_trap();
          __builtin_amdgcn_s_sleep(2);
        }
        if (!exch || __hip_atomic_fetch_add(tk + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3) {
          __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // workspace zero on return
          __hip_atomic_store(tk + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __syncthreads();
#ifndef QQQ_WIDE_TRACE_TRIPS
    if (pass == first_pass && fold) QQQ_WTR(6);   // the deposits this workgroup folds are complete
#endif
    if (pass == first_pass && fold && qqq_formal_acquire(hflags)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (off by default: qqq_common.hip.h)
    if (pass == first_pass) {  // (the pin: behind the first image's LDS writes, which cover the loads' round trip)
#pragma unroll
      for (int p2 = 0; p2 < ROWS / EPR; ++p2)
#pragma unroll
        for (int ps = 0; ps < EP_PASSES; ++ps) asm volatile("" : "+v"(a_s[p2][ps]));
    }
    if (interior && !fold) {
#pragma unroll
      for (int hb = 0; hb < EP_PASSES / RB; ++hb) out_rows(pass, hb, false, false);
    } else if (interior) {
#pragma unroll
      for (int hb = 0; hb < EP_PASSES / RB; ++hb) out_rows(pass, hb, false, true);
    } else {
#pragma unroll
      for (int hb = 0; hb < EP_PASSES / RB; ++hb) out_rows(pass, hb, true, fold);
    }
  }
#ifdef QQQ_PANEL_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  QQQ_WTR(3);
#endif
}

#endif  // QQQ_AMD_QQQ_WIDE_HIP_H_
