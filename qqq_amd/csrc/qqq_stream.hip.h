// qqq_stream.hip.h -- "stream" kernel (m <= 128, HBM-bound, weights HBM -> VGPR -> MFMA 16x16x64)
// Part of the single translation unit qqq_w4a8.hip (see its header comment for the design).
#ifndef QQQ_AMD_QQQ_STREAM_HIP_H_
#define QQQ_AMD_QQQ_STREAM_HIP_H_

// ------------------------------------------------------------------------------------------
// "stream" kernel: small m, weights HBM -> VGPR -> MFMA 16x16x64
// ------------------------------------------------------------------------------------------
//
// grid = (ceil(N/128) strips, ksplit, ceil(M / (16*MT)));  block = WAVES * 64.
// MFMA A operand lane l = (i = l & 15, h = l >> 4): i = 8*g + c  <->  weight columns
//     n = 128*strip + 64*g + 16*jt + 8*b + c   for the 8 MFMAs (jt, b) of a step,
//     k = 64*s + 16*h + [0,16)   (k-tile 4*s + h).
// MFMA B operand lane l = (j = l & 15, h): token m = mbase + 16*mt + j, same 16 k.
// MFMA D: lane l holds column j = l & 15 (token) and rows i = 4*(l >> 4) + r, r = 0..3, i.e.
//     g = l >> 5, c = 4*((l >> 4) & 1) + r  ->  4 consecutive n.

// QQQ_STREAM_LINES (round 6): whole-line weight loads.  In the mapping above a lane owns a whole 64-byte chunk (4 loads of 16 bytes at a 64-byte stride between lanes): every
// 128-byte line of the weight stream is touched by FOUR wave instructions, which is why non-temporal loads LOSE 11 % here while they gave the column kernel 10 %.  With
// LINES the wave loads like four column-kernel waves: load j (0 .. 3) of lane (h = l >> 4, cq = (l >> 2) & 3, q4 = l & 3) is the 16 bytes of piece q4 of chunk 4 (j & 1) + cq
// of 64-column group j >> 1 -- 256 contiguous bytes (two whole lines) per k-tile row and instruction, each line exactly once -- and a 4 x 4 quad transpose per load
// (quad_transpose4, as in the column kernel) hands MFMA lane (h, cq, jt = q4) the words kq = 0 .. 3 of its (chunk, jt): the A operand of MFMA (j, b) with rows i = 4 cq + jt,
// i.e. columns n = 128 strip + 64 (j >> 1) + 16 jt + 8 b + 4 (j & 1) + cq.  D lane l holds rows 4 (l >> 4) + r: cq = l >> 4 of the OUTPUT lane, jt = r -- four consecutive
// n sit in the four lanes token + 16 cq (the column kernel's write-out).  The loads are non-temporal (QQQ_W_NT bit 1 selects them for this mapping).
// Built, bit-exact (28 GPU tests on the build), measured cold against the mapping above (profiles/r06b_stream_whole_line_loads.txt): N = 8192, K = 21760 per-channel
// 16 / 32 / 48 tokens -2 ... -3 %, 64 tokens +1.5 %; per-group 16 tokens +13 %, 64 tokens +7 %; the Llama-2-7B layers +2 ... +11 % -- four quad transposes per step cost more
// VALU time than the non-temporal loads give back wherever the launch is not purely HBM-bound.  Off.
#ifndef QQQ_STREAM_LINES
#define QQQ_STREAM_LINES 0
#endif

template <int MT>
struct StreamStep {
  v4u w[4];   // w[kq][jt]
  v4i x[MT];  // activation operands
  h8 sc;      // per-group scales [2*jt + b]  (LINES: [2*j + b] of this lane's (chunk, jt), load by load)
};

template <int MT, bool GROUPED, int WAVES, int PF>
__global__ __launch_bounds__(WAVES * 64) void qqq_stream_kernel(
    const int8_t* __restrict__ A, const unsigned char* __restrict__ B, int32_t* __restrict__ C,
    _Float16* __restrict__ D, const float* __restrict__ s1, const float* __restrict__ s2,
    const _Float16* __restrict__ s3, int32_t* __restrict__ acc_out, int* __restrict__ tickets,
    const _Float16* __restrict__ bias, const int M, const int N, const int K, const int ksplit_sk,
    const int fused) {
  // (bits 16..23 of the K-split argument: `skew`, the 64-k steps the LAST K slice gets on top of an even share -- fused == 3 only)
  const int ksplit = ksplit_sk & 0xffff, skew = (ksplit_sk >> 16) & 0xff;
  constexpr int NQ = MT * 8;  // MFMA output tiles per wave
  __shared__ int red[NQ * 4 * 64 + 64];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = blockIdx.x;
  const int sp = blockIdx.y;
  const int mbase = blockIdx.z * (16 * MT);

  if constexpr (WAVES > 1) {
    for (int i = tid; i < NQ * 4 * 64; i += WAVES * 64) red[i] = 0;
    __syncthreads();
  }

  const int j = lane & 15;
  const int h = lane >> 4;
  const int g = j >> 3;
  const int c = j & 7;
  const int ngroups = N >> 6;
  int ng = strip * 2 + g;
  if (ng >= ngroups) ng = ngroups - 1;  // clamp (N % 128 == 64): loads stay legal, output dropped
  const size_t rowbytes = (size_t)N * 8;
  constexpr bool LINES = QQQ_STREAM_LINES != 0;
  const int cq = (lane >> 2) & 3, q4 = lane & 3;  // LINES: chunk within the load's four, piece as a load lane / jt as an MFMA lane
  int ngl[2] = {strip * 2, strip * 2 + 1};        // LINES: the strip's two 64-column groups (loads 0, 1 / 2, 3)
  if (ngl[1] >= ngroups) ngl[1] = ngroups - 1;
  const unsigned char* bptr = LINES ? B + (size_t)h * rowbytes + cq * 64 + q4 * 16 : B + (size_t)h * rowbytes + (size_t)ng * 512 + c * 64;

  const int8_t* xptr[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = mbase + 16 * mt + j;
    if (row >= M) row = M - 1;
    xptr[mt] = A + (size_t)row * K + 16 * h;
  }
  const _Float16* sptr = GROUPED ? (LINES ? s3 + cq * 8 + 2 * q4 : s3 + (size_t)ng * 64 + c * 8) : nullptr;

  const int KS = K >> 6;  // 64-k steps
  const int KSE = KS - skew;  // uneven slices (skew > 0): the last one is longer -- it arrives last and finds the other deposits complete
  const int ks_begin = (int)(((long long)KSE * sp) / ksplit);
  const int ks_end = (sp == ksplit - 1) ? KS : (int)(((long long)KSE * (sp + 1)) / ksplit);

  v4i acc[MT][4][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      acc[mt][jt][0] = (v4i){0, 0, 0, 0};
      acc[mt][jt][1] = (v4i){0, 0, 0, 0};
    }

  auto load_step = [&](const int s, StreamStep<MT>& r) {
    const unsigned char* p = bptr + (size_t)(4 * s) * rowbytes;
    if constexpr (LINES) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // w[j]: load j -- group j >> 1, chunks 4 (j & 1) .. + 3: two whole lines per k-tile row
        const unsigned char* pj = p + (size_t)ngl[j >> 1] * 512 + (j & 1) * 256;
        if constexpr ((QQQ_W_NT & 1) != 0) r.w[j] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(pj));  // streamed once, every line by one instruction
        else r.w[j] = *reinterpret_cast<const v4u*>(pj);
      }
    } else {
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        if constexpr ((QQQ_W_NT & 2) != 0) r.w[kq] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p + 16 * kq));  // streamed once
        else r.w[kq] = *reinterpret_cast<const v4u*>(p + 16 * kq);
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) r.x[mt] = *reinterpret_cast<const v4i*>(xptr[mt] + 64 * s);
    if constexpr (GROUPED && LINES) {
      const _Float16* ps = sptr + (size_t)(s >> 1) * N;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const h2 v = *reinterpret_cast<const h2*>(ps + (size_t)ngl[j >> 1] * 64 + (j & 1) * 32);
        r.sc[2 * j] = v[0];
        r.sc[2 * j + 1] = v[1];
      }
    } else if constexpr (GROUPED) {
      r.sc = *reinterpret_cast<const h8*>(sptr + (size_t)(s >> 1) * N);
    }
  };

  auto compute_step = [&](const StreamStep<MT>& r) {
    if constexpr (LINES) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned y[4];
        quad_transpose4(r.w[j], y);  // y[kq] (lane q4 = jt) = word jt of piece kq of this lane's chunk
        h2 sb0 = {(_Float16)0, (_Float16)0}, sb1 = sb0;
        if constexpr (GROUPED) {
          sb0 = (h2){r.sc[2 * j], r.sc[2 * j]};
          sb1 = (h2){r.sc[2 * j + 1], r.sc[2 * j + 1]};
        }
        v4i a0, a1;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          int w0, w1;
          unpack_pair<GROUPED>(y[kq], sb0, sb1, w0, w1);
          a0[kq] = w0;
          a1[kq] = w1;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt][j][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, r.x[mt], acc[mt][j][0], 0, 0, 0);
          acc[mt][j][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, r.x[mt], acc[mt][j][1], 0, 0, 0);
        }
      }
      return;
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      v4i a0, a1;
      h2 sb0 = {(_Float16)0, (_Float16)0}, sb1 = sb0;
      if constexpr (GROUPED) {
        sb0 = (h2){r.sc[2 * jt], r.sc[2 * jt]};
        sb1 = (h2){r.sc[2 * jt + 1], r.sc[2 * jt + 1]};
      }
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        int w0, w1;
        unpack_pair<GROUPED>(r.w[kq][jt], sb0, sb1, w0, w1);
        a0[kq] = w0;
        a1[kq] = w1;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][jt][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, r.x[mt], acc[mt][jt][0], 0, 0, 0);
        acc[mt][jt][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, r.x[mt], acc[mt][jt][1], 0, 0, 0);
      }
    }
  };

  // software pipeline: PF steps (4 KiB of weights each) in flight per wave.  The steady-state loop is
  // branch-free so that hipcc can place COUNTED s_waitcnt vmcnt(N) (loads of the younger ring slots
  // stay in flight while the oldest slot is consumed); the ragged tail takes the checked path.
  StreamStep<MT> ring[PF];
  int s = ks_begin + wave;
  if (s + (2 * PF - 1) * WAVES < ks_end) {
    // unconditional prologue + branch-free loop: the wait counters are exact on every path
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      load_step(s + p * WAVES, ring[p]);
      __builtin_amdgcn_sched_barrier(0);  // ring order == issue order, so vmcnt(N) can be counted
    }
    for (; s + (2 * PF - 1) * WAVES < ks_end; s += PF * WAVES) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        compute_step(ring[p]);
        __builtin_amdgcn_sched_barrier(0);  // keep the refill right behind its consumer (hipcc would
        load_step(s + (p + PF) * WAVES, ring[p]);  // otherwise sink all loads to the loop end)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (s + p * WAVES < ks_end) load_step(s + p * WAVES, ring[p]);
  }
  // fused == 3: the arrival ticket of this K slice, taken HERE -- a few steps ahead of the end of the loop, so that the atomic's round trip
  // (agent scope, ~1 us) runs under the tail and the LDS reduction; consumed behind the barrier below.  (Taking it early is safe: whoever
  // holds the last ticket only ever waits for workgroups that are resident and running.)
  int arrival = 0;
  if (fused == 3 && ksplit > 1 && tid == 0)
    arrival = __hip_atomic_fetch_add(tickets + 2 * (size_t)(blockIdx.z * gridDim.x + strip), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (; s < ks_end; s += PF * WAVES) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int sc = s + p * WAVES;
      if (sc < ks_end) {
        compute_step(ring[p]);
        const int sn = sc + PF * WAVES;
        if (sn < ks_end) load_step(sn, ring[p]);
      }
    }
  }

  // ---- reduce the waves of this workgroup through LDS (int adds: order-independent) ----
  if constexpr (WAVES > 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __hip_atomic_fetch_add(&red[(((mt * 4 + jt) * 2 + b) * 4 + r) * 64 + lane],
                                   acc[mt][jt][b][r], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            red[(((mt * 4 + jt) * 2 + b) * 4 + r) * 64 + lane] = acc[mt][jt][b][r];
    __syncthreads();
  }

  // ---- write out: item = (q = (mt, jt, b), lane) -> 4 consecutive n of one token ----
  // (LINES: item = (q = (mt, j, b), jt, token) as in the column kernel: the 4 consecutive n of a token sit in the lanes token + 16 cq of red[(q * 4 + jt) * 64 ..])
  auto item_coords = [&](const int it, int& m, int& n) {
    const int q = it >> 6, ln = it & 63;
    const int mt = q >> 3, jt = (q >> 1) & 3, b = q & 1;
    const int qd = ln >> 4;
    m = mbase + 16 * mt + (ln & 15);
    if constexpr (LINES) n = strip * 128 + 64 * (jt >> 1) + 16 * qd + 8 * b + 4 * (jt & 1);  // (the q field called jt above is the load index j here; qd = the item's jt)
    else n = strip * 128 + 64 * (qd >> 1) + 16 * jt + 8 * b + 4 * (qd & 1);
  };
  auto item_vals = [&](const int it) -> v4i {
    const int q = it >> 6, ln = it & 63;
    if constexpr (LINES) {
      const int* rp = &red[(q * 4 + (ln >> 4)) * 64 + (ln & 15)];
      return (v4i){rp[0], rp[16], rp[32], rp[48]};
    } else {
      const int* rp = &red[(q * 4) * 64 + ln];
      return (v4i){rp[0], rp[64], rp[128], rp[192]};
    }
  };

  if (ksplit == 1) {
    for (int it = tid; it < NQ * 64; it += WAVES * 64) {
      int m, n;
      item_coords(it, m, n);
      if (m < M && n < N) {
        const v4i v = item_vals(it);
        epilogue_store4(v[0], v[1], v[2], v[3], m, n, N, s1[m], s2, D, acc_out, bias);
      }
    }
    return;
  }

  if (fused == 3) {
    // ---- in-launch split-K, arrival-order slots (round 5; the panel kernel's protocol): slice with ticket t < ksplit - 1 writes its partial
    // tile -- lane-linear, item `it` at 16 * it bytes: full lines -- through to slot t of the tile in C, drains, and counts itself complete;
    // the LAST arrival deposits nothing: its own partial tile stays in LDS, it waits until the others' count is complete (with uneven slices
    // it already is), adds their slots (agent-scope loads) and runs the epilogue.  One launch, no slab of the finisher's own, no re-read of it.
    // Two ticket words per tile (arrivals, completed deposits), zero again on exit. ----
    const size_t tile = (size_t)blockIdx.z * gridDim.x + strip;
    int* tk = tickets + 2 * tile;
    int* flag = &red[NQ * 4 * 64];
    if (tid == 0) *flag = arrival;
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane(*flag);
    constexpr int SLOT_INTS = NQ * 64 * 4;
    int32_t* slots = C + tile * (size_t)(ksplit - 1) * SLOT_INTS;
    if (t < ksplit - 1) {
      for (int it = tid; it < NQ * 64; it += WAVES * 64) {
        const v4i v = item_vals(it);
        int32_t* dst = slots + (size_t)t * SLOT_INTS + (size_t)it * 4;
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // every wave's part of the deposit has reached memory
      if (tid == 0) __hip_atomic_fetch_add(tk + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid == 0) {
      int spin = 0;
      while (__hip_atomic_load(tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ksplit - 1) {
        if (++spin > QQQ_SPIN_LIMIT) __builtin_trap();  // (a depositor that never completes must not end in a silently wrong D)
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    for (int it = tid; it < NQ * 64; it += WAVES * 64) {
      int m, n;
      item_coords(it, m, n);
      v4i sum = item_vals(it);
      const unsigned off = (unsigned)it * 16u;
      for (int p0 = 0; p0 < ksplit - 1; p0 += 4) {  // four slots in flight; surplus loads re-read the last slot, only the add is skipped
        v4i d[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) d[b] = load16_agent(agent_view(slots + (size_t)min(p0 + b, ksplit - 2) * SLOT_INTS), off);
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (p0 + b < ksplit - 1) sum += d[b];
      }
      if (m < M && n < N) epilogue_store4(sum[0], sum[1], sum[2], sum[3], m, n, N, s1[m], s2, D, acc_out, bias);
    }
    if (tid < 2) __hip_atomic_store(tk + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // workspace zero on return
    return;
  }

  // split-K: partial sums -> slab sp of C  (C[(sp*M + m)*N + n])
  for (int it = tid; it < NQ * 64; it += WAVES * 64) {
    int m, n;
    item_coords(it, m, n);
    if (m < M && n < N) {
      v4i v = item_vals(it);
      int32_t* dst = C + ((size_t)sp * M + m) * N + n;
      *reinterpret_cast<v4i*>(dst) = v;
    }
  }
  if (!fused) return;  // a separate reduce launch finishes the job

  // fused == 1: the formally fenced in-launch reduction over the SLABS by the last-arriving workgroup of this (strip, m-block) tile -- plain
  // stores -> agent-scope release -> ticket; the last arriver acquires and folds the slabs.  Kept as the by-the-book variant the
  // tests run next to the slot protocol above (one ticket word per tile).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* flag = &red[NQ * 4 * 64];
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int* tk = tickets + (blockIdx.z * gridDim.x + strip);
    const int t = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == ksplit - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // workspace zero on return
    }
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  for (int it = tid; it < NQ * 64; it += WAVES * 64) {
    int m, n;
    item_coords(it, m, n);
    if (m < M && n < N) {
      v4i sum = {0, 0, 0, 0};
      for (int p = 0; p < ksplit; ++p) sum += *reinterpret_cast<const v4i*>(C + ((size_t)p * M + m) * N + n);
      epilogue_store4(sum[0], sum[1], sum[2], sum[3], m, n, N, s1[m], s2, D, acc_out, bias);
    }
  }
}


#endif  // QQQ_AMD_QQQ_STREAM_HIP_H_
