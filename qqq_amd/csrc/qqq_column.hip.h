// qqq_column.hip.h -- "column" kernel (decode, m <= 16: 32 columns x all of K per workgroup, no split-K)
// Part of the single translation unit qqq_w4a8.hip (see its header comment for the design).
#ifndef QQQ_AMD_QQQ_COLUMN_HIP_H_
#define QQQ_AMD_QQQ_COLUMN_HIP_H_

// ------------------------------------------------------------------------------------------
// "column" kernel: decode (m <= 16).  One workgroup = 32 weight columns over the WHOLE K range, so the
// N/32 workgroups of a wide layer fill the chip without split-K: no slabs, no reduce launch (worth ~4 us of
// a ~21 us call at N=8192, K=21760).
// ------------------------------------------------------------------------------------------
//
// grid = (N/32, ksplit, ceil(M / (16*MT)));  block = WAVES * 64.  Workgroup x -> column group ng = x >> 1,
// half = x & 1: the 32 columns n = 64*ng + 16*jt + 8*b + 4*half + c', c' in [0,4) -- chunks c = 4*half + c'
// of the packed layout, 256 contiguous bytes of every 16-k row of B.  The waves split the 64-k steps of the
// K range round-robin and are summed through LDS at the end (as in the stream kernel).
// Weights and activations come in with plain 16-byte loads and the packed words are re-distributed between
// lanes in registers:
//     load lane L = 16*h + 4*c' + kq holds the 4 words (jt = 0..3) of piece (k-tile 4*s + h, chunk c', kq);
//     MFMA lane l = 16*h + 4*c' + jt needs the word jt of the pieces kq = 0..3 of the same (h, c'),
// i.e. a 4x4 transpose between the 4 registers and the 4 lanes of every quad: two butterfly stages of DPP
// quad_perm selects (quad_transpose4).  The MFMA row of lane l is then i = 4*c' + jt (any bijection onto the 16 rows will do);
// D lane l holds rows 4*(l >> 4) + r, i.e. c' = l >> 4 (of the OUTPUT lane) and jt = r.
// Each workgroup re-reads all m x K activation bytes (from L2), which at m = 16 equals its weight bytes:
// the kernel pays off for m <= 8 everywhere and up to m = 16 while m*K stays small (host heuristic).
// (An LDS-DMA ring variant -- global_load_lds into wave-private slots, no VGPR staging -- measured the same
// at m = 1 and slower at m = 16: the DMA path moves ~34 B/clk/CU against 64 B/clk for plain loads.)
template <int MT, bool GROUPED, int WAVES, int PF>
__global__ __launch_bounds__(WAVES * 64) void qqq_column_kernel(
    const int8_t* __restrict__ A, const unsigned char* __restrict__ B, int32_t* __restrict__ C,
    _Float16* __restrict__ D, const float* __restrict__ s1, const float* __restrict__ s2,
    const _Float16* __restrict__ s3, int32_t* __restrict__ acc_out, const _Float16* __restrict__ bias,
    const int M, const int N, const int K, const int ksplit) {
  constexpr int NQ = MT * 2;
  __shared__ int red[NQ * 4 * 64];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ng = blockIdx.x >> 1, half = blockIdx.x & 1;
  const int sp = blockIdx.y;
  const int mbase = blockIdx.z * (16 * MT);
  if constexpr (WAVES > 1) {
    for (int i = tid; i < NQ * 4 * 64; i += WAVES * 64) red[i] = 0;
    __syncthreads();
  }
  const size_t rowbytes = (size_t)N * 8;
  const int h = lane >> 4, cq = (lane >> 2) & 3, q4 = lane & 3;  // q4: kq as a load lane, jt as an MFMA lane
  const unsigned char* bptr = B + (size_t)h * rowbytes + (size_t)ng * 512 + (4 * half + cq) * 64 + q4 * 16;
  const int8_t* xptr[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = mbase + 16 * mt + (lane & 15);
    if (row >= M) row = M - 1;
    xptr[mt] = A + (size_t)row * K + 16 * h;
  }
  const _Float16* sptr = GROUPED ? (s3 + (size_t)ng * 64 + (4 * half + cq) * 8 + 2 * q4) : nullptr;

  const int KS = K >> 6;
  const int ks_begin = (int)(((long long)KS * sp) / ksplit);
  const int ks_end = (int)(((long long)KS * (sp + 1)) / ksplit);

  v4i acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt][0] = acc[mt][1] = (v4i){0, 0, 0, 0};

  struct Step {
    v4u w;
    v4i x[MT];
    h2 sc;
  };
  auto load_step = [&](const int s, Step& r) {
    if constexpr ((QQQ_W_NT & 1) != 0) r.w = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(bptr + (size_t)(4 * s) * rowbytes));
    else r.w = *reinterpret_cast<const v4u*>(bptr + (size_t)(4 * s) * rowbytes);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) r.x[mt] = *reinterpret_cast<const v4i*>(xptr[mt] + 64 * s);
    if constexpr (GROUPED) r.sc = *reinterpret_cast<const h2*>(sptr + (size_t)(s >> 1) * N);
  };
  auto compute_step = [&](const Step& r) {
    // 4x4 transpose over (register e, quad lane q): y[e](lane q) = w[q](lane e)   (qqq_common.hip.h: 8 VALU)
    unsigned y[4];
    quad_transpose4(r.w, y);
    h2 sb0 = {(_Float16)0, (_Float16)0}, sb1 = sb0;
    if constexpr (GROUPED) {
      sb0 = (h2){r.sc[0], r.sc[0]};
      sb1 = (h2){r.sc[1], r.sc[1]};
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
      acc[mt][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, r.x[mt], acc[mt][0], 0, 0, 0);
      acc[mt][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, r.x[mt], acc[mt][1], 0, 0, 0);
    }
  };

  // software pipeline as in the stream kernel: PF steps in flight per wave, branch-free steady state
  Step ring[PF];
  int s = ks_begin + wave;
  if (s + (2 * PF - 1) * WAVES < ks_end) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      load_step(s + p * WAVES, ring[p]);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (; s + (2 * PF - 1) * WAVES < ks_end; s += PF * WAVES) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        compute_step(ring[p]);
        __builtin_amdgcn_sched_barrier(0);
        load_step(s + (p + PF) * WAVES, ring[p]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (s + p * WAVES < ks_end) load_step(s + p * WAVES, ring[p]);
  }
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

  if constexpr (WAVES > 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __hip_atomic_fetch_add(&red[((mt * 2 + b) * 4 + r) * 64 + lane], acc[mt][b][r], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((mt * 2 + b) * 4 + r) * 64 + lane] = acc[mt][b][r];
  }
  __syncthreads();

  // write out: MFMA row i = 4*c' + jt, D lane ln holds rows 4*(ln >> 4) + r  ->  c' = ln >> 4, jt = r.
  // item = (q = (mt, b), jt, token): gathers c' = 0..3 (4 consecutive n) from the 4 lanes token + 16*c'.
  for (int it = tid; it < NQ * 64; it += WAVES * 64) {
    const int q = it >> 6, jt = (it >> 4) & 3, tok = it & 15;
    const int mt = q >> 1, b = q & 1;
    const int m = mbase + 16 * mt + tok;
    const int n = 64 * ng + 16 * jt + 8 * b + 4 * half;
    if (m < M) {
      const int* rp = &red[(q * 4 + jt) * 64 + tok];
      if (ksplit == 1) {
        epilogue_store4(rp[0], rp[16], rp[32], rp[48], m, n, N, s1[m], s2, D, acc_out, bias);
      } else {
        v4i v = {rp[0], rp[16], rp[32], rp[48]};
        *reinterpret_cast<v4i*>(C + ((size_t)sp * M + m) * N + n) = v;
      }
    }
  }
}


#endif  // QQQ_AMD_QQQ_COLUMN_HIP_H_
