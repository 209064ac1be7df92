"""Lane-level numpy model of the HIP kernels' data movement (tests only).

It mirrors, formula by formula, the address arithmetic of qqq_amd/csrc/qqq_{stream,column,tiled,panel}.hip.h
(`qqq_stream_kernel`, `qqq_column_kernel`, `qqq_tiled_kernel`, `qqq_panel_kernel`): which bytes each lane loads, how the LDS image is
swizzled, which MFMA operand slot they land in, and where each accumulator register is stored.
The MFMA lane maps assumed here (and checked on the device by tests/test_gpu_probe.py):

  v_mfma_i32_16x16x64_i8 : A lane l -> row i = l & 15, k-group h = l >> 4 (16 bytes = 16 k-slots)
                           B lane l -> col j = l & 15, same h; slot s of A pairs with slot s of B
                           D lane l -> col j = l & 15, rows i = 4*(l >> 4) + r, r = 0..3
  v_mfma_i32_32x32x32_i8 : A lane l -> row i = l & 31, h = l >> 5; B lane l -> col j = l & 31, h
                           D lane l -> col j = l & 31, rows i = (r & 3) + 8*(r >> 2) + 4*(l >> 5)
"""
import numpy as np

MASK = np.uint32(0xF0F0F0F0)


def _bytes_to_i8(words_u32):
    """[..., 4] uint32 -> [..., 16] int8 in memory (little endian) order."""
    return np.ascontiguousarray(words_u32.astype("<u4")).view(np.int8).reshape(*words_u32.shape[:-1], 16)


def _dequant4(q, s_half):
    """dequant_group4 of the kernel: nibbles p0,p4,p1,p5 -> 4 int8 packed in a uint32."""
    from oracle.qqq_ref import dequant_per_group_faithful as dq

    q = np.asarray(q, dtype=np.uint32)
    nib = [(q >> np.uint32(4 * p)) & np.uint32(0xF) for p in (0, 4, 1, 5)]
    out = np.zeros_like(q)
    for byte, u in enumerate(nib):
        w8 = dq(u.astype(np.int8), np.broadcast_to(s_half, u.shape)).view(np.uint8).astype(np.uint32)
        out |= w8 << np.uint32(8 * byte)
    return out


def unpack_pair(q, grouped, s_b0=None, s_b1=None):
    q = np.asarray(q, dtype=np.uint32)
    if grouped:
        return _dequant4(q, s_b0), _dequant4(q >> np.uint32(8), s_b1)
    return q & MASK, (q << np.uint32(4)) & MASK


def mfma_16x16x64(a_ops, b_ops):
    """a_ops, b_ops: [64 lanes, 16] int8 -> D regs [64 lanes, 4] int32."""
    l = np.arange(64)
    A = np.zeros((16, 4, 16), np.int64)
    Bm = np.zeros((16, 4, 16), np.int64)
    A[l & 15, l >> 4] = a_ops
    Bm[l & 15, l >> 4] = b_ops
    Dm = np.einsum("ihs,jhs->ij", A, Bm)
    out = np.zeros((64, 4), np.int64)
    for r in range(4):
        out[:, r] = Dm[4 * (l >> 4) + r, l & 15]
    return out


def mfma_32x32x32(a_ops, b_ops):
    l = np.arange(64)
    A = np.zeros((32, 2, 16), np.int64)
    Bm = np.zeros((32, 2, 16), np.int64)
    A[l & 31, l >> 5] = a_ops
    Bm[l & 31, l >> 5] = b_ops
    Dm = np.einsum("ihs,jhs->ij", A, Bm)
    out = np.zeros((64, 16), np.int64)
    for r in range(16):
        out[:, r] = Dm[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def stream_kernel_model(A, B, s3, M, N, K, MT, WAVES, ksplit, grouped):
    """Returns acc[M,N] int64 as the stream kernel would produce it (all m-blocks, K slices summed)."""
    Bb = np.ascontiguousarray(B).view(np.uint8).reshape(-1)  # bytes
    Ab = np.ascontiguousarray(A).view(np.uint8).reshape(-1)
    s3h = None if not grouped else np.ascontiguousarray(s3).reshape(-1)
    rowbytes = N * 8
    ngroups = N >> 6
    KS = K >> 6
    lane = np.arange(64)
    j = lane & 15
    h = lane >> 4
    g = j >> 3
    c = j & 7
    out = np.zeros((M, N), np.int64)
    strips = (N + 127) // 128
    mblocks = (M + 16 * MT - 1) // (16 * MT)
    for mb in range(mblocks):
        mbase = mb * 16 * MT
        for strip in range(strips):
            ng = np.minimum(strip * 2 + g, ngroups - 1)
            boff = h * rowbytes + ng * 512 + c * 64
            for sp in range(ksplit):
                ks_begin = (KS * sp) // ksplit
                ks_end = (KS * (sp + 1)) // ksplit
                red = np.zeros((MT * 8 * 4, 64), np.int64)
                for wave in range(WAVES):
                    acc = np.zeros((MT, 4, 2, 64, 4), np.int64)
                    for s in range(ks_begin + wave, ks_end, WAVES):
                        p = boff + 4 * s * rowbytes
                        idx = p[:, None] + np.arange(64)[None, :]
                        w = Bb[idx].reshape(64, 4, 4, 4).view(np.uint8)  # [lane, kq, jt, byte]
                        w = (w.astype(np.uint32) << (8 * np.arange(4, dtype=np.uint32))).sum(-1).astype(np.uint32)
                        xs = []
                        for mt in range(MT):
                            row = np.minimum(mbase + 16 * mt + j, M - 1)
                            xo = row * K + 16 * h + 64 * s
                            xs.append(Ab[xo[:, None] + np.arange(16)[None, :]].view(np.int8))
                        if grouped:
                            so = ng * 64 + c * 8 + (s >> 1) * N
                            sc = s3h[so[:, None] + np.arange(8)[None, :]]  # [lane, 8]
                        for jt in range(4):
                            if grouped:
                                w0, w1 = unpack_pair(w[:, :, jt], True, sc[:, 2 * jt][:, None], sc[:, 2 * jt + 1][:, None])
                            else:
                                w0, w1 = unpack_pair(w[:, :, jt], False)
                            a0 = _bytes_to_i8(w0)
                            a1 = _bytes_to_i8(w1)
                            for mt in range(MT):
                                acc[mt, jt, 0] += mfma_16x16x64(a0, xs[mt])
                                acc[mt, jt, 1] += mfma_16x16x64(a1, xs[mt])
                    for mt in range(MT):
                        for jt in range(4):
                            for b in range(2):
                                for r in range(4):
                                    red[((mt * 4 + jt) * 2 + b) * 4 + r] += acc[mt, jt, b, :, r]
                # write-out: item_coords
                for it in range(MT * 8 * 64):
                    q, ln = it >> 6, it & 63
                    mt, jt, b = q >> 3, (q >> 1) & 3, q & 1
                    qd = ln >> 4
                    m = mbase + 16 * mt + (ln & 15)
                    n = strip * 128 + 64 * (qd >> 1) + 16 * jt + 8 * b + 4 * (qd & 1)
                    if m < M and n < N:
                        out[m, n : n + 4] += red[q * 4 : q * 4 + 4, ln]
    return out


def _quad_perm(v, perm):
    """DPP quad_perm: lane l reads lane (l & ~3) + perm[l & 3]."""
    l = np.arange(64)
    return v[(l & ~3) + np.asarray(perm)[l & 3]]


def column_kernel_model(A, B, s3, M, N, K, MT, WAVES, ksplit, grouped):
    """qqq_column_kernel: 16-byte weight loads by lane (h, c', kq), 4x4 register/lane transpose in two
    DPP butterfly stages, MFMA row i = 4*c' + jt, D lane -> (c' = lane >> 4, jt = register)."""
    Bb = np.ascontiguousarray(B).view(np.uint8).reshape(-1)
    Ab = np.ascontiguousarray(A).view(np.uint8).reshape(-1)
    s3h = None if not grouped else np.ascontiguousarray(s3).reshape(-1)
    rowbytes = N * 8
    KS = K >> 6
    lane = np.arange(64)
    h, cq, q4 = lane >> 4, (lane >> 2) & 3, lane & 3
    odd, hi = (lane & 1) != 0, (lane & 2) != 0
    out = np.zeros((M, N), np.int64)
    mblocks = (M + 16 * MT - 1) // (16 * MT)
    for mb in range(mblocks):
        mbase = mb * 16 * MT
        for wg in range(N // 32):
            ng, half = wg >> 1, wg & 1
            bptr = h * rowbytes + ng * 512 + (4 * half + cq) * 64 + q4 * 16
            for sp in range(ksplit):
                ks_begin = (KS * sp) // ksplit
                ks_end = (KS * (sp + 1)) // ksplit
                red = np.zeros((MT * 2 * 4, 64), np.int64)
                for wave in range(WAVES):
                    acc = np.zeros((MT, 2, 64, 4), np.int64)
                    for s in range(ks_begin + wave, ks_end, WAVES):
                        src = bptr + 4 * s * rowbytes
                        w = Bb[src[:, None] + np.arange(16)[None, :]].reshape(64, 4, 4)
                        w = (w.astype(np.uint32) << (8 * np.arange(4, dtype=np.uint32))).sum(-1).astype(np.uint32)  # [lane, e]
                        z = np.zeros_like(w)
                        y = np.zeros_like(w)
                        for e in range(4):
                            t = _quad_perm(w[:, e ^ 1], [1, 0, 3, 2])
                            z[:, e] = np.where(odd == bool(e & 1), w[:, e], t)
                        for e in range(4):
                            t = _quad_perm(z[:, e ^ 2], [2, 3, 0, 1])
                            y[:, e] = np.where(hi == bool(e & 2), z[:, e], t)
                        if grouped:
                            so = ng * 64 + (4 * half + cq) * 8 + 2 * q4 + (s >> 1) * N
                            w0, w1 = unpack_pair(y, True, s3h[so][:, None], s3h[so + 1][:, None])
                        else:
                            w0, w1 = unpack_pair(y, False)
                        a0, a1 = _bytes_to_i8(w0), _bytes_to_i8(w1)
                        for mt in range(MT):
                            row = np.minimum(mbase + 16 * mt + (lane & 15), M - 1)
                            xo = row * K + 16 * h + 64 * s
                            x = Ab[xo[:, None] + np.arange(16)[None, :]].view(np.int8)
                            acc[mt, 0] += mfma_16x16x64(a0, x)
                            acc[mt, 1] += mfma_16x16x64(a1, x)
                    for mt in range(MT):
                        for b in range(2):
                            for r in range(4):
                                red[(mt * 2 + b) * 4 + r] += acc[mt, b, :, r]
                for it in range(MT * 2 * 64):
                    q, jt, tok = it >> 6, (it >> 4) & 3, it & 15
                    mt, b = q >> 1, q & 1
                    m = mbase + 16 * mt + tok
                    n = 64 * ng + 16 * jt + 8 * b + 4 * half
                    if m < M:
                        out[m, n : n + 4] += red[q * 4 + jt, tok + 16 * np.arange(4)]
    return out


def tiled_kernel_model(A, B, s3, M, N, K, BM, MTW, JW, ksplit, grouped, return_tile_order=False, NB=2):
    Bb = np.ascontiguousarray(B).view(np.uint8).reshape(-1)
    Ab = np.ascontiguousarray(A).view(np.uint8).reshape(-1)
    s3h = None if not grouped else np.ascontiguousarray(s3).reshape(-1)
    WM, WN = BM // (32 * MTW), (4 // JW) * (2 // NB)
    NT = WM * WN * 64
    W_BYTES, X_BYTES = 8 * 2048, BM * 128
    W_CHUNKS, X_CHUNKS = W_BYTES // 16, X_BYTES // 16
    rowbytes = N * 8
    ngroups = N >> 6
    NKB = K >> 7
    tiles_m, tiles_n = (M + BM - 1) // BM, (N + 255) // 256
    ntiles = tiles_m * tiles_n
    out = np.zeros((M, N), np.int64)
    seen = []
    for bid in range(ntiles):
        q, rr = ntiles >> 3, ntiles & 7
        xcd, idx = bid & 7, bid >> 3
        lin = (xcd * (q + 1) if xcd < rr else rr * (q + 1) + (xcd - rr) * q) + idx
        PW = 4
        full = (tiles_n // PW) * PW * tiles_m
        if lin < full:
            panel, within = lin // (PW * tiles_m), lin % (PW * tiles_m)
            tile_m, tile_n = within // PW, panel * PW + within % PW
        else:
            rem, pw = lin - full, tiles_n % PW
            tile_m, tile_n = rem // pw, (tiles_n // PW) * PW + rem % pw
        seen.append((tile_m, tile_n))
        m0, ng0 = tile_m * BM, tile_n * 4
        for sp in range(ksplit):
            kb_begin, kb_end = (NKB * sp) // ksplit, (NKB * (sp + 1)) // ksplit
            acc = np.zeros((WM * WN, MTW, JW, NB, 64, 16), np.int64)
            for kb in range(kb_begin, kb_end):
                # ---- build the LDS stage image exactly as the staging threads do ----
                lds = np.zeros(W_BYTES + X_BYTES, np.uint8)
                tid = np.arange(NT)
                for i in range(W_CHUNKS // NT):
                    cw = tid + i * NT
                    ktl, gl, pos = cw >> 7, (cw >> 5) & 3, cw & 31
                    cc, kq = pos >> 2, (pos & 3) ^ gl
                    ngx = np.minimum(ng0 + gl, ngroups - 1)
                    src = kb * 8 * rowbytes + ktl * rowbytes + ngx * 512 + (4 * cc + kq) * 16
                    lds[(cw[:, None] * 16 + np.arange(16)[None, :])] = Bb[src[:, None] + np.arange(16)[None, :]]
                for i in range(X_CHUNKS // NT):
                    cx = tid + i * NT
                    row, pos = cx >> 3, cx & 7
                    chunk = pos ^ ((row >> 1) & 7)
                    grow = np.minimum(m0 + row, M - 1)
                    src = grow * K + kb * 128 + chunk * 16
                    lds[W_BYTES + cx[:, None] * 16 + np.arange(16)[None, :]] = Ab[src[:, None] + np.arange(16)[None, :]]
                # ---- every wave reads its fragments ----
                lane = np.arange(64)
                li, h = lane & 31, lane >> 5
                g, c = li >> 3, li & 7
                for wave in range(WM * WN):
                    wm = wave // WN
                    wn, bsel = (wave % WN) // (2 // NB), (wave % WN) % (2 // NB)
                    half = (wn ^ (c >> 2)) if JW == 2 else (((wn >> 1) ^ (c >> 2)) if JW == 1 else 0 * c)
                    esel = (wn & 1) if JW == 1 else 0
                    jt0 = 0 * c if JW == 4 else 2 * half + esel
                    if grouped:
                        ngx = np.minimum(ng0 + g, ngroups - 1)
                        so = ngx * 64 + c * 8 + 2 * jt0 + kb * N
                        sc = s3h[so[:, None] + np.arange(2 * JW)[None, :]]
                    for t in range(4):
                        wq = np.zeros((64, 4, JW), np.uint32)
                        for kq in range(4):
                            off = h * 2048 + g * 512 + (4 * c + (kq ^ g)) * 16 + (0 if JW == 4 else half * 8) + t * 4096
                            nby = 16 if JW == 4 else 8
                            raw = lds[off[:, None] + np.arange(nby)[None, :]].reshape(64, nby // 4, 4)
                            words = (raw.astype(np.uint32) << (8 * np.arange(4, dtype=np.uint32))).sum(-1)
                            if JW == 1:
                                wq[:, kq, 0] = words[:, esel]
                            else:
                                wq[:, kq, :] = words
                        xops = []
                        for mt in range(MTW):
                            xo = W_BYTES + (wm * MTW * 32 + li) * 128 + (((2 * t + h) ^ ((li >> 1) & 7)) * 16) + mt * 32 * 128
                            xops.append(lds[xo[:, None] + np.arange(16)[None, :]].view(np.int8))
                        for jj in range(JW):
                            if grouped:
                                w0, w1 = unpack_pair(wq[:, :, jj], True, sc[:, 2 * jj][:, None], sc[:, 2 * jj + 1][:, None])
                            else:
                                w0, w1 = unpack_pair(wq[:, :, jj], False)
                            ws = [w0, w1] if NB == 2 else [w1 if bsel else w0]
                            for bi, w in enumerate(ws):
                                a = _bytes_to_i8(w)
                                for mt in range(MTW):
                                    acc[wave, mt, jj, bi] += mfma_32x32x32(a, xops[mt])
            # ---- epilogue mapping ----
            lane = np.arange(64)
            li, h = lane & 31, lane >> 5
            for wave in range(WM * WN):
                wm = wave // WN
                wn, bsel = (wave % WN) // (2 // NB), (wave % WN) % (2 // NB)
                for jj in range(JW):
                    jt = jj + 0 * h if JW == 4 else (2 * (wn ^ h) + jj if JW == 2 else 2 * ((wn >> 1) ^ h) + (wn & 1))
                    for bi in range(NB):
                        b = bi if NB == 2 else bsel
                        for gq in range(4):
                            n = ng0 * 64 + 4 * h + 64 * gq + 16 * jt + 8 * b
                            for mt in range(MTW):
                                m = m0 + (wm * MTW + mt) * 32 + li
                                for ln in range(64):
                                    if n[ln] < N and m[ln] < M:
                                        out[m[ln], n[ln] : n[ln] + 4] += acc[wave, mt, jj, bi, ln, 4 * gq : 4 * gq + 4]
    if return_tile_order:
        return out, seen
    return out


def _quad_transpose4(w):
    """quad_transpose4 of qqq_common.hip.h: y[e](lane q of a quad) = w[q](lane e); w, y: [64 lanes, 4] uint32."""
    lane = np.arange(64)
    odd, hi = (lane & 1) != 0, (lane & 2) != 0
    z = np.zeros_like(w)
    y = np.zeros_like(w)
    for e in range(4):
        z[:, e] = np.where(odd == bool(e & 1), w[:, e], _quad_perm(w[:, e ^ 1], [1, 0, 3, 2]))
    for e in range(4):
        y[:, e] = np.where(hi == bool(e & 2), z[:, e], _quad_perm(z[:, e ^ 2], [2, 3, 0, 1]))
    return y


def panel_stage_image(Ab, M, K, mbase, ROWS, st, k_tail_stage):
    """LDS image of one 128-k activation stage as store_x writes it: chunk (row, 16-byte piece pos) at
    row*128 + ((pos ^ ((row >> 1) & 7)) << 4); the upper half of a trailing 64-k stage re-fetches the lower half."""
    img = np.zeros(ROWS * 128, np.uint8)
    for row in range(ROWS):
        grow = min(mbase + row, M - 1)
        for pos in range(8):
            src = grow * K + st * 128 + pos * 16 - (64 if (k_tail_stage and pos >= 4) else 0)
            dst = row * 128 + ((pos ^ ((row >> 1) & 7)) << 4)
            img[dst : dst + 16] = Ab[src : src + 16]
    return img


def panel_fragment_addr(mt, tk):
    """byte address inside a stage image of lane l's ds_read_b128 for m-tile mt, 64-k step tk of the stage"""
    lane = np.arange(64)
    h = lane >> 4
    return (lane & 15) * 128 + mt * 2048 + (((4 * tk + h) ^ (((lane & 15) >> 1) & 7)) << 4)


def panel_slot_index(MT, WN, KG, HW):
    """int32 index inside a split-K slot of every (wave column set wn, m-tile, operand q, lane, register r) a finishing wave
    deposits: must tile the ROWS x BN ints of the slot exactly once."""
    NQ = 2 * HW
    split = KG == 2 and MT >= 2
    MTO = MT // 2 if split else MT
    idx = []
    for kg in range(KG if split else 1):
        mb = kg * MTO if split else 0
        for wn in range(WN):
            for j in range(MTO):
                for q in range(NQ):
                    base = ((wn * MT + mb) * NQ + (j * NQ + q)) * 256
                    idx.append(base + np.arange(64)[:, None] * 4 + np.arange(4)[None, :])
    return np.concatenate([i.reshape(-1) for i in idx])


def panel_kernel_model(A, B, s3, M, N, K, MT, WN, KG, HW, ksplit, grouped):
    """qqq_panel_kernel: a workgroup = 16*MT tokens x BN = 32*WN*HW columns x one K slice; wave (wn, kg) owns HW 32-column
    sets of a 64-column group and (KG == 2) the 64-k half kg of every 128-k stage; weights by 16-byte loads + quad transpose,
    activations through the swizzled LDS stage image; D lane -> (token j = l & 15, c' = l >> 4, jt = register)."""
    Bb = np.ascontiguousarray(B).view(np.uint8).reshape(-1)
    Ab = np.ascontiguousarray(A).view(np.uint8).reshape(-1)
    s3h = None if not grouped else np.ascontiguousarray(s3).reshape(-1)
    rowbytes = N * 8
    BN, ROWS = 32 * WN * HW, 16 * MT
    KS = K >> 6
    NST = (KS + 1) >> 1
    k_tail = (KS & 1) != 0
    SPW = 2 // KG
    ngroups = N >> 6
    lane = np.arange(64)
    h, cq, q4 = lane >> 4, (lane >> 2) & 3, lane & 3
    out = np.zeros((M, N), np.int64)
    for mblk in range((M + ROWS - 1) // ROWS):
        mbase = mblk * ROWS
        for strip in range((N + BN - 1) // BN):
            tile = np.zeros((ROWS, BN), np.int64)  # what the epilogue image holds after every slice is folded
            for sp in range(ksplit):
                st_begin, st_end = (NST * sp) // ksplit, (NST * (sp + 1)) // ksplit
                for wave in range(WN * KG):
                    wn, kg = wave % WN, wave // WN
                    gl = (wn * HW) >> 1
                    ng = min(strip * (BN // 64) + gl, ngroups - 1)
                    half = 0 if HW == 2 else (wn & 1)
                    wptr = h * rowbytes + ng * 512 + (4 * half + cq) * 64 + q4 * 16
                    sptr = ng * 64 + (4 * half + cq) * 8 + 2 * q4
                    acc = np.zeros((MT, 2 * HW, 64, 4), np.int64)
                    for st in range(st_begin, st_end):
                        img = panel_stage_image(Ab, M, K, mbase, ROWS, st, k_tail and st == NST - 1)
                        for t in range(SPW):
                            tk = kg if KG == 2 else t
                            s = 2 * st + tk
                            valid = s < KS
                            sl = min(s, KS - 1)
                            ops = []
                            for hf in range(HW):
                                src = wptr + 4 * sl * rowbytes + 256 * hf
                                w = Bb[src[:, None] + np.arange(16)[None, :]].reshape(64, 4, 4)
                                w = (w.astype(np.uint32) << (8 * np.arange(4, dtype=np.uint32))).sum(-1).astype(np.uint32)
                                y = _quad_transpose4(w)
                                if grouped:
                                    so = sptr + st * N + 32 * hf
                                    sc0, sc1 = s3h[so][:, None], s3h[so + 1][:, None]
                                    if not valid:  # scale 0 re-quantises every nibble to 0
                                        sc0, sc1 = np.zeros_like(sc0), np.zeros_like(sc1)
                                    w0, w1 = unpack_pair(y, True, sc0, sc1)
                                else:
                                    nm = MASK if valid else np.uint32(0)
                                    w0, w1 = y & nm, (y << np.uint32(4)) & nm
                                ops += [_bytes_to_i8(w0), _bytes_to_i8(w1)]  # q = 2*hf + b
                            for mt in range(MT):
                                ad = panel_fragment_addr(mt, tk)
                                x = img[ad[:, None] + np.arange(16)[None, :]].view(np.int8)
                                for q in range(2 * HW):
                                    acc[mt, q] += mfma_16x16x64(ops[q], x)
                    # epilogue image: token row 16*mt + j, column 64*gl + 16*r + 8*b + 4*(half + hf) + c'
                    j, cp = lane & 15, lane >> 4
                    for mt in range(MT):
                        for q in range(2 * HW):
                            for r in range(4):
                                col = 64 * gl + 16 * r + 8 * (q & 1) + 4 * (half + (q >> 1)) + cp
                                np.add.at(tile, (16 * mt + j, col), acc[mt, q, :, r])
            for row in range(ROWS):
                m = mbase + row
                if m < M:
                    n0 = strip * BN
                    w = min(BN, N - n0)
                    if w > 0:
                        out[m, n0 : n0 + w] += tile[row, :w]
    return out


def wide_tile_order(tiles_m, tiles_n, ksplit, PW):
    """(tile_m, tile_n, slice) of every block id of the wide kernel's 1-D grid: block b runs on XCD b % 8; an XCD walks a
    contiguous run of the panel-major sequence (panels of PW strips x all m-tiles), the slices of a tile take consecutive
    positions of the run (qqq_wide.hip.h: "XCD-aware tile order")."""
    ntot = tiles_m * tiles_n * ksplit
    q, rr = ntot >> 3, ntot & 7
    out = []
    for bid in range(ntot):
        xcd, idx = bid & 7, bid >> 3
        lin2 = (xcd * (q + 1) if xcd < rr else rr * (q + 1) + (xcd - rr) * q) + idx
        lin, sp = lin2 // ksplit, lin2 % ksplit
        full = (tiles_n // PW) * PW * tiles_m
        if lin < full:
            panel, within = lin // (PW * tiles_m), lin % (PW * tiles_m)
            tm, tn = within // PW, panel * PW + within % PW
        else:
            rem, pw = lin - full, tiles_n % PW
            tm, tn = rem // pw, (tiles_n // PW) * PW + rem % pw
        out.append((tm, tn, sp, lin))
    return out


def wide_kernel_model(A, B, s3, M, N, K, MT, ksplit, grouped, PW=8, BN=256):
    """qqq_wide_kernel: a workgroup = 16*MT tokens x 256 columns x one K slice, four waves; wave wn owns the 64-column group
    tile_n*4 + wn (both halves) and all m-tiles.  Weights: buffer loads at lane offset h*rowbytes + cq*64 + q4*16 (+256 per
    half) over the descriptor base B + ng*512, scalar offset 4*(2*st0 + s)*rowbytes, quad transpose; activations by LDS-DMA: instruction
    q of wave wn fills the lane-linear 1 KiB at q*4096 + wn*1024 of the stage image (lane l -> byte 16 l: row (tid >> 3) + 32 q,
    slot tid & 7) and FETCHES piece slot ^ ((row >> 1) & 7) of that row (the swizzle is on the source side), scalar
    offset (st0 + st)*128 over the base A + mbase*K; fragments as in the panel kernel; epilogue / split-K slot image
    row-major [ROWS][BN] with column 64*wn + 16*r + 8*b + 4*hf + c'.  BN = 128 (MT = 16 only): a wave owns ONE 32-column half
    (cb & 1) of the 64-column group cb >> 1, cb = tile_n*4 + wn: lane offset + 256*half, one weight load per step, image column
    64*(wn >> 1) + 16*r + 8*b + 4*(wn & 1) + c'.  Returns (acc [M,N], the list of slot images)."""
    assert K % 128 == 0
    Bb = np.ascontiguousarray(B).view(np.uint8).reshape(-1)
    Ab = np.ascontiguousarray(A).view(np.uint8).reshape(-1)
    s3h = None if not grouped else np.ascontiguousarray(s3).reshape(-1)
    rowbytes = N * 8
    ROWS = 16 * MT
    HW = BN // 128
    assert BN == 256 or (BN == 128 and MT == 16)
    XPT = ROWS * 128 // 16 // 256
    ngroups = N >> 6
    tiles_m, tiles_n = -(-M // ROWS), -(-N // BN)
    lane = np.arange(64)
    h, cq, q4 = lane >> 4, (lane >> 2) & 3, lane & 3
    tid = np.arange(256)
    xr0, xslot = tid >> 3, tid & 7
    out = np.zeros((M, N), np.int64)
    seen = set()
    partial = {}
    for (tile_m, tile_n, sp, lin) in wide_tile_order(tiles_m, tiles_n, ksplit, PW):
        assert (tile_m, tile_n, sp) not in seen and tile_m < tiles_m and tile_n < tiles_n
        seen.add((tile_m, tile_n, sp))
        mbase = tile_m * ROWS
        st0 = ((K >> 7) * sp) // ksplit
        nst = ((K >> 7) * (sp + 1)) // ksplit - st0
        image = np.zeros((ROWS, BN), np.int64)  # this slice's partial tile as the LDS transposition lays it out
        # activation stage images, built chunk by chunk as the threads store them
        imgs = []
        for st in range(nst):
            img = np.zeros(ROWS * 128, np.uint8)
            for q in range(XPT):
                row = np.minimum(xr0 + 32 * q, M - 1 - mbase)          # rows past M: the last row again
                voff = row * K + (xslot ^ ((xr0 >> 1) & 7)) * 16        # 32-bit lane offset over the base A + mbase*K
                src = mbase * K + voff + (st0 + st) * 128
                dst = q * 4096 + tid * 16                                # M0 = q*4096 + wn*1024 (+ buffer), + 16 * lane
                for t_ in range(256):
                    img[dst[t_] : dst[t_] + 16] = Ab[src[t_] : src[t_] + 16]
            imgs.append(img)
        for wn in range(4):
            cb = tile_n * 4 + wn
            ng = min(cb if HW == 2 else cb >> 1, ngroups - 1)
            whalf = 0 if HW == 2 else (cb & 1)
            woff = h * rowbytes + cq * 64 + q4 * 16 + 256 * whalf
            acc = np.zeros((MT, 2 * HW, 64, 4), np.int64)
            for st in range(nst):
                for t in range(2):
                    s = 2 * st + t
                    ops = []
                    for hf in range(HW):
                        src = ng * 512 + woff + 256 * hf + 4 * (2 * st0 + s) * rowbytes
                        w = Bb[src[:, None] + np.arange(16)[None, :]].reshape(64, 4, 4)
                        w = (w.astype(np.uint32) << (8 * np.arange(4, dtype=np.uint32))).sum(-1).astype(np.uint32)
                        y = _quad_transpose4(w)
                        if grouped:
                            so = ng * 64 + cq * 8 + 2 * q4 + 32 * (hf + whalf) + (st0 + st) * N   # halves: lane offset + 64 B per half + stage
                            w0, w1 = unpack_pair(y, True, s3h[so][:, None], s3h[so + 1][:, None])
                        else:
                            w0, w1 = y & MASK, (y << np.uint32(4)) & MASK
                        ops += [_bytes_to_i8(w0), _bytes_to_i8(w1)]  # q = 2*hf + b
                    for mt in range(MT):
                        ad = panel_fragment_addr(mt, t)
                        x = imgs[st][ad[:, None] + np.arange(16)[None, :]].view(np.int8)
                        for q in range(2 * HW):
                            acc[mt, q] += mfma_16x16x64(ops[q], x)
            j, cp = lane & 15, lane >> 4
            for mt in range(MT):
                for q in range(2 * HW):
                    for r in range(4):
                        col = (64 * wn + 4 * (q >> 1) if HW == 2 else 64 * (wn >> 1) + 4 * (wn & 1)) + 16 * r + 8 * (q & 1) + cp
                        np.add.at(image, (16 * mt + j, col), acc[mt, q, :, r])
        partial.setdefault((tile_m, tile_n), []).append(image)
    assert len(seen) == tiles_m * tiles_n * ksplit
    slots = []
    for (tile_m, tile_n), parts in partial.items():
        tile = sum(parts)                       # arrival order is irrelevant: integer adds
        slots += parts[:-1]                     # every slice but the last arrival deposits its image, row-major
        for row in range(ROWS):
            m = tile_m * ROWS + row
            if m < M:
                n0 = tile_n * BN
                w = min(BN, N - n0)
                out[m, n0 : n0 + w] += tile[row, :w]
    return out, slots
