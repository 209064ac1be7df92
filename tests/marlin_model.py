"""Thread-level functional model of the REFERENCE kernel's own reads (test infrastructure, CPU only).

`Marlin<>` (/root/reference/csrc/qqq_gemm.cu:240-820) is followed thread by thread -- for ONE threadblock (`marlin_one_block`: gridDim.x
== 1, one stripe that walks every column slice, slice_count == 1, no global reduce) and, since round 4, for a GRID of threadblocks
(`marlin_grid`: the striped partition `:261-338`, stripes that start in the middle of a column slice, `global_reduce` through the int32
buffer C in lock order `:606-676, :792-812`; all four tile configurations incl. the two 128-thread ones, `:935-945`): every index expression below is the kernel's
own, with its line cited -- which 16-byte piece of A / B / s2 / s3 a thread copies to which shared-memory slot
(:351-407, :468-497), which slot it reads back (`ldmatrix`, `frag_b_quant`, `frag_s3`; :510-522), how the registers are
interpreted by `mma.m16n8k16` (PTX fragment layouts for .s8), how the four k-slices of a tile meet (:555-596) and how the
accumulators travel through shared memory to D (:680-726, :766-792).  What is NOT modelled is time: the 4-stage cp.async ring
only decides WHEN a tile is in shared memory, not which bytes (every read of stage p sees tile p).

The per-group re-quantiser's ARITHMETIC (dequant_per_group, :167-210) is the oracle's bit-faithful restatement
(oracle.qqq_ref.dequant_per_group_faithful, checked exhaustively elsewhere); this model pins which NIBBLES and which SCALE the
kernel hands to it.  tests/test_marlin_model_cpu.py holds the model to the oracle's int32 accumulators and fp16 outputs on
operands packed by the reference's pack(): the oracle's closed-form layout is thereby checked against the way the reference
kernel itself reads the packed tensors, not only against pack()'s output.
"""
import numpy as np

from oracle import qqq_ref as R


def ceildiv(a, b):
    return -(-a // b)


def marlin_one_block(A, B, s1, s2, s3=None, thread_m_blocks=1, thread_n_blocks=8, thread_k_blocks=8, threads=256, grid=1):
    """(D fp16 [m,n], acc int32 [m,n]) as ONE threadblock of Marlin<threads, thread_m_blocks, thread_n_blocks,
    thread_k_blocks, 4, group_blocks> computes them.  A int8 [m,k]; B int32 [k/16, 2n]; s1 f32 [m]; s2 f32 [n] (stored
    order); s3 fp16 [k/128, n] (stored order) or None."""
    prob_m, prob_k = A.shape
    prob_n = B.shape[1] // 2
    grouped = s3 is not None and np.asarray(s3).size > 0
    group_blocks = 8 if grouped else -1
    assert prob_m <= 16 * thread_m_blocks, "parallel > 1 (qqq_gemm.cu:272-276) is a pointer offset, not modelled"
    assert prob_k % (16 * thread_k_blocks) == 0 and prob_n % (16 * thread_n_blocks) == 0
    tid = np.arange(threads)
    # global memory in the kernel's units: int4 = 16 bytes
    A4 = np.ascontiguousarray(A, np.int8).reshape(-1, 16)
    B4 = np.ascontiguousarray(B).view(np.uint32).reshape(-1, 4)
    S2 = np.ascontiguousarray(s2, np.float32).reshape(-1, 4)
    S3 = np.ascontiguousarray(s3).view(np.uint16).reshape(-1, 8) if grouped else None
    s1 = np.ascontiguousarray(s1, np.float32).reshape(-1)
    D4 = np.zeros((prob_m * prob_n // 8, 8), np.float16)
    ACC4 = np.zeros((prob_m * prob_n // 8, 8), np.int32)

    k_tiles = prob_k // 16 // thread_k_blocks                    # :278
    n_tiles = prob_n // 16 // thread_n_blocks                    # :279
    # ---- strides and per-thread indices (:338-407) ----
    a_gl_stride = prob_k // 16
    a_sh_stride = 16 * thread_k_blocks // 16
    a_gl_rd_delta_o = 16 * thread_k_blocks // 16
    a_gl_rd_delta_i = a_gl_stride * (threads // a_gl_rd_delta_o)
    a_sh_wr_delta = a_sh_stride * (threads // a_gl_rd_delta_o)
    a_sh_rd_delta_o = 1 * ((threads // 32) // (thread_n_blocks // 4))
    a_sh_rd_delta_i = a_sh_stride * 16
    a_sh_stage = a_sh_stride * (16 * thread_m_blocks)
    a_sh_wr_iters = ceildiv(a_sh_stage, a_sh_wr_delta)
    b_gl_stride = 16 * prob_n // 32
    b_sh_stride = 32 * thread_n_blocks // 4
    b_gl_rd_delta_o = b_gl_stride * thread_k_blocks
    b_gl_rd_delta_i = b_gl_stride * (threads // b_sh_stride)
    b_sh_wr_delta = threads
    b_sh_rd_delta = threads
    b_sh_stage = b_sh_stride * thread_k_blocks
    b_sh_wr_iters = b_sh_stage // b_sh_wr_delta
    s2_sh_stride = 16 * thread_n_blocks // 4
    s3_gl_stride = prob_n // 8
    s3_sh_stride = 16 * thread_n_blocks // 8

    a_sh_wr = a_sh_stride * (tid // a_gl_rd_delta_o) + (tid % a_gl_rd_delta_o)                          # :372
    a_sh_rd = a_sh_stride * ((tid % 32) % 16) + 1 * ((tid // 32) // (thread_n_blocks // 4))             # :375-376
    b_sh_wr = tid
    b_sh_rd = tid
    s1_sh_wr = (tid // 16) * 16 + (tid % 8) * 2 + (tid % 16) // 8                                        # :389
    s1_sh_rd = (tid % 32) // 4
    s2_sh_rd = 16 * ((tid // 32) % (thread_n_blocks // 4)) + 2 * ((tid % 32) % 4)                       # :395
    s3_sh_rd = 8 * ((tid // 32) % (thread_n_blocks // 4)) + (tid % 32) // 4                             # :403

    def transform_a(i):                                                                                 # :420-423
        row = i // a_gl_rd_delta_o
        return (a_gl_rd_delta_o * row + (i % a_gl_rd_delta_o)) ^ row   # C precedence: `+` binds tighter than `^`

    a_sh_wr_pred = [a_sh_wr_delta * i + a_sh_wr < a_sh_stride * prob_m for i in range(a_sh_wr_iters)]   # :413-415
    a_sh_wr_trans = [transform_a(a_sh_wr_delta * i + a_sh_wr) for i in range(a_sh_wr_iters)]           # :428-430
    a_sh_rd_trans = [[transform_a(a_sh_rd_delta_o * i + a_sh_rd_delta_i * j + a_sh_rd) for j in range(thread_m_blocks)]
                     for i in range(b_sh_wr_iters)]                                                     # :431-437
    lane, warp = tid % 32, tid // 32
    nwarps = threads // 32

    def slice_pass(slice_col, slice_row, slice_iters):
        """the main loop over `slice_iters` k-tiles of column slice `slice_col` starting at tile row `slice_row` (:729-760), then
        thread_block_reduce (:555-596): the fragments of the threads tid < b_sh_stride, [b_sh_stride, thread_m_blocks, 4, 2, 4]"""
        a_gl_rd = a_gl_stride * (tid // a_gl_rd_delta_o) + (tid % a_gl_rd_delta_o)                      # :369; the slice_row term (:370) is `kt` below
        b_gl_rd = b_gl_stride * (tid // b_sh_stride) + (tid % b_sh_stride) + b_sh_stride * slice_col    # :378-379; :380 likewise
        frag_c = np.zeros((threads, thread_m_blocks, 4, 2, 4), np.int64)
        for kt in range(slice_row, slice_row + slice_iters):
            # ---- fetch_to_shared (:468-497): tile kt of this slice into its pipeline stage ----
            sh_a = np.zeros((a_sh_stage, 16), np.int8)
            for i in range(a_sh_wr_iters):
                p = a_sh_wr_pred[i]
                sh_a[a_sh_wr_trans[i][p]] = A4[(a_gl_rd_delta_i * i + a_gl_rd + a_gl_rd_delta_o * kt)[p]]
            sh_b = np.zeros((b_sh_stage, 4), np.uint32)
            for i in range(b_sh_wr_iters):
                sh_b[b_sh_wr_delta * i + b_sh_wr] = B4[b_gl_rd_delta_i * i + b_gl_rd + b_gl_rd_delta_o * kt]
            if grouped:  # group_blocks == thread_k_blocks: every tile starts a new group (:486-492)
                sh_s3 = np.zeros((s3_sh_stride, 8), np.uint16)
                pr = tid < s3_sh_stride
                s3_gl_rd = s3_gl_stride * ((thread_k_blocks * kt) // group_blocks) + s3_sh_stride * slice_col + tid   # :400
                sh_s3[tid[pr]] = S3[s3_gl_rd[pr]]
            for k in range(b_sh_wr_iters):
                # ---- fetch_to_registers (:510-522) ----
                frag_s3 = sh_s3[s3_sh_rd] if grouped else None        # [threads, 8 halfs] = FragS_GROUP[4] (half2 each)
                fa = np.zeros((threads, thread_m_blocks, 2, 4), np.int8)  # FragA: two 32-bit registers = 2 x 4 int8
                for i in range(thread_m_blocks):
                    addr = a_sh_rd_trans[k % b_sh_wr_iters][i]  # the row address each thread hands to ldmatrix
                    # ldmatrix .m8n8.x2 .b16: matrix q's row r comes from thread 8q + r's address; thread t receives the
                    # 32-bit element t % 4 of row t / 4 of each matrix
                    for q in range(2):
                        src = sh_a[addr[32 * warp + 8 * q + lane // 4]]             # [threads, 16 bytes]
                        fa[:, i, q, :] = np.take_along_axis(src, (4 * (lane % 4))[:, None] + np.arange(4)[None, :], axis=1)
                fbq = sh_b[b_sh_rd_delta * (k % b_sh_wr_iters) + b_sh_rd]  # [threads, 4] packed words
                # ---- matmul (:527-550) ----
                for j in range(4):
                    q = fbq[:, j]
                    fb = np.zeros((threads, 2, 4), np.int8)  # frag_b0 / frag_b1: 4 int8 along k
                    if grouped:
                        for i2 in range(2):
                            qq = q if i2 == 0 else (q >> np.uint32(8))                                 # :536
                            # t0 = nibbles at bits 0-3 / 16-19, t1 = bits 4-7 / 20-23 (:173-174); prmt 0x6420 picks the low
                            # bytes of t0.lo, t0.hi, t1.lo, t1.hi (:205): k order within the fragment = (p0, p4, p1, p5)
                            nib = np.stack([qq & 0xF, (qq >> 16) & 0xF, (qq >> 4) & 0xF, (qq >> 20) & 0xF], axis=1).astype(np.int8)
                            sc = frag_s3[:, 2 * j + i2].view(np.float16)                                # frag_s3[k % 2][j], half i (:193)
                            fb[:, i2, :] = R.dequant_per_group_faithful(nib, np.repeat(sc[:, None], 4, axis=1))
                    else:
                        for i2, qq in enumerate((q, q << np.uint32(4))):                                # :540-542
                            w = (qq & np.uint32(0xF0F0F0F0)).astype(np.uint32)                          # dequant_per_channel (:146-151)
                            fb[:, i2, :] = w[:, None].view(np.uint8).reshape(threads, 4).view(np.int8)
                    for i in range(thread_m_blocks):
                        for i2 in range(2):
                            frag_c[:, i, j, i2, :] += _mma_m16n8k16(fa[:, i], fb[:, i2], nwarps)
        # ---- thread_block_reduce (:555-596): the threads/b_sh_stride k-slices of the tile meet in shared memory; thread t of the
        # first slice ends up with the sum over the threads t + b_sh_stride * r (same fragment slots: red_sh_rd keeps t % b_sh_stride)
        red = threads // b_sh_stride
        return frag_c.reshape(red, b_sh_stride, thread_m_blocks, 4, 2, 4).sum(axis=0)  # valid for tid < b_sh_stride

    def write_out(slice_col, fc):
        # ---- scales for the write-out (:766-792) ----
        sh_s1 = {}
        for t in range(min(threads, prob_m)):                          # s1_sh_wr_pred = tid < prob_m; s1_gl_rd = tid
            sh_s1[int(s1_sh_wr[t])] = s1[t]
        sh_s2 = S2[s2_sh_stride * slice_col + np.arange(s2_sh_stride)]  # s2_gl_rd (:393), pred tid < s2_sh_stride
        # ---- write_result (:680-726) ----
        d_gl_stride = prob_n // 8
        d_sh_stride = 2 * thread_n_blocks + 1
        sh_h = np.zeros((d_sh_stride * 16 * thread_m_blocks * 4, 2), np.float16)  # shared memory as half2
        sh_i = np.zeros((d_sh_stride * 16 * thread_m_blocks * 4, 2), np.int32)    # the same slots, raw accumulators
        for t in range(32 * (thread_n_blocks // 4)):
            ln = t % 32
            d_sh_wr = (4 * d_sh_stride) * (ln // 4) + ln % 4 + 32 * (t // 32)
            f2 = np.concatenate([sh_s2[s2_sh_rd[t] + o] for o in (0, 1, 8, 9)])  # frag_s2[2][4] as 16 floats (:782-785)
            for i in range(thread_m_blocks):
                fs1 = (sh_s1.get(16 * i + 2 * int(s1_sh_rd[t]), np.float32(0)), sh_s1.get(16 * i + 2 * int(s1_sh_rd[t]) + 1, np.float32(0)))
                for j in range(4):
                    wr = d_sh_wr + 8 * j
                    c = fc[t, i, j]
                    for (off, half, pair, si) in ((0, 0, 0, 0), ((4 * d_sh_stride) * 8, 0, 1, 1), (4, 1, 0, 0), ((4 * d_sh_stride) * 8 + 4, 1, 1, 1)):
                        ws = f2[((j // 2) * 4 + 2 * (j % 2) + half) * 2: ((j // 2) * 4 + 2 * (j % 2) + half) * 2 + 2]
                        c0, c1 = np.int32(c[half, 2 * pair]), np.int32(c[half, 2 * pair + 1])
                        a_s = np.float32(fs1[si])
                        sh_h[wr + off, 0] = np.float16(np.float32(np.float32(c0) * ws[0]) * a_s)   # :697-698: two fp32 multiplies, RN to fp16
                        sh_h[wr + off, 1] = np.float16(np.float32(np.float32(c1) * ws[1]) * a_s)
                        sh_i[wr + off] = (c0, c1)
                d_sh_wr += 16 * (4 * d_sh_stride)
        d_gl_wr = d_gl_stride * (tid // (2 * thread_n_blocks)) + (tid % (2 * thread_n_blocks)) + (2 * thread_n_blocks) * slice_col
        d_sh_rd = d_sh_stride * (tid // (2 * thread_n_blocks)) + (tid % (2 * thread_n_blocks))
        d_gl_wr_delta = d_gl_stride * (threads // (2 * thread_n_blocks))
        d_sh_rd_delta = d_sh_stride * (threads // (2 * thread_n_blocks))
        for _ in range(ceildiv(16 * thread_m_blocks, threads // (2 * thread_n_blocks))):
            ok = d_gl_wr < d_gl_stride * prob_m
            D4[d_gl_wr[ok]] = sh_h.reshape(-1, 8)[d_sh_rd[ok]]
            ACC4[d_gl_wr[ok]] = sh_i.reshape(-1, 8)[d_sh_rd[ok]]
            d_gl_wr = d_gl_wr + d_gl_wr_delta
            d_sh_rd = d_sh_rd + d_sh_rd_delta

    if grid == 1:
        for slice_col in range(n_tiles):  # gridDim.x == 1: the block walks the column slices in order (:792-812)
            write_out(slice_col, slice_pass(slice_col, 0, k_tiles))
        return D4.reshape(prob_m, prob_n), ACC4.reshape(prob_m, prob_n)

    # ---- a grid of threadblocks: the striped partition (:261-338) ----
    # (parallel == 1: prob_m <= 16 * thread_m_blocks is asserted above, so slice_col_par == slice_col and no pointer offsets)
    iters = ceildiv(k_tiles * n_tiles, grid)                                                            # :280
    if group_blocks != -1:
        iters = (group_blocks // thread_k_blocks) * ceildiv(iters, group_blocks // thread_k_blocks)     # :284-285
    work = {}   # column slice -> [(slice_idx, slice_count, block, fragments)]
    covered = np.zeros((n_tiles, k_tiles), np.int32)
    for b in range(grid):
        slice_row = (iters * b) % k_tiles                                                               # :287
        slice_col = (iters * b) // k_tiles                                                              # :288-289
        while True:
            # init_slice (:305-337)
            slice_iters = iters * (b + 1) - (k_tiles * slice_col + slice_row)
            if slice_iters < 0 or slice_col >= n_tiles:
                slice_iters = 0
            if slice_iters == 0:
                break
            if slice_row + slice_iters > k_tiles:
                slice_iters = k_tiles - slice_row
            slice_count, slice_idx = 1, 0
            col_first = iters * ceildiv(k_tiles * slice_col, iters)
            if col_first <= k_tiles * (slice_col + 1):
                col_off = col_first - k_tiles * slice_col
                slice_count = ceildiv(k_tiles - col_off, iters)
                if col_off > 0:
                    slice_count += 1
                delta_first = iters * b - col_first
                if delta_first < 0 or (col_off == 0 and delta_first == 0):
                    slice_idx = slice_count - 1
                else:
                    slice_idx = slice_count - 1 - delta_first // iters
                    if col_off > 0:
                        slice_idx -= 1
            covered[slice_col, slice_row:slice_row + slice_iters] += 1
            work.setdefault(slice_col, []).append((slice_idx, slice_count, b, slice_pass(slice_col, slice_row, slice_iters)))
            slice_row, slice_col = 0, slice_col + 1                                                     # :806-809
    assert (covered == 1).all(), "the stripes must cover every (column slice, k-tile) exactly once"
    # ---- global_reduce (:606-676) in lock order (barrier_acquire(&locks[slice_col], slice_idx), :213-237, :800-803): the block
    # with slice_idx 0 only writes its fragments to C, the others first add what C holds, all but the last write back ----
    C4 = np.full((16 * thread_m_blocks * prob_n // 4, 4), 0x7B7B7B7B, np.int64)   # int4 units of the int32 reduce buffer; poisoned
    active = 32 * thread_n_blocks // 4
    t = np.arange(active)
    c_gl_stride = prob_n // 4
    c_gl_wr_delta_o, c_gl_wr_delta_i = 8 * c_gl_stride, 8 * (active // 32)
    row = (t % 32) // 4
    for slice_col, parts in sorted(work.items()):
        parts.sort(key=lambda x: x[0])
        assert [p[0] for p in parts] == list(range(parts[0][1])) and all(p[1] == parts[0][1] for p in parts), (slice_col, [(p[0], p[1], p[2]) for p in parts])
        c_gl_wr = c_gl_stride * row + 8 * (t // 32) + (t % 4) * 2 + (4 * thread_n_blocks) * slice_col   # :618-619
        for slice_idx, slice_count, b, fc in parts:
            first, last = slice_idx == 0, slice_idx == slice_count - 1
            if slice_count == 1:
                write_out(slice_col, fc)
                continue
            # frag_c as the flat int array the kernel indexes: [thread_m_blocks][4][2][4] -> 32 ints per m-block
            flat = fc[:active].reshape(active, thread_m_blocks * 32).copy()
            for i in range(thread_m_blocks * 4):
                ok = np.full(active, True) if i < (thread_m_blocks - 1) * 4 else (8 * (i // 2) + row < prob_m)   # :627, :640
                addr = c_gl_wr + c_gl_wr_delta_o * (i // 2) + c_gl_wr_delta_i * (i % 2)
                for half in range(2):      # d_red1 / d_red2 (:642-655), d1 / d2 (:657-670)
                    idx = 4 * 2 * 4 * (i // 4) + 4 * (np.arange(4) + 4 * half) + (i % 4)
                    if not first:
                        flat[np.ix_(ok, idx)] += C4[addr[ok] + half]
                    if not last:
                        C4[addr[ok] + half] = flat[np.ix_(ok, idx)]
            fc = fc.copy()
            fc[:active] = flat.reshape(active, thread_m_blocks, 4, 2, 4)
            if last:
                write_out(slice_col, fc)
    return D4.reshape(prob_m, prob_n), ACC4.reshape(prob_m, prob_n)


def marlin_grid(A, B, s1, s2, s3=None, grid=3, thread_m_blocks=1, thread_n_blocks=8, thread_k_blocks=8, threads=256):
    """as marlin_one_block, for `grid` threadblocks: striped partition, partial column slices, global_reduce"""
    return marlin_one_block(A, B, s1, s2, s3, thread_m_blocks, thread_n_blocks, thread_k_blocks, threads, grid=grid)


def _mma_m16n8k16(fa, fb, nwarps):
    """mma.sync.aligned.m16n8k16.row.col.s32.s8.s8.s32 for every warp.  fa [threads, 2, 4] int8 (a0: row lane/4, a1: row
    lane/4 + 8; k = 4 * (lane % 4) + byte), fb [threads, 4] int8 (k = 4 * (lane % 4) + byte, n = lane / 4).  Returns
    [threads, 4]: c0, c1 = C[lane/4][2 * (lane % 4) + {0, 1}], c2, c3 = C[lane/4 + 8][...] (PTX ISA, matrix fragments for
    mma.m16n8k16 with .s8 operands)."""
    fa = fa.reshape(nwarps, 8, 4, 2, 4).astype(np.int64)   # [warp, g, kq, reg, byte]
    fb = fb.reshape(nwarps, 8, 4, 4).astype(np.int64)      # [warp, n, kq, byte]
    Am = np.concatenate([fa[:, :, :, 0, :].reshape(nwarps, 8, 16), fa[:, :, :, 1, :].reshape(nwarps, 8, 16)], axis=1)  # [warp, 16 rows, 16 k]
    Bm = fb.reshape(nwarps, 8, 16)                         # [warp, n, k]
    C = np.einsum("wrk,wnk->wrn", Am, Bm)                  # [warp, 16, 8]
    out = np.zeros((nwarps, 32, 4), np.int64)
    ln = np.arange(32)
    out[:, :, 0] = C[:, ln // 4, 2 * (ln % 4)]
    out[:, :, 1] = C[:, ln // 4, 2 * (ln % 4) + 1]
    out[:, :, 2] = C[:, ln // 4 + 8, 2 * (ln % 4)]
    out[:, :, 3] = C[:, ln // 4 + 8, 2 * (ln % 4) + 1]
    return out.reshape(nwarps * 32, 4)
