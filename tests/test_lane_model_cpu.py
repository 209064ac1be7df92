"""The kernels' lane-level address arithmetic (tests/lane_model.py mirrors qqq_w4a8.hip) reproduces the
oracle's int32 accumulators: packed-layout decode, LDS swizzles, MFMA operand/accumulator maps, split-K,
m/n edge handling and the XCD-aware tile order.  CPU only."""
import numpy as np
import pytest

import lane_model as LM
from oracle import qqq_ref as R


def _case(rng, M, N, K, grouped):
    if grouped:
        codes = rng.integers(0, 16, size=(K, N), dtype=np.int8)
        s3 = (rng.random((K // 128, N), dtype=np.float32) * 8 + 0.5).astype(np.float16)
    else:
        codes = rng.integers(-8, 8, size=(K, N), dtype=np.int8)
        s3 = None
    B = R.pack_codes(codes, grouped)
    A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    acc = A.astype(np.int64) @ R.weight_operand(B, s3, grouped).astype(np.int64)
    return A, B, s3, acc


@pytest.mark.parametrize("grouped", [False, True])
def test_stream_kernel_model(grouped):
    rng = np.random.default_rng(11)
    A, B, s3, acc = _case(rng, 20, 192, 256, grouped)  # N % 128 == 64 edge, ragged m
    assert np.array_equal(LM.stream_kernel_model(A, B, s3, 20, 192, 256, MT=2, WAVES=4, ksplit=2, grouped=grouped), acc)
    assert np.array_equal(LM.stream_kernel_model(A, B, s3, 20, 192, 256, MT=1, WAVES=2, ksplit=1, grouped=grouped), acc)


@pytest.mark.parametrize("grouped", [False, True])
def test_column_kernel_model(grouped):
    rng = np.random.default_rng(13)
    A, B, s3, acc = _case(rng, 20, 192, 384, grouped)  # ragged m (2 m-blocks at MT=1), odd number of 64-k steps
    assert np.array_equal(LM.column_kernel_model(A, B, s3, 20, 192, 384, MT=1, WAVES=4, ksplit=1, grouped=grouped), acc)
    assert np.array_equal(LM.column_kernel_model(A, B, s3, 20, 192, 384, MT=2, WAVES=8, ksplit=2, grouped=grouped), acc)


@pytest.mark.parametrize("grouped", [False, True])
def test_tiled_kernel_model(grouped):
    rng = np.random.default_rng(12)
    A, B, s3, acc = _case(rng, 70, 320, 256, grouped)  # N % 256 == 64 edge, ragged m
    out, order = LM.tiled_kernel_model(A, B, s3, 70, 320, 256, BM=64, MTW=1, JW=2, ksplit=2, grouped=grouped,
                                       return_tile_order=True)
    assert np.array_equal(out, acc)
    assert len(set(order)) == len(order)
    assert np.array_equal(LM.tiled_kernel_model(A, B, s3, 70, 320, 256, BM=128, MTW=2, JW=2, ksplit=1, grouped=grouped), acc)
    # wave shapes of the 256-row tile: column-owner (one (jt, b) per wave) and 128 rows x one jt
    assert np.array_equal(LM.tiled_kernel_model(A, B, s3, 70, 320, 256, BM=256, MTW=8, JW=1, ksplit=1, grouped=grouped, NB=1), acc)
    assert np.array_equal(LM.tiled_kernel_model(A, B, s3, 70, 320, 256, BM=256, MTW=4, JW=1, ksplit=2, grouped=grouped, NB=2), acc)


def test_lds_swizzles_are_bank_conflict_free():
    """ds_read_b128 is serviced in 4 groups of 16 lanes, bank = (addr/4) % 64 (MI355X_MICROARCH LDS table)."""
    lane = np.arange(64)
    li, h = lane & 31, lane >> 5
    g, c = li >> 3, li & 7
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in grp] for grp in groups]

    def worst(addr):
        w = 0
        for grp in groups:
            banks = {}
            for l in grp:
                for d in range(4):
                    banks.setdefault(((addr[l] // 4) + d) % 64, set()).add(addr[l] + 4 * d)
            w = max(w, max(len(v) for v in banks.values()))
        return w

    for kq in range(4):
        assert worst(h * 2048 + g * 512 + (4 * c + (kq ^ g)) * 16) == 1
    for t in range(4):
        assert worst(16384 + li * 128 + (((2 * t + h) ^ ((li >> 1) & 7)) * 16)) == 1


@pytest.mark.parametrize("grouped", [False, True])
def test_panel_kernel_model(grouped):
    """The panel kernel's shapes: 32 / 64 columns per wave (HW), one / two k-groups, in-launch split-K, a trailing 64-k half
    stage (K % 128 == 64, per-channel only: group size 128), N not a multiple of the strip, ragged m."""
    rng = np.random.default_rng(14)
    K = 256 if grouped else 320
    A, B, s3, acc = _case(rng, 40, 320, K, grouped)
    kw = dict(grouped=grouped)
    assert np.array_equal(LM.panel_kernel_model(A, B, s3, 40, 320, K, MT=2, WN=4, KG=2, HW=1, ksplit=1, **kw), acc)   # BN = 128
    assert np.array_equal(LM.panel_kernel_model(A, B, s3, 40, 320, K, MT=1, WN=8, KG=1, HW=1, ksplit=2, **kw), acc)   # BN = 256, one k-group
    assert np.array_equal(LM.panel_kernel_model(A, B, s3, 40, 320, K, MT=4, WN=4, KG=2, HW=2, ksplit=2, **kw), acc)   # 64 columns per wave
    A, B, s3, acc = _case(rng, 130, 256, K, grouped)
    assert np.array_equal(LM.panel_kernel_model(A, B, s3, 130, 256, K, MT=8, WN=4, KG=2, HW=2, ksplit=1, **kw), acc)  # the M >= 768 shape


def test_panel_split_k_slot_and_fragment_layout():
    """Every finishing wave's deposit lands on its own ints of the slot (the fold of the last arrival reads them back with
    the same formula), and the activation fragment reads are bank-conflict free (ds_read_b128: 4 groups of 16 lanes)."""
    for MT, WN, KG, HW in ((8, 4, 2, 1), (8, 4, 2, 2), (8, 8, 1, 1), (4, 4, 2, 1), (2, 4, 2, 2), (1, 4, 2, 1), (1, 8, 1, 1)):
        idx = LM.panel_slot_index(MT, WN, KG, HW)
        assert idx.size == 16 * MT * 32 * WN * HW and np.array_equal(np.sort(idx), np.arange(idx.size)), (MT, WN, KG, HW)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in grp] for grp in groups]
    for mt in (0, 3, 7):
        for tk in (0, 1):
            addr = LM.panel_fragment_addr(mt, tk)
            for grp in groups:
                banks = {}
                for l in grp:
                    for d in range(4):
                        banks.setdefault(((addr[l] // 4) + d) % 64, set()).add(int(addr[l]) + 4 * d)
                assert max(len(v) for v in banks.values()) == 1, (mt, tk)


@pytest.mark.parametrize("grouped", [False, True])
def test_wide_kernel_model(grouped):
    """the wide kernel's address arithmetic (round 3): buffer-descriptor offsets of the weight / scale / activation loads, the
    chunk-by-chunk LDS stage image, fragment reads, the row-major epilogue / split-K slot image, K slices, ragged m and n,
    both tile heights, and the XCD-aware 1-D tile order with the slices of a tile side by side"""
    rng = np.random.default_rng(17)
    K = 512
    A, B, s3, acc = _case(rng, 300, 320, K, grouped)       # N % 256 == 64, m not a multiple of either tile height
    kw = dict(grouped=grouped)
    got, _ = LM.wide_kernel_model(A, B, s3, 300, 320, K, MT=16, ksplit=1, **kw)
    assert np.array_equal(got, acc)
    got, slots = LM.wide_kernel_model(A, B, s3, 300, 320, K, MT=8, ksplit=2, PW=4, **kw)
    assert np.array_equal(got, acc) and len(slots) == 3 * 2 * 1 and all(sl.shape == (128, 256) for sl in slots)
    A, B, s3, acc = _case(rng, 70, 512, 1024, grouped)
    got, slots = LM.wide_kernel_model(A, B, s3, 70, 512, 1024, MT=16, ksplit=3, **kw)
    assert np.array_equal(got, acc) and len(slots) == 2 * 2
    # 256 x 128 tiles: a wave owns one 32-column half of a 64-column group (ragged n: 320 = 2 strips + half a strip)
    got, slots = LM.wide_kernel_model(A, B, s3, 70, 512, 1024, MT=16, ksplit=2, BN=128, **kw)
    assert np.array_equal(got, acc) and len(slots) == 4 and all(sl.shape == (256, 128) for sl in slots)
    A, B, s3, acc = _case(rng, 300, 320, K, grouped)
    got, _ = LM.wide_kernel_model(A, B, s3, 300, 320, K, MT=16, ksplit=1, BN=128, PW=4, **kw)
    assert np.array_equal(got, acc)


def test_wide_tile_order_is_a_bijection_and_keeps_slices_together():
    for (tm, tn, ks, pw) in ((16, 32, 1, 8), (4, 32, 2, 8), (5, 43, 1, 8), (3, 7, 3, 4), (1, 1, 2, 8), (8, 16, 1, 16)):
        order = LM.wide_tile_order(tm, tn, ks, pw)
        assert sorted((a, b, c) for a, b, c, _ in order) == [(a, b, c) for a in range(tm) for b in range(tn) for c in range(ks)]
        # the slices of a tile sit at consecutive positions of one XCD's run: block ids 8 apart (same XCD) or the run's edge
        pos = {}
        for bid, (a, b, c, lin) in enumerate(order):
            pos.setdefault((a, b), []).append(bid)
        if ks > 1:  # only a tile that straddles the edge between two XCDs' runs is split: at most 7 of them
            split = sum(len({p % 8 for p in v}) > 1 for v in pos.values())
            assert split <= 7, (tm, tn, ks, split, len(pos))
