"""M-sharded GEMM + all-gather (qqq_amd/parallel.py) with world_size 2 on the gloo backend (CPU).
The per-rank compute is injected (here: the CPU oracle -- test infrastructure); on the GPU bench the same
ShardedGemm wraps qqq_amd.qqq_gemm over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from qqq_amd.parallel import ShardedGemm, allgather_us_model, gemm_us_model, pick_chunks, row_spans, take_rows


def test_pick_chunks_follows_the_cost_model():
    """the BASELINE point M=4096: communication ~ compute at 4 and 8 GPUs -> the pipeline must have >= 2 chunks
    there; a chunk never drops below 256 rows per rank; tiny shards go in one piece"""
    assert pick_chunks(4096, 8192, 1) == 1
    for P in (2, 4, 8):
        c = pick_chunks(4096, 8192, P)
        assert c >= 2 and -(-(4096 // P) // c) >= 256, (P, c)
    assert pick_chunks(512, 8192, 8) == 1 and pick_chunks(128, 8192, 2) == 1
    assert pick_chunks(32768, 4096, 2, K=4096) >= 2
    # the model itself: the estimate of the chosen count is within 2 % of the best count's
    for (M, N, P) in ((4096, 8192, 2), (4096, 8192, 8), (1024, 8192, 4), (32768, 11008, 8)):
        def est(c):
            w = -(-(-(-M // P)) // c)
            g, a = gemm_us_model(w, N, 21760), allgather_us_model(w, N, P)
            return g + (c - 1) * max(g, a) + a
        feasible = [c for c in (1, 2, 3, 4) if c == 1 or -(-(-(-M // P)) // c) >= 256]
        assert est(pick_chunks(M, N, P)) <= min(est(c) for c in feasible) * 1.021


def test_row_spans_cover_every_row_once():
    """chunk-cyclic ownership: super-block c is one contiguous slab in which rank p's piece sits at offset p*w"""
    for M in (0, 1, 5, 37, 64, 4096, 4097, 32768):
        for P in (1, 2, 3, 8):
            for chunks in (1, 2, 3, 4):
                owner = -np.ones(M, np.int64)
                w = -(-M // (P * chunks)) if M else 0
                for r in range(P):
                    sp = row_spans(M, P, r, chunks)
                    assert len(sp) == chunks
                    for c, (s, e) in enumerate(sp):
                        assert 0 <= s <= e <= M and e - s <= w
                        if e > s:
                            assert s == (c * P + r) * w
                            assert (owner[s:e] == -1).all()
                            owner[s:e] = r
                assert (owner >= 0).all()
    t = torch.arange(10).reshape(10, 1)
    assert take_rows(t, [(0, 2), (6, 8)]).flatten().tolist() == [0, 1, 6, 7]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, chunks, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import c_oracle as C, qqq_ref as R

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, K = 128, 256
        rng = np.random.default_rng(5)  # same data on every rank: weights replicated
        codes = rng.integers(-8, 8, size=(K, N), dtype=np.int8)
        B = R.pack_codes(codes, False)
        A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
        s1 = (rng.random((M, 1), dtype=np.float32) * 0.05 + 0.01)
        s2 = (rng.random((1, N), dtype=np.float32) * 0.01 + 0.001)
        full = C.qqq_gemm(A, B, s1, s2, None)

        def gemm_fn(a_rows, s1_rows, d_out):
            d = C.qqq_gemm(a_rows.numpy(), B, s1_rows.numpy(), s2, None)
            d_out.copy_(torch.from_numpy(d.copy()))

        sg = ShardedGemm(gemm_fn, chunks=chunks)
        spans = sg.spans(M, N)
        D = sg(torch.from_numpy(A), torch.from_numpy(s1), M, N)  # replicated inputs, sliced by ShardedGemm
        ok = np.array_equal(D.numpy().view(np.uint16), full.view(np.uint16))
        # caller-owned output, rows cut by the caller along spans() (local=True), called twice (scratch slab reuse)
        D2 = torch.full((M, N), float("nan"), dtype=torch.float16)
        sg(take_rows(torch.from_numpy(A), spans), take_rows(torch.from_numpy(s1), spans), M, N, D2, local=True)
        ok = ok and np.array_equal(D2.numpy().view(np.uint16), full.view(np.uint16))
        # a contiguous shard cut by the caller (the pre-round-2 ownership) must fail loudly, not mis-order rows
        if world > 1 and M >= world:
            lo, hi = rank * M // world, (rank + 1) * M // world
            try:
                sg(torch.from_numpy(A[lo:hi]), torch.from_numpy(s1[lo:hi]), M, N)
                ok = False
            except ValueError:
                pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M,chunks", [(64, 2), (37, 3), (1, 2), (130, None), (5, 4)])
def test_sharded_gemm_all_gather_gloo(M, chunks):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert res == {0: True, 1: True}
