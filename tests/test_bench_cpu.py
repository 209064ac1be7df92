"""bench.py's pure host logic (no GPU): the timed region replays EXACTLY K steps whatever the graph batching, and the algorithmic
work / byte counts are SURVEY 8d's."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_graph_schedule_replays_exactly_k_steps():
    import bench

    for steps in range(0, 130):
        for spg in (1, 2, 3, 5, 10, 16):
            sc = bench.graph_schedule(steps, spg)
            assert sum(sc) == steps, (steps, spg, sc)
            assert all(1 <= n <= max(spg, 1) for n in sc), (steps, spg, sc)
            if steps and spg > 1:
                assert sc[0] == 1 and len(set(sc)) <= 3, (steps, spg, sc)  # one-step opener; at most three graphs to capture
    assert bench.graph_schedule(20, 10) == [1, 9, 10]  # the driver's K
    assert bench.graph_schedule(100, 10) == [1, 9] + [10] * 9


def test_algorithmic_work_is_the_surveys():
    import bench

    N, K = bench.N_FULL, bench.K_FULL
    assert (N, K) == (8192, 21760) and tuple(bench.SWEEP_M) == (1, 16, 128, 1024, 4096)
    for M, ops, b_pc, b_g in ((1, 0.357e9, 89_199_876, 91_985_156), (4096, 1460.29e9, 245_415_936, 248_201_216)):
        assert abs(bench.algorithmic_ops(M, N, K) - ops) / ops < 2e-3
        assert bench.algorithmic_bytes(M, N, K, False) == b_pc and bench.algorithmic_bytes(M, N, K, True) == b_g
