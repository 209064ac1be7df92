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


def _worst_case_result():
    """round 4's full 21 KB result (the one the driver could not parse) + an N > 1 block + pessimistic float lengths"""
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r04_bench_final.json")))
    d["multi_gpu"] = {str(M): {"gemm_only_us": 123.456789012345, "allgather_only_us": 98.7654321098765, "overlapped_us": 150.123456789012,
                               "chunks": 4, "rows_per_rank": M // 8} for M in (1024, 4096)}
    d["multi_gpu"].update(world_seen=8, backend="nccl (RCCL)", launch="eager (hipGraph capture failed on at least one rank)")
    d["detail"] = "gpurun_out/bench_detail_n8.json"
    d["device"] = "AMD Instinct MI355X " * 2
    return d


def test_printed_line_is_bounded_and_carries_the_contract():
    """the driver keeps an 8 KB stdout tail: the printed line must stay far below it and still hold every contract key,
    `roofline` and `cpu_baseline` (BENCH_r04.parsed = null)"""
    import json

    import bench

    full = _worst_case_result()
    assert len(json.dumps(full)) > 15000  # the input really is the oversized one
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 4000, len(line)
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_hbm", "cpu_baseline", "per_m", "per_m_g128", "llama7b"):
        assert k in out, k
    assert "workload" in out["config"] and "model" not in out["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"] and k in out["roofline_hbm"], k
    assert abs(out["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-4
    assert abs(out["roofline"]["achieved"] / out["roofline"]["peak"] - out["roofline"]["frac"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in out["cpu_baseline"], k
    assert set(out["per_m"]) == {"1", "16", "128", "1024", "4096"}
    assert abs(out["value"] - full["value"]) / full["value"] < 1e-4


def test_printed_line_survives_pathological_blocks():
    """an optional block that explodes (say, a per-layer table leaking into llama7b) is dropped, never the contract keys"""
    import json

    import bench

    full = _worst_case_result()
    full["llama7b"]["sum_of_7_linears"]["per_channel"] = {str(i): {"quantlinear_us": 1.0 / 3, "speedup": 2.0 / 3} for i in range(400)}
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_LIMIT
    out = json.loads(line)
    assert "roofline" in out and "cpu_baseline" in out and out["llama7b"] == {"dropped": "see detail file"}
