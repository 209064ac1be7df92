#!/usr/bin/env python3
"""BASELINE configs[3]: Llama-2-7B linear shapes, batch in {1,8,32} x seq 1024 (M = 1024, 8192, 32768 tokens),
QuantLinear (fused dynamic_quant + W4A8 GEMM) vs fp16 nn.Linear on the same GPU.  Synthetic weights/tokens.
Prints one JSON object (kept under profiles/)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qqq_amd import QuantLinear, pack as P

LAYERS = [("q_proj", 4096, 4096), ("k_proj", 4096, 4096), ("v_proj", 4096, 4096), ("o_proj", 4096, 4096),
          ("gate_proj", 11008, 4096), ("up_proj", 11008, 4096), ("down_proj", 4096, 11008)]  # (name, N, K)


def make_ql(dev, N, K, group_size, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    ql = QuantLinear(4, group_size, K, N, bias=False).to(dev)
    grouped = group_size != -1
    if grouped:
        codes = torch.randint(0, 16, (K, N), generator=g, dtype=torch.int8, device=dev)
        ql.s_group.copy_((torch.rand((K // 128, N), generator=g, device=dev) * 15 + 0.5).half())
    else:
        codes = torch.randint(-7, 8, (K, N), generator=g, dtype=torch.int8, device=dev)
    ql.B.copy_(P.pack_codes(codes, grouped))
    ql.s_channel.copy_(torch.rand((1, N), generator=g, device=dev) * 2e-4 + 1e-5)
    return ql


def time_fn(fn, iters=10, reps=8):
    """median per-call GPU time with the launches replayed from a hipGraph (`reps` calls per graph), i.e.
    without Python / launch-enqueue gaps between the kernels -- the way a serving stack issues a layer.
    Both the QuantLinear and the fp16 nn.Linear are timed this way."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); g.replay(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]) * 1e3 / reps)


TOKENS = (1024, 8192, 32768)  # batch 1 / 8 / 32 x seq 1024


def llama_matrix(dev, tokens=TOKENS, modes=(-1, 128, (128, True)), budget_s=None, merged=True):
    """BASELINE configs[3] as a dict: per linear / token count / mode the QuantLinear time (fused dynamic quant + W4A8 GEMM),
    the GEMM alone, the fp16 nn.Linear, `gemm_tops` and `speedup_vs_fp16`; the sum over the 7 linears of a block; and (merged)
    the same block with the projections that share an input fused.  One implementation for `python
    tools/bench_llama.py` and for bench.py's `llama7b` object.  budget_s: wall-clock bound; what did not fit is listed under
    "skipped" (nothing is extrapolated).  A mode is a group size or (group size, True): the layers with expand_for_prefill() -- the
    opt-in load-time int8 expansion (round 6) -- under the key "g128_expanded"; the fp16 time of a (layer, token count) is measured once."""
    import time
    from qqq_amd import fuse_quant_linears, ops

    t_start = time.perf_counter()
    out = {"device": torch.cuda.get_device_name(dev), "launch": "hipGraph replay (8 calls per graph)", "tokens": list(tokens),
           "layers": {}, "skipped": []}

    def over():
        return budget_s is not None and time.perf_counter() - t_start > budget_s

    fp16_us = {}
    for spec in modes:
        gs, expand = spec if isinstance(spec, tuple) else (spec, False)
        mode = ("per_channel" if gs == -1 else "g128") + ("_expanded" if expand else "")
        tot = {}
        complete = True
        for (name, N, K) in LAYERS:
            if over():
                out["skipped"].append(f"{mode}/{name}")
                complete = False
                continue
            ql = make_ql(dev, N, K, gs, hash((name, gs)) & 0xFFFF)
            if expand:
                ql.expand_for_prefill(per_channel=True)
            lin = torch.nn.Linear(K, N, bias=False).half().to(dev)
            for M in tokens:
                x = torch.randn((M, K), device=dev, dtype=torch.float16)
                t_q = time_fn(lambda: ql(x))
                xq, s1 = ql.dynamic_quant(x)
                D = torch.empty((M, N), dtype=torch.float16, device=dev)
                t_g = time_fn(lambda: ops.qqq_gemm_w8(xq, ql.B, ql.reduce_buffer, D, s1, ql.s_channel, ql.s_group, ql.workspace, None, ql.W8, 16))
                if (N, K, M) not in fp16_us:
                    fp16_us[(N, K, M)] = time_fn(lambda: lin(x))
                t_f = fp16_us[(N, K, M)]
                out["layers"].setdefault(mode, {}).setdefault(name, {})[str(M)] = {
                    "quantlinear_us": t_q, "gemm_only_us": t_g, "fp16_linear_us": t_f,
                    "gemm_tops": 2.0 * M * N * K / t_g / 1e6, "speedup_vs_fp16": t_f / t_q}
                tot.setdefault(M, [0.0, 0.0])
                tot[M][0] += t_q; tot[M][1] += t_f
                del x, D
            del ql, lin
            torch.cuda.empty_cache()
        if complete:
            out.setdefault("sum_of_7_linears", {})[mode] = {str(M): {"quantlinear_us": a, "fp16_us": b, "speedup": b / a} for M, (a, b) in tot.items()}
        if not merged:
            continue
        # the same block with the projections that share an input merged (what vLLM does): q/k/v -> one layer
        # with N = 12288, gate/up -> N = 22016; the fp16 side gets the same merge (one nn.Linear each)
        ftot = {}
        complete = True
        for (name, parts, K) in (("qkv_proj", [4096, 4096, 4096], 4096), ("o_proj", [4096], 4096), ("gate_up_proj", [11008, 11008], 4096), ("down_proj", [4096], 11008)):
            if over():
                out["skipped"].append(f"{mode}/merged/{name}")
                complete = False
                continue
            qls = [make_ql(dev, n, K, gs, hash((name, i, gs)) & 0xFFFF) for i, n in enumerate(parts)]
            ql = fuse_quant_linears(qls) if len(qls) > 1 else qls[0]
            if expand:
                ql.expand_for_prefill(per_channel=True)
            N = sum(parts)
            lin = torch.nn.Linear(K, N, bias=False).half().to(dev)
            for M in tokens:
                x = torch.randn((M, K), device=dev, dtype=torch.float16)
                t_q = time_fn(lambda: ql(x))
                if (N, K, M) not in fp16_us:
                    fp16_us[(N, K, M)] = time_fn(lambda: lin(x))
                t_f = fp16_us[(N, K, M)]
                out.setdefault("fused_layers", {}).setdefault(mode, {}).setdefault(name, {})[str(M)] = {
                    "quantlinear_us": t_q, "fp16_linear_us": t_f, "speedup_vs_fp16": t_f / t_q}
                ftot.setdefault(M, [0.0, 0.0])
                ftot[M][0] += t_q; ftot[M][1] += t_f
                del x
            del ql, qls, lin
            torch.cuda.empty_cache()
        if complete:
            out.setdefault("sum_of_4_merged_linears", {})[mode] = {str(M): {"quantlinear_us": a, "fp16_us": b, "speedup": b / a} for M, (a, b) in ftot.items()}
    out["seconds"] = time.perf_counter() - t_start
    return out


def main():
    print(json.dumps(llama_matrix(torch.device("cuda:0"))))


if __name__ == "__main__":
    main()
