#!/usr/bin/env python3
"""bench.py -- W4A8 GEMM throughput on MI355X behind QQQ's `qqq_gemm` (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = one pass of the hot path over the BASELINE sweep (configs[1]): five `qqq_gemm` calls,
per-channel W4A8, M in {1,16,128,1024,4096}, N=8192, K=21760, on synthetic int8 activations (full int8
range, produced by the fused dynamic_quant of N(0,1) fp16 tokens) and random int4 weights, all resident
in HBM before the timed region.  Consecutive calls use DIFFERENT 89 MB weight buffers (12 of them, 1.07 GB
> the 256 MiB Infinity Cache) so that "HBM GB/s" is not an L3 number.  `value` = sum(2*M*N*K) / time.
The timed region replays hipGraphs: a one-step graph opens it (the GPU idles behind the barrier until the host has launched the
first graph, and a short one launches fastest), the rest hold several consecutive steps each (--steps-per-graph, default 10,
plus one shorter graph for the remainder; exactly K steps run): the idle gap at a replay boundary is launch plumbing, not GEMM.

N > 1 (BASELINE configs[4]): same sweep, rows of every point with M >= 64*N sharded over the ranks
(weights replicated), output shards all-gathered over RCCL/xGMI, chunk-pipelined against the GEMM
(qqq_amd/parallel.py); smaller points are computed redundantly on every rank (no collective).  Strong
scaling: the total work is fixed.

Extra objects on the JSON line: `roofline` (dominant kernel = whatever the dispatcher runs at M=4096, durations
from HIP event pairs recorded on the launch stream inside this process), `roofline_hbm` (the HBM-bound
decode kernel at M=1), `cpu_baseline` (the C oracle timed on the host cores, bounded sample) and
`per_m` (per sweep point: us, TOPS, GB/s, speedup vs torch fp16 GEMM on the same GPU).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FULL, K_FULL = 8192, 21760
SWEEP_M = (1, 16, 128, 1024, 4096)
NBUF = 12  # weight copies per mode: 1.07 GB, so that a rotation never finds a copy in the 256 MiB Infinity Cache whatever its replacement policy (round 5: with 5 copies = 445 MB
           # the per-M timing loop of the plain-load kernels did: decode read 17.9 us where 12 copies give 19.8; the column kernel's nt loads do not use that cache either way)
MAX_PAR = 16
PEAK_MFMA_TOPS = 5033.0  # dense int8: 256 CU x 2.4 GHz x 8192 op/clk/CU (MI355X_MICROARCH.md, = 2x bf16 2.5 PF)
PEAK_HBM_GBS = 8000.0    # HBM3E spec; ~6300 achievable (MI355X_MICROARCH.md)


def algorithmic_bytes(M, N, K, grouped=False):
    b = M * K + K * N // 2 + 2 * M * N + 4 * M + 4 * N
    if grouped:
        b += 2 * (K // 128) * N
    return b


def algorithmic_ops(M, N, K):
    return 2 * M * N * K


def make_weights(dev, grouped, nbuf, seed=0, N=N_FULL, K=K_FULL, dist=None):
    """`nbuf` independent packed weight buffers + the scales of buffer 0.
    dist "gptq" (default): W ~ N(0, 0.02^2) fp16 [N,K] quantised the way the reference's GPTQ
    flow does it -- per-channel scale = max_k|W|/7, w4 = clamp(round(W/scale), -7, 7), s_channel = scale/16
    (qlinear_marlin.py:222-226); per-group scale_g = 2 max|W_g|/15, u = clamp(round(W/scale_g)+8, 0, 15),
    s_extra = max_k|W_fq|/127, s_group = half(scale_g/s_extra), s_channel = s_extra (gptq.py:204-216,
    qlinear_marlin.py:209-219).  dist "uniform": uniformly random nibbles (worst case for MFMA toggling power)."""
    from qqq_amd import pack as P

    dist = dist or os.environ.get("QQQ_BENCH_WEIGHTS", "gptq")
    g = torch.Generator(device=dev).manual_seed(seed)
    Bs, s2, s3 = [], None, None
    for i in range(nbuf):
        if dist == "uniform":
            if grouped:
                codes = torch.randint(0, 16, (K, N), generator=g, dtype=torch.int8, device=dev)
            else:
                codes = torch.randint(-7, 8, (K, N), generator=g, dtype=torch.int8, device=dev)
            if i == 0:
                s2 = (torch.rand((1, N), generator=g, device=dev) * 2e-4 + 1e-5).to(torch.float32)
                if grouped:
                    s3 = (torch.rand((K // 128, N), generator=g, device=dev) * 15.0 + 0.5).to(torch.float16)
        else:
            W = (torch.randn((N, K), generator=g, device=dev, dtype=torch.float32) * 0.02).half().float()
            if grouped:
                Wg = W.view(N, K // 128, 128)
                scale_g = 2.0 * Wg.abs().amax(dim=2) / 15.0  # [N, G]
                u = torch.clamp(torch.round(Wg / scale_g[..., None]) + 8, 0, 15)
                if i == 0:
                    s_extra = ((u - 8) * scale_g[..., None]).abs().amax(dim=(1, 2)) / 127.0  # [N]
                    s3 = (scale_g / s_extra[:, None]).t().contiguous().to(torch.float16)
                    s2 = s_extra[None, :].contiguous().to(torch.float32)
                codes = u.view(N, K).t().contiguous().to(torch.int8)
                del Wg, u, scale_g
            else:
                scale = W.abs().amax(dim=1, keepdim=True) / 7.0
                codes = torch.clamp(torch.round(W / scale), -7, 7).t().contiguous().to(torch.int8)
                if i == 0:
                    s2 = (scale.t() / 16.0).contiguous().to(torch.float32)
            del W
        Bs.append(P.pack_codes(codes, grouped))
        del codes
    return Bs, s2, s3


def make_tokens(dev, M, seed, K=K_FULL):
    from qqq_amd import dynamic_quant

    g = torch.Generator(device=dev).manual_seed(100 + seed)
    x = torch.randn((M, K), generator=g, device=dev, dtype=torch.float32).to(torch.float16)
    xq, s1 = dynamic_quant(x)
    return xq, s1


class Layer:
    """device buffers of one QuantLinear-like layer + the rotating weight copies"""

    def __init__(self, dev, grouped=False, nbuf=NBUF, N=N_FULL, K=K_FULL, expand=False):
        self.dev, self.N, self.K, self.grouped = dev, N, K, grouped
        self.Bs, self.s2, s3 = make_weights(dev, grouped, nbuf, N=N, K=K)
        self.s3 = s3 if s3 is not None else torch.empty(0, dtype=torch.float16, device=dev)
        self.C = torch.zeros((MAX_PAR * 64, N), dtype=torch.int32, device=dev)
        self.ws = torch.zeros(N // 128 * MAX_PAR, dtype=torch.int32, device=dev)
        self.groupsize = 128 if grouped else -1
        self.W8s = None
        if expand and grouped:
            self.expand()

    def expand(self):
        """opt-in load-time re-layout (QuantLinear.expand_for_prefill): every weight copy also as expanded int8 (K x N bytes each)"""
        from qqq_amd import ops

        self.W8s = [ops.expand_int8(B, self.s3) for B in self.Bs]
        torch.cuda.synchronize()
        return self

    def time_calls(self, A, s1, D, iters, tune=None, rotate=True):
        """per-call durations (ms) from HIP event pairs recorded natively around each launch"""
        from qqq_amd import _dev, _lib

        L = _dev.lib()  # the event-timed loop lives in the test/tuning library; it calls the operator library's GEMM
        nb = len(self.Bs) if rotate else 1
        rot = getattr(self, "_rot", 0) % nb  # the rotation goes on where the previous timed group stopped: a copy comes back after ALL the others
        self._rot = rot + iters
        arr = (ctypes.c_void_p * nb)(*[self.Bs[(rot + i) % nb].data_ptr() for i in range(nb)])
        out = (ctypes.c_float * iters)()
        tn = None
        if tune:
            tn = _lib.QQQTune()
            for k, v in tune.items():
                setattr(tn, k, int(v))
        st = torch.cuda.current_stream(self.dev).cuda_stream
        if self.W8s is not None:  # the calls also get the copy's expanded int8 weights (the library uses them where its plan is the wide kernel's)
            arr8 = (ctypes.c_void_p * nb)(*[self.W8s[(rot + i) % nb].data_ptr() for i in range(nb)])
            rc = L.qqq_dev_bench_gemm2(
                _dev.gemm_ex2_ptr(), A.data_ptr(), arr, arr8, nb, self.C.data_ptr(), D.data_ptr(), s1.data_ptr(), self.s2.data_ptr(),
                self.s3.data_ptr() if self.s3.numel() else None, A.shape[0], self.N, self.K, self.ws.data_ptr(),
                self.groupsize, self.dev.index or 0, ctypes.c_void_p(st), MAX_PAR,
                ctypes.byref(tn) if tn is not None else None, iters, out,
            )
        else:
            rc = L.qqq_dev_bench_gemm(
                _dev.gemm_ex_ptr(), A.data_ptr(), arr, nb, self.C.data_ptr(), D.data_ptr(), s1.data_ptr(), self.s2.data_ptr(),
                self.s3.data_ptr() if self.s3.numel() else None, A.shape[0], self.N, self.K, self.ws.data_ptr(),
                self.groupsize, self.dev.index or 0, ctypes.c_void_p(st), MAX_PAR,
                ctypes.byref(tn) if tn is not None else None, iters, out,
            )
        if rc:
            raise RuntimeError(f"qqq_dev_bench_gemm rc={rc}: {_lib.last_error()} {_dev.last_error()}")
        return np.array(out[:], dtype=np.float64)


def copies_for(N, K, total_bytes=1.1e9, lo=4, hi=160):
    """weight copies of an N x K layer that make a rotation longer than anything the caches hold (tools/ab.py, tools/dispatch_check.py)"""
    return int(min(hi, max(lo, -(-total_bytes // (N * K // 2)))))


FAMILY = {1: "stream", 2: "tiled", 3: "column", 4: "panel", 5: "wide"}


def plan_label(pln):
    """kernel family of a plan as a label; a call the library splits along M (plan field split_m: rows of the first of two launches) says so --
    the other plan fields then describe the unsplit call, which is not what runs"""
    name = FAMILY[pln["kernel"]]
    return name if not pln.get("split_m") else f"{name}, two launches ({pln['split_m']} rows + the rest)"


def graph_schedule(steps, spg):
    """Sizes (in steps) of the hipGraphs replayed in the timed region, in order: a one-step opener, the remainder of
    (steps - 1) % spg, then spg steps per graph; they add up to EXACTLY `steps`."""
    if steps <= 0:
        return []
    if spg <= 1:
        return [1] * steps
    rest = steps - 1
    return [1] + ([rest % spg] if rest % spg else []) + [spg] * (rest // spg)


def fp16_gemm_us(dev, M, iters=10, N=N_FULL, K=K_FULL):
    """torch fp16 GEMM (hipBLASLt) on the same GPU, weights rotated over 2 x 356 MB buffers"""
    Ws = [torch.randn((K, N), device=dev, dtype=torch.float16) * 0.02 for _ in range(2)]
    x = torch.randn((M, K), device=dev, dtype=torch.float16)
    for i in range(3):
        torch.matmul(x, Ws[i % 2])
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i, (a, b) in enumerate(evs):
        a.record()
        torch.matmul(x, Ws[i % 2])
        b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
    del Ws
    return float(np.median(t))


def int8_gemm_us(dev, M, iters=10, N=N_FULL, K=K_FULL):
    """vendor int8 x int8 -> int32 GEMM (torch._int_mm = hipBLASLt) on the same GPU and shape: what a library kernel
    gets out of the int8 matrix pipe under the same power limit (2x the weight bytes, no scales, no fp16 epilogue --
    a ceiling reference, not an equivalent operator).  None when the op rejects the shape (it needs M > 16)."""
    try:
        Ws = [torch.randint(-128, 128, (K, N), device=dev, dtype=torch.int8) for _ in range(3)]
        x = torch.randint(-128, 128, (M, K), device=dev, dtype=torch.int8)
        for i in range(3):
            torch._int_mm(x, Ws[i % 3])
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for i, (a, b) in enumerate(evs):
            a.record()
            torch._int_mm(x, Ws[i % 3])
            b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs]) * 1e3)
    except Exception:
        return None


def cpu_baseline(sample_rows=64, budget_s=12.0, fp16_budget_s=30.0):
    """The C oracle (restatement of the reference arithmetic, `kind: port`) on the host cores, on a bounded
    sample of the same workload: `sample_rows` tokens x the full N x K weight matrix, repeated until
    ~budget_s of CPU work.  Also times torch's fp16 CPU GEMM (north_star's "PyTorch fp16 CPU GEMM")."""
    from oracle import c_oracle as C
    from oracle import qqq_ref as R

    rng = np.random.default_rng(0)
    codes = rng.integers(-7, 8, size=(K_FULL, N_FULL), dtype=np.int8)
    B = C.pack(codes, False)
    A = rng.integers(-128, 128, size=(sample_rows, K_FULL), dtype=np.int8)
    s1 = (rng.random((sample_rows, 1), dtype=np.float32) * 0.05 + 0.001)
    s2 = (rng.random((1, N_FULL), dtype=np.float32) * 2e-4 + 1e-5)
    C.qqq_gemm(A[:4], B, s1[:4], s2)  # warm (page-in, omp pool)
    t0 = time.perf_counter()
    reps = 0
    while True:
        C.qqq_gemm(A, B, s1, s2)
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= 20:
            break
    dt = (time.perf_counter() - t0) / reps
    tops = algorithmic_ops(sample_rows, N_FULL, K_FULL) / dt / 1e12
    cores = os.cpu_count() or 1
    out = {
        "value": tops, "unit": "TOPS", "cores": cores, "kind": "port",
        "sample": f"C oracle (oracle/qqq_oracle.c, OpenMP, includes int4 unpack) M={sample_rows} N={N_FULL} K={K_FULL} per-channel, {reps} reps, {dt*1e3:.0f} ms each",
    }
    try:
        torch.set_num_threads(cores)
        W = (torch.randn((N_FULL, K_FULL)) * 0.02).to(torch.float16)
        res = {}
        t_fp16 = time.perf_counter()
        for M in (1, 16, 128, 1024, 4096):  # the whole sweep (north_star: "in the same run"); one repetition at large m
            if M > 128:
                # bounded: a full-size point is run only if its time, extrapolated from the M=128 rate, fits what is left of the
                # budget (measured on the 256-thread bench host: 2 s and 15-19 s; an ordinary host would take minutes); otherwise
                # a row sample of the same GEMM is timed and scaled, and the entry says so
                est = res["128"]["ms"] * 1e-3 * M / 128.0
                left = fp16_budget_s - (time.perf_counter() - t_fp16)
                rows = M if est <= left else max(128, int(M * max(left, 1.0) / est) // 128 * 128)
            else:
                rows = M
            x = torch.randn((rows, K_FULL)).to(torch.float16)
            if M <= 128:
                torch.matmul(x, W.t())
            ts = []
            for _ in range(3 if M <= 128 else 1):
                t1 = time.perf_counter()
                torch.matmul(x, W.t())
                ts.append(time.perf_counter() - t1)
            ms = float(np.median(ts) * 1e3) * M / rows
            res[str(M)] = {"ms": ms, "tflops": algorithmic_ops(M, N_FULL, K_FULL) / (ms * 1e-3) / 1e12}
            if rows != M:
                res[str(M)]["sampled_rows"] = rows
        out["torch_fp16_cpu_gemm"] = {"threads": torch.get_num_threads(), "per_m": res}
    except Exception as e:  # pragma: no cover
        out["torch_fp16_cpu_gemm"] = {"error": str(e)}
    return out


def sustained_matrix_rate(dev, reps=3, iters=20000, nwg=256):
    """What the matrix pipe sustains on THIS part under its power limit, on random int8 operands, in the tiled kernel's
    k-step shape (dev library probe, tools/mfma_ceiling.py): rung 0 = 32x32x32 MFMAs on register operands only, rung 3 = the
    same MACs issued as 16x16x64 MFMAs (the panel kernel's instruction: half the accumulator traffic per MAC, it sustains
    ~15 % more), rung 2 = 32x32x32 + the k-step's LDS fragment reads + the int4 unpack.  Context for `roofline.frac` (profiles/r02_mfma_power_ceiling.txt)."""
    from qqq_amd import _dev

    L = _dev.lib()
    g = torch.Generator(device=dev).manual_seed(3)
    buf = torch.randint(-128, 128, (nwg * (65536 + 512 * 12 * 16),), generator=g, dtype=torch.int8, device=dev)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = {}
    for mode, key in ((0, "mfma_only_tops"), (3, "mfma_16x16x64_only_tops"), (2, "mfma_lds_unpack_tops")):
        v = []
        for _ in range(reps):
            ms = ctypes.c_float()
            if L.qqq_dev_probe_mfma_rate(mode, buf.data_ptr(), nwg, iters, sink.data_ptr(), dev.index or 0, st, ctypes.byref(ms)) != 0:
                return {"error": _dev.last_error()}
            v.append(ms.value)
        out[key] = nwg * 8 * iters * 8 * 65536.0 / float(np.median(v)) / 1e9
    return out


def committed_traffic(key):
    """bytes per launch from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json, tools/pmc_traffic.sh)"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        return d[key]["bytes"]
    except Exception:
        return None


def live_traffic(timeout_s=240):
    """HBM/fabric bytes per launch of the two roofline kernels, measured NOW with rocprofv3 exactly as
    MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (with
    --kernel-trace only), read bytes = 2 x FETCH_SIZE KiB (gfx950 correction), write bytes = WRITE_SIZE KiB.
    Each pass re-runs the same launches (tools/prof_calls.py: M=1 and M=4096 on the BASELINE layer) in a child process.
    Returns ({kernel-key: bytes}, note); on any failure ({}, reason) and the caller falls back to the committed pass."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="qqq_pmc_", dir="/tmp")
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, c)
            cmd = [exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "t", "--",
                   sys.executable, os.path.join(ROOT, "tools", "prof_calls.py"), "--ms", "1,4096", "--iters", "3"]
            p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return {}, f"rocprofv3 --pmc {c} failed (rc={p.returncode})"
            for r in csv.DictReader(open(files[0])):
                if r["Counter_Name"] != c:
                    continue
                name = r["Kernel_Name"]
                # the child runs M=1 (decode: column / stream kernel, <= 512 workgroups) and M=4096 (wide / tiled / panel)
                big = any(k in name for k in ("qqq_tiled_kernel", "qqq_panel_kernel", "qqq_wide_kernel"))
                small = any(k in name for k in ("qqq_column_kernel", "qqq_stream_kernel"))
                key = "tiled_m4096" if big else "column_m1" if small else None
                if key:
                    vals.setdefault(key, {}).setdefault(c, []).append(float(r["Counter_Value"]))
        res = {}
        for key, v in vals.items():
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                res[key] = int(2 * 1024 * np.mean(v["FETCH_SIZE"]) + 1024 * np.mean(v["WRITE_SIZE"]))
        return res, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) run by bench.py after the timed region"
    except Exception as e:  # pragma: no cover
        return {}, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def gpu_clocks(dev, busy_fn=None):
    """SURVEY 8d "print rocminfo clocks": the part's maximum shader clock (rocminfo) and the clock rocm-smi reports while the GPU
    is busy with `busy_fn` (about a second of M=4096 launches queued ahead of the query) -- under the wide kernel the chip runs
    well below its 2.4 GHz maximum (power), which is half of what `roofline.frac` is made of."""
    import re
    import shutil
    import subprocess

    out = {"max_mhz": None, "busy_sclk": None}
    try:
        exe = shutil.which("rocminfo") or "/opt/rocm/bin/rocminfo"
        txt = subprocess.run([exe], capture_output=True, text=True, timeout=30).stdout
        blocks = [b for b in txt.split("*******") if "gfx" in b and "Max Clock Freq" in b]
        if blocks:
            out["max_mhz"] = int(re.search(r"Max Clock Freq\. \(MHz\):\s+(\d+)", blocks[0]).group(1))
    except Exception as e:  # pragma: no cover
        out["rocminfo_error"] = f"{type(e).__name__}: {e}"
    try:
        exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        if busy_fn is not None:
            busy_fn()
        txt = subprocess.run([exe, "--showclocks", "-d", str(dev.index or 0)], capture_output=True, text=True, timeout=30).stdout
        torch.cuda.synchronize()
        m = re.search(r"sclk clock level[^\n]*\((\d+)Mhz\)", txt)
        out["busy_sclk"] = {"mhz": int(m.group(1)) if m else None,
                            "raw": [ln.strip() for ln in txt.split("\n") if "sclk" in ln or "mclk" in ln][:4],
                            "note": "rocm-smi --showclocks sampled while ~1 s of M=4096 launches was queued"}
    except Exception as e:  # pragma: no cover
        out["rocm_smi_error"] = f"{type(e).__name__}: {e}"
    return out


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _r(x, sig=5):
    """floats to `sig` significant digits (the printed line is bounded; the side file keeps full precision)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


LINE_LIMIT = 4000  # bytes; the driver keeps an 8 KB stdout tail and parses the LAST line (round 4's 21 KB line was unreadable)


def compact_line(result):
    """The ONE JSON line printed on stdout: the contract's keys, `roofline` / `roofline_hbm` / `cpu_baseline` with numbers only,
    one (us_median, roof_frac_median, speedup) triple per sweep point and mode, the configs[3] sums.  Everything else (per-layer
    Llama table, probe rungs, notes, clocks' raw text, spreads) goes to the side file named in `detail`."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "timed_region_ms", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    out = {k: result[k] for k in keep if k in result}
    cfg = result.get("config", {})
    out["config"] = {"workload": "qqq_gemm per-channel sweep M in {1,16,128,1024,4096} N=8192 K=21760 (configs[1]); step = the 5 calls",
                     "weights": cfg.get("weights_short", "12 rotating 89 MB int4 buffers, GPTQ-style N(0,0.02^2)"),
                     "launch": cfg.get("launch_short", "eager"), "parallelism": cfg.get("parallelism_short", cfg.get("parallelism", ""))}
    for key in ("roofline", "roofline_hbm"):
        r = result.get(key)
        if r:
            out[key] = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "avg_launch_us")}
            if key == "roofline" and isinstance(r.get("sustained_on_random_int8"), dict):
                out[key]["sustained_16x16x64_tops"] = r["sustained_on_random_int8"].get("mfma_16x16x64_only_tops")
                out[key]["frac_of_sustained"] = r.get("frac_of_sustained")  # achieved / the same run's register-only 16x16x64 rate: separates box spread from kernel changes
    for key in ("per_m", "per_m_g128", "per_m_g128_expanded"):
        pm = result.get(key)
        if isinstance(pm, dict) and "error" not in pm:
            out[key] = {m: {"us": e.get("us_median"), "frac": e.get("roof_frac_median"), "roof": e.get("roof"), "kernel": e.get("kernel"),
                            "ksplit": e.get("ksplit"), "x_fp16": e.get("speedup_vs_fp16")} for m, e in pm.items()}
        elif pm:
            out[key] = pm
    ll = result.get("llama7b")
    if isinstance(ll, dict):
        if "error" in ll:
            out["llama7b"] = ll
        else:
            o = {}
            for blk, short in (("sum_of_7_linears", "sum7"), ("sum_of_4_merged_linears", "sum4_merged")):
                for mode, per in (ll.get(blk) or {}).items():
                    o.setdefault(short, {})[mode] = {m: {"us": e["quantlinear_us"], "x_fp16": e["speedup"]} for m, e in per.items()}
            q = {}
            for mode, layers in (ll.get("layers") or {}).items():
                e = layers.get("q_proj", {}).get("1024")
                if e:
                    q[mode] = {"quantlinear_us": e["quantlinear_us"], "gemm_only_us": e["gemm_only_us"], "fp16_us": e["fp16_linear_us"]}
            if q:
                o["q_proj_1024"] = q
            if ll.get("skipped"):
                o["skipped"] = len(ll["skipped"])
            out["llama7b"] = o
    cb = result.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind")}
        out["cpu_baseline"]["sample"] = "C oracle (OpenMP) M=64 N=8192 K=21760 per-channel, <= 12 s"
        out["cpu_baseline"]["cpu_model"] = cb.get("cpu_model")
        f = cb.get("torch_fp16_cpu_gemm", {})
        if "per_m" in f:
            out["cpu_baseline"]["torch_fp16_cpu_gemm_ms"] = {m: e["ms"] for m, e in f["per_m"].items()}
            out["cpu_baseline"]["torch_threads"] = f.get("threads")
    if "eager" in result:
        out["eager_ms_per_step"] = result["eager"]["ms_per_step"]
    if "step_us" in result:
        out["step_us_median"] = result["step_us"]["median"]
    ck = result.get("clocks")
    if isinstance(ck, dict):
        out["clocks_mhz"] = {"max": ck.get("max_mhz"), "busy": (ck.get("busy_sclk") or {}).get("mhz")}
    mg = result.get("multi_gpu")
    if mg:
        out["multi_gpu"] = mg
    for k in ("device", "detail"):
        if k in result:
            out[k] = result[k]
    out = _r(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:  # never let an optional block make the line unreadable again: drop them in order of dispensability
        for k in ("llama7b", "per_m_g128_expanded", "per_m_g128", "multi_gpu", "clocks_mhz", "per_m"):
            if k in out:
                out[k] = {"dropped": "see detail file"}
                line = json.dumps(out, separators=(",", ":"))
                if len(line) <= LINE_LIMIT:
                    break
    return line


def write_detail(result):
    """full result (every field of earlier rounds' line) -> gpurun_out/bench_detail.json (falls back to /tmp); returns the path"""
    for d in (os.path.join(ROOT, "gpurun_out"), "/tmp"):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, f"bench_detail_n{result.get('n_gpus', 1)}.json")
            with open(path, "w") as f:
                json.dump(result, f, indent=1)
            return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
        except OSError:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-fp16", action="store_true", help="skip the torch fp16 GPU GEMM comparison")
    ap.add_argument("--detail-iters", type=int, default=100)
    ap.add_argument("--steps-per-graph", type=int, default=0,
                    help="steps captured per hipGraph of the timed region (0 = 10; a shorter graph takes the remainder of --steps)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 PMC traffic passes; quote the committed ones")
    ap.add_argument("--no-llama", action="store_true", help="skip the BASELINE configs[3] block (Llama-2-7B linears)")
    ap.add_argument("--llama-budget", type=float, default=90.0, help="wall-clock bound of the configs[3] block, seconds")
    ap.add_argument("--check", action="store_true",
                    help="after the timed region verify the gathered outputs of the sharded points against a local full GEMM")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # QQQ_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, M-sharding, in-place all_gather_into_tensor on the side stream, --check) with whatever
    # world size the launcher gave -- ONE rank on a 1-GPU box: how RCCL, the side-stream gather and the CU-capped GEMM beside it are smoke-tested before the 8-GPU run
    is_multi = world > 1 or os.environ.get("QQQ_BENCH_FORCE_DIST", "0") == "1"
    bench_sms = int(os.environ.get("QQQ_BENCH_SMS", "-1"))  # the reference's `sms` for the GEMMs of the sharded points (a CU cap: leaves CUs to RCCL's kernels); -1 = all
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    # one process per GPU; the modulo only matters for the single-GPU functional test of the N>1 path
    # (QQQ_BENCH_BACKEND=gloo, two ranks sharing cuda:0) -- RCCL itself refuses two ranks on one device
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    if is_multi:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # (the forced one-rank run needs no launcher)
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("QQQ_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from qqq_amd import ops
    from qqq_amd.parallel import ShardedGemm, take_rows

    layer = Layer(dev, grouped=False)
    toks = {M: make_tokens(dev, M, M) for M in SWEEP_M}
    Dfull = {M: torch.empty((M, N_FULL), dtype=torch.float16, device=dev) for M in SWEEP_M}

    # ---- one step = the 5-call sweep ----
    # (single GPU: step t binds sweep point j to weight copy (5 t + j) % NBUF -- inside a multi-step graph as well as eagerly --, so a copy is
    #  re-read only after the other NBUF - 1 have passed through the Infinity Cache: what `config.weights` says)
    step_no = [0]
    if not is_multi:
        def step_body(t=None):
            if t is None:
                t = step_no[0]
                step_no[0] += 1
            for j, M in enumerate(SWEEP_M):
                A, s1 = toks[M]
                ops.qqq_gemm(A, layer.Bs[(len(SWEEP_M) * t + j) % NBUF], layer.C, Dfull[M], s1, layer.s2, layer.s3, layer.ws, -1, -1, -1, MAX_PAR)
    else:
        sharded = {}
        for j, M in enumerate(SWEEP_M):
            if M >= 64 * world:
                A, s1 = toks[M]
                Bj = layer.Bs[j % NBUF]

                def gemm_fn(a_rows, s1_rows, d_rows, Bj=Bj):
                    ops.qqq_gemm(a_rows, Bj, layer.C, d_rows, s1_rows, layer.s2, layer.s3, layer.ws, -1, -1, bench_sms, MAX_PAR)

                sg = ShardedGemm(gemm_fn, K=K_FULL)  # chunk count from the shard size and the layer (qqq_amd.parallel.pick_chunks)
                spans = sg.spans(M, N_FULL)
                sharded[M] = (sg, take_rows(A, spans), take_rows(s1, spans))

        def step_body(t=None):
            for j, M in enumerate(SWEEP_M):
                if M in sharded:
                    sg, a_loc, s1_loc = sharded[M]
                    sg(a_loc, s1_loc, M, N_FULL, Dfull[M], local=True)
                else:
                    A, s1 = toks[M]
                    ops.qqq_gemm(A, layer.Bs[j % NBUF], layer.C, Dfull[M], s1, layer.s2, layer.s3, layer.ws, -1, -1, -1, MAX_PAR)

    # launch-bound inner loop -> hipGraph (single GPU; collectives are left eager)
    step_body()
    torch.cuda.synchronize()
    graph = None
    by_size, spg = {}, 1  # graphs holding 1, (K - 1) % spg and spg consecutive steps: fewer replay boundaries in the timed region
    if not is_multi:
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step_body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step_body(0)
            g.replay()
            torch.cuda.synchronize()
            graph = g
            spg = min(10, args.steps) if args.steps_per_graph <= 0 else min(args.steps_per_graph, args.steps)
            if spg > 1:
                def capture(n):
                    gg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gg):
                        for i in range(n):
                            step_body(1 + i)  # (behind the one-step opener's copies 0 .. 4)
                    gg.replay()
                    torch.cuda.synchronize()
                    return gg
                # The timed region opens with the single-step graph `g` (graph_schedule): the GPU is idle behind the barrier until
                # the first graph has been launched and taken up; the larger graphs behind it are launched while the GPU works.
                sizes = set(graph_schedule(args.steps, spg))
                by_size = {1: g}
                for n in sorted(sizes - {1}):
                    by_size[n] = capture(n)
            else:
                spg = 1
        except Exception as e:  # pragma: no cover
            print(f"[bench] hipGraph capture unavailable ({e}); running eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    # N > 1: the per-rank step (local chunk GEMMs + the collectives on the side stream) is an eager Python loop by default.
    # QQQ_BENCH_NGRAPH=1 tries to capture it into a hipGraph (RCCL supports stream capture; gloo cannot be captured), falls
    # back to eager on any error and the JSON line says which one ran (`config.launch`, `multi_gpu.launch`).  Opt-in because a
    # capture that goes wrong inside the collective library can hang instead of raising, and no multi-GPU box was available to
    # try it on; the eager N = 1 figure (`eager`) is printed so that a 1 -> N curve can compare like with like either way.
    n_launch = "eager"
    if is_multi and os.environ.get("QQQ_BENCH_NGRAPH", "0") == "1" and os.environ.get("QQQ_BENCH_BACKEND", "nccl") == "nccl":
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step_body()
            g.replay()
            torch.cuda.synchronize()
            graph = g
            n_launch = "hipGraph (one step per graph, collectives captured)"
        except Exception as e:  # pragma: no cover
            print(f"[bench] rank {rank}: hipGraph capture of the sharded step failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
        if dist is not None:  # every rank must run the same way
            ok = torch.tensor([1 if graph is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                graph, n_launch = None, "eager (hipGraph capture failed on at least one rank)"

    def run_step():
        if graph is not None:
            graph.replay()
        else:
            step_body()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step()
    barrier()
    t0 = time.perf_counter()
    if graph is not None and spg > 1:
        for n in graph_schedule(args.steps, spg):   # EXACTLY args.steps steps
            by_size[n].replay()
    else:
        for _ in range(args.steps):
            run_step()
    barrier()
    dt = time.perf_counter() - t0
    # per-step spread, measured AFTER the timed region with stream events (diagnostic only; `value` uses `dt`)
    step_us = []
    if not is_multi:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 50))]
        for a, b in evs:
            a.record()
            run_step()
            b.record()
        torch.cuda.synchronize()
        step_us = [a.elapsed_time(b) * 1e3 for a, b in evs]
    eager = None
    if not is_multi and graph is not None:
        # the same K steps launched eagerly (Python loop, compiled binding): what an N > 1 run, whose step is eager, should be
        # compared with -- the graph figure above removes ~5 host calls per step that the N > 1 loop still pays
        for _ in range(3):
            step_body()
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(args.steps):
            step_body()
        torch.cuda.synchronize()
        de = time.perf_counter() - te
        eager = {"ms_per_step": de / args.steps * 1e3, "steps": args.steps}
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # N > 1: where the step's time goes at the sharded points: GEMM-only, all-gather-only, overlapped total;
    # event-timed on the launch stream after the timed region, max over ranks
    multi = None
    if is_multi:
        multi = {}
        for j, M in enumerate(SWEEP_M):
            if M not in sharded:
                continue
            sg, a_loc, s1_loc = sharded[M]
            res = {}
            for name, kw in (("gemm_only_us", dict(do_gather=False)), ("allgather_only_us", dict(do_gemm=False)), ("overlapped_us", {})):
                for _ in range(2):
                    sg(a_loc, s1_loc, M, N_FULL, Dfull[M], local=True, **kw)
                barrier()
                reps = 8
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    sg(a_loc, s1_loc, M, N_FULL, Dfull[M], local=True, **kw)
                e1.record()
                torch.cuda.synchronize()
                t = torch.tensor([e0.elapsed_time(e1) * 1e3 / reps], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                res[name] = float(t.item())
            from qqq_amd.parallel import pick_chunks

            res["chunks"] = pick_chunks(M, N_FULL, world, K_FULL)
            res["rows_per_rank"] = -(-M // world)
            multi[str(M)] = res

    if args.check and is_multi:
        for j, M in enumerate(SWEEP_M):
            if M in sharded:
                A, s1 = toks[M]
                ref = torch.empty((M, N_FULL), dtype=torch.float16, device=dev)
                ops.qqq_gemm(A, layer.Bs[j % NBUF], layer.C, ref, s1, layer.s2, layer.s3, layer.ws, -1, -1, -1, MAX_PAR)
                torch.cuda.synchronize()
                if not torch.equal(ref, Dfull[M]):
                    raise SystemExit(f"[bench --check] rank {rank}: gathered output differs from the local GEMM at M={M}")
        if rank == 0:
            print(f"[bench --check] sharded + all-gathered outputs identical to local full GEMMs for M in {sorted(sharded)}", file=sys.stderr)

    total_ops = sum(algorithmic_ops(M, N_FULL, K_FULL) for M in SWEEP_M)
    ms_per_step = dt / args.steps * 1e3
    value = total_ops * args.steps / dt / 1e12

    result = {
        "metric": "W4A8 GEMM TOPS + speedup vs fp16, M in {1..4096} N=8192 K=21760",
        "value": value, "unit": "TOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "timed_region_ms": dt * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "i8 x i4 -> i32 -> f16", "data": "synthetic",
        "config": {
            "workload": "qqq_gemm per-channel sweep M in {1,16,128,1024,4096}, N=8192, K=21760 (BASELINE configs[1]); one step = the 5 calls",
            "weights": f"{NBUF} rotating packed-int4 buffers of 89 MB (cold Infinity Cache); "
                       + ("W ~ N(0, 0.02^2) quantised GPTQ-style (SURVEY 8d)" if os.environ.get("QQQ_BENCH_WEIGHTS", "gptq") != "uniform"
                          else "uniformly random int4 codes"),
            "tokens": "x ~ N(0,1) fp16 through the fused dynamic int8 quantiser",
            "launch": (f"hipGraph replay: a one-step graph opens the timed region, then {spg} step(s) per graph (step t of a graph binds sweep point j to weight copy "
                       f"(5 t + j) % {NBUF}: a copy is re-read only after the other {NBUF - 1}, {(NBUF - 1) * 89} MB, have passed through the 256 MiB Infinity Cache)"
                       if not is_multi else "hipGraph replay, one step per graph (each sweep point bound to one of the weight copies)") if graph is not None else "eager",
            "weights_short": f"{NBUF} rotating 89 MB int4 buffers (cold Infinity Cache), " + ("GPTQ-style N(0,0.02^2)" if os.environ.get("QQQ_BENCH_WEIGHTS", "gptq") != "uniform" else "uniform int4 codes"),
            "launch_short": (f"hipGraph replay, 1-step opener then {spg} steps/graph" if graph is not None and not is_multi else n_launch),
            "parallelism_short": "single GPU" if not is_multi else f"M-sharded over {world} GPUs + RCCL all-gather (M >= {64*world})",
            "parallelism": "single GPU" if not is_multi else f"M-sharded over {world} GPUs + RCCL all-gather of fp16 shards (points with M >= {64*world}); step launch: {n_launch}",
        },
    }

    if eager is not None:
        eager["value"] = total_ops / (eager["ms_per_step"] * 1e-3) / 1e12
        eager["note"] = "N = 1 with the step launched eagerly instead of replayed from a hipGraph: the like-for-like base of an N > 1 point"
        result["eager"] = eager
    if multi is not None:
        # proof that the collective backend really spanned `world` ranks: an all_reduce of ones
        seen = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(seen)
        multi["world_seen"] = int(seen.item())
        multi["backend"] = dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")
        multi["launch"] = n_launch
        result["multi_gpu"] = multi
    if step_us:
        result["step_us"] = {"min": float(np.min(step_us)), "median": float(np.median(step_us)), "max": float(np.max(step_us)),
                             "n": len(step_us), "note": "event-timed replays after the timed region"}
    if rank == 0 and not is_multi:
        # ---- per-point detail, HIP event pairs around every launch (cold = rotating weights) ----
        per_m = {}
        it = args.detail_iters
        for M in SWEEP_M:
            A, s1 = toks[M]
            layer.time_calls(A, s1, Dfull[M], 3)
            cold = layer.time_calls(A, s1, Dfull[M], it, rotate=True) * 1e3
            warm = layer.time_calls(A, s1, Dfull[M], it, rotate=False) * 1e3
            # average launch duration; samples above 3x the median (a box hiccup: one preempted launch of 300 us among
            # a hundred 18-us ones moves the mean by 15 %) are dropped and counted
            keep = cold[cold <= 3.0 * np.median(cold)]
            us = float(np.mean(keep))
            from qqq_amd import _lib as _L

            pln = _L.plan(M, N_FULL, K_FULL, -1, MAX_PAR)
            entry = {
                "us": us, "us_mean_untrimmed": float(np.mean(cold)), "us_median": float(np.median(cold)), "us_min": float(np.min(cold)),
                "us_warm_l3": float(np.median(warm)),
                "outliers_dropped": int(len(cold) - len(keep)),
                "tops": algorithmic_ops(M, N_FULL, K_FULL) / us / 1e6,
                "gbs": algorithmic_bytes(M, N_FULL, K_FULL) / us / 1e3,
                "kernel": plan_label(pln), "ksplit": pln["ksplit"],
            }
            # the roof that binds this point: HBM below the ridge (~629 op/B), MFMA above
            hbm_us = algorithmic_bytes(M, N_FULL, K_FULL) / PEAK_HBM_GBS / 1e3
            mfma_us = algorithmic_ops(M, N_FULL, K_FULL) / PEAK_MFMA_TOPS / 1e6
            entry["roof"] = "hbm" if hbm_us >= mfma_us else "mfma"
            entry["roof_frac"] = max(hbm_us, mfma_us) / us
            entry["roof_frac_median"] = max(hbm_us, mfma_us) / entry["us_median"]
            if not args.no_fp16:
                f = fp16_gemm_us(dev, M)
                entry["fp16_gemm_us"] = f
                if M > 16:
                    v8 = int8_gemm_us(dev, M)
                    if v8:
                        entry["vendor_int8_gemm_us"] = v8
                        entry["speedup_vs_vendor_int8"] = v8 / entry["us"]
                entry["speedup_vs_fp16"] = f / us
            per_m[str(M)] = entry
        result["per_m"] = per_m
        live, traffic_note = ({}, "skipped (--no-pmc)") if args.no_pmc else live_traffic()
        a = per_m["4096"]
        sus = sustained_matrix_rate(dev)
        fam = FAMILY[_L.plan(4096, N_FULL, K_FULL, -1, MAX_PAR)["kernel"]]  # the family the dispatcher runs at M=4096 ("wide" since round 3; "panel" / "tiled" before)
        result["roofline"] = {
            "kernel": f"qqq_{fam}_kernel (M=4096)", "bound": "mfma",
            # average launch duration: the MEDIAN of the event-timed launches (a trimmed mean reads a few % better; both are in per_m)
            "achieved": algorithmic_ops(4096, N_FULL, K_FULL) / a["us_median"] / 1e6, "peak": PEAK_MFMA_TOPS,
            "unit": "TOPS", "frac": algorithmic_ops(4096, N_FULL, K_FULL) / a["us_median"] / 1e6 / PEAK_MFMA_TOPS,
            "frac_trimmed_mean": a["tops"] / PEAK_MFMA_TOPS,
            "traffic": live.get("tiled_m4096", committed_traffic(f"qqq_{fam}_kernel_M4096")),
            "traffic_unit": "bytes/launch, L2<->fabric (HBM + Infinity Cache)",
            "traffic_source": traffic_note if "tiled_m4096" in live else f"profiles/hbm_traffic.json (committed PMC pass; live pass: {traffic_note})",
            "algorithmic_bytes": algorithmic_bytes(4096, N_FULL, K_FULL), "avg_launch_us": a["us_median"],
            "frac_of_ubench_ceiling": a["tops"] / 4404.0,
            "sustained_on_random_int8": sus,
            # (the kernel issues v_mfma_i32_16x16x64_i8: the only probe rung that is a ceiling FOR IT is the register-only loop of that
            # instruction; the 32x32x32 rungs -- the round-1 tiled kernel's shapes -- are context, not ceilings, and carry no fraction)
            "frac_of_sustained_mfma_16x16x64_only": (a["tops"] / sus["mfma_16x16x64_only_tops"]) if "mfma_16x16x64_only_tops" in sus else None,
            "frac_of_sustained": (algorithmic_ops(4096, N_FULL, K_FULL) / a["us_median"] / 1e6 / sus["mfma_16x16x64_only_tops"]) if "mfma_16x16x64_only_tops" in sus else None,
            "note": "peak = 256 CU x 2.4 GHz x 8192 int8 op/clk; 4404 TOPS is the v_mfma_i32_32x32x32_i8 micro-benchmark ceiling; "
                    "under the M=4096 kernel the chip clocks ~2.06 GHz (power; `clocks`, profiles/r03_pmc_wide_m4096.txt); "
                    "sustained_on_random_int8 = this part's matrix pipe measured in this run on random operands: register-only loops of "
                    "32x32x32 / 16x16x64 MFMAs, and the round-1 tiled kernel's k-step (32x32x32 + LDS reads + int4 unpack) for context "
                    "(profiles/r02_mfma_power_ceiling.txt)",
        }
        h = per_m["1"]
        result["roofline_hbm"] = {
            "kernel": "qqq_column_kernel (M=1, decode)", "bound": "hbm", "achieved": algorithmic_bytes(1, N_FULL, K_FULL) / h["us_median"] / 1e3,
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": algorithmic_bytes(1, N_FULL, K_FULL) / h["us_median"] / 1e3 / PEAK_HBM_GBS,
            "frac_trimmed_mean": h["gbs"] / PEAK_HBM_GBS,
            "traffic": live.get("column_m1", committed_traffic("qqq_column_kernel_M1")),
            "traffic_unit": "bytes/launch, L2<->fabric (HBM + Infinity Cache)",
            "traffic_source": traffic_note if "column_m1" in live else f"profiles/hbm_traffic.json (committed PMC pass; live pass: {traffic_note})",
            "algorithmic_bytes": algorithmic_bytes(1, N_FULL, K_FULL), "avg_launch_us": h["us_median"],
            "frac_of_measured_read_ceiling": h["gbs"] / 6290.0,
            "note": "one launch per call (no split-K); the figure is the whole call incl. launch latency, cold Infinity Cache; "
                    "6.29 TB/s is the streaming-read ceiling measured on this part (MI355X_MICROARCH.md)",
        }
        # per-group (BASELINE configs[2]) detail
        try:
            del layer.Bs
            torch.cuda.empty_cache()
            lg = Layer(dev, grouped=True)
            pg = {}
            for M in SWEEP_M:
                A, s1 = toks[M]
                lg.time_calls(A, s1, Dfull[M], 3)
                cold = lg.time_calls(A, s1, Dfull[M], it, rotate=True) * 1e3
                us = float(np.mean(cold[cold <= 3.0 * np.median(cold)]))
                hbm_us = algorithmic_bytes(M, N_FULL, K_FULL, True) / PEAK_HBM_GBS / 1e3
                mfma_us = algorithmic_ops(M, N_FULL, K_FULL) / PEAK_MFMA_TOPS / 1e6
                pln = _L.plan(M, N_FULL, K_FULL, 128, MAX_PAR)
                pg[str(M)] = {"us": us, "us_median": float(np.median(cold)), "tops": algorithmic_ops(M, N_FULL, K_FULL) / us / 1e6,
                              "gbs": algorithmic_bytes(M, N_FULL, K_FULL, True) / us / 1e3,
                              "kernel": plan_label(pln), "ksplit": pln["ksplit"],
                              "roof": "hbm" if hbm_us >= mfma_us else "mfma", "roof_frac": max(hbm_us, mfma_us) / us,
                              "roof_frac_median": max(hbm_us, mfma_us) / float(np.median(cold))}
                if "fp16_gemm_us" in per_m[str(M)]:
                    pg[str(M)]["speedup_vs_fp16"] = per_m[str(M)]["fp16_gemm_us"] / us
            result["per_m_g128"] = pg
            # ... and the same layer with the opt-in load-time int8 expansion (QuantLinear.expand_for_prefill): the points whose plan reads it
            lg.expand()
            px = {}
            for M in SWEEP_M:
                pln = _L.plan(M, N_FULL, K_FULL, 128, MAX_PAR, tune=dict(w8=1))
                if not pln["w8"]:
                    continue
                A, s1 = toks[M]
                lg.time_calls(A, s1, Dfull[M], 3)
                cold = lg.time_calls(A, s1, Dfull[M], it, rotate=True) * 1e3
                med = float(np.median(cold))
                mfma_us = algorithmic_ops(M, N_FULL, K_FULL) / PEAK_MFMA_TOPS / 1e6
                px[str(M)] = {"us_median": med, "tops": algorithmic_ops(M, N_FULL, K_FULL) / med / 1e6, "kernel": plan_label(pln) + " (expanded int8 weights)",
                              "ksplit": pln["ksplit"], "roof": "mfma", "roof_frac_median": mfma_us / med, "vs_in_loop_requantiser": pg[str(M)]["us_median"] / med}
                if "fp16_gemm_us" in per_m[str(M)]:
                    px[str(M)]["speedup_vs_fp16"] = per_m[str(M)]["fp16_gemm_us"] / med
            result["per_m_g128_expanded"] = px
            lg.W8s = None
            torch.cuda.empty_cache()
        except Exception as e:  # pragma: no cover
            result.setdefault("per_m_g128", {"error": str(e)})
            result["per_m_g128_expanded"] = {"error": str(e)}
        # the clocks behind `roofline.frac`: rocminfo's maximum, and rocm-smi's reading with ~1 s of M=4096 launches queued
        try:
            A4, s14 = toks[4096]
            result["clocks"] = gpu_clocks(dev, busy_fn=lambda: [ops.qqq_gemm(A4, lg.Bs[i % NBUF], lg.C, Dfull[4096], s14, lg.s2, lg.s3, lg.ws, -1, -1, -1, MAX_PAR) for i in range(1800)])
        except Exception as e:  # pragma: no cover
            result["clocks"] = {"error": f"{type(e).__name__}: {e}"}
        # BASELINE configs[3]: the seven Llama-2-7B linears at batch 1 / 8 / 32 x seq 1024, both modes (tools/bench_llama.py: ONE
        # implementation for this block and for the stand-alone tool), bounded in wall-clock time
        if not args.no_llama:
            try:
                del lg
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_llama

                result["llama7b"] = bench_llama.llama_matrix(dev, budget_s=args.llama_budget)
            except Exception as e:  # pragma: no cover
                result["llama7b"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline()
            result["cpu_baseline"]["cpu_model"] = cpu_model_name()
        result["device"] = torch.cuda.get_device_name(dev)

    if rank == 0:
        result["detail"] = write_detail(result)
        sys.stdout.flush()
        print(compact_line(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
