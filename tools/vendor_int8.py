#!/usr/bin/env python3
"""Vendor reference point: torch._int_mm (hipBLASLt int8 x int8 -> int32) and torch fp16 matmul on the BASELINE
shape, timed like the W4A8 kernel (HIP events, hipGraph-free eager calls, median of 20).  int8 weights are 2x the
bytes of the packed int4 ones and no scales / fp16 epilogue are applied: it is the matrix-pipe ceiling a library
kernel reaches under the same power limit, not an equivalent operator."""
import numpy as np, torch
dev = torch.device("cuda:0")
N, K = 8192, 21760
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]) * 1e3)
for M in (32, 128, 1024, 4096, 8192):
    A = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    Bs = [torch.randint(-128, 128, (K, N), dtype=torch.int8, device=dev) for _ in range(3)]
    i = [0]
    def f():
        i[0] += 1
        return torch._int_mm(A, Bs[i[0] % 3])
    try:
        us = t(f)
        print(f"M={M}: torch._int_mm {us:8.1f} us  {2*M*N*K/us/1e6:7.0f} TOPS")
    except Exception as e:
        print(f"M={M}: torch._int_mm failed: {e}")
    Ah = torch.randn((M, K), dtype=torch.float16, device=dev)
    Wh = [torch.randn((K, N), dtype=torch.float16, device=dev) * 0.02 for _ in range(2)]
    def fh():
        i[0] += 1
        return Ah @ Wh[i[0] % 2]
    us = t(fh)
    print(f"M={M}: torch fp16 matmul {us:8.1f} us  {2*M*N*K/us/1e6:7.0f} TFLOPS")
