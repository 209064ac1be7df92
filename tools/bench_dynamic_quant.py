#!/usr/bin/env python3
"""Times qqq_dynamic_quant (eager calls, HIP events) and checks it bit-for-bit against the torch-GPU evaluation of the
reference expression (qlinear_marlin.py:265-268)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qqq_amd import dynamic_quant
dev = torch.device("cuda:0")
for (M, K) in ((4096, 21760), (32768, 4096), (8192, 11008), (1024, 4096), (16, 21760)):
    x = torch.randn((M, K), device=dev, dtype=torch.float16)
    for _ in range(3): dynamic_quant(x)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); dynamic_quant(x); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)[5] * 1e3
    xq, s1 = dynamic_quant(x)
    s = (x.abs().amax(dim=-1, keepdim=True).to(torch.float32) * (1.0 / 127.0)).half().float()  # torch-GPU lowering of .div(127.0)
    ref = (x.float() / s).round().clamp(-128, 127).to(torch.int8)
    print(f"M={M} K={K}: {t:.1f} us  {(M*K*3)/t/1e3:.0f} GB/s  mismatches xq={(ref != xq).sum().item()} s1={(s != s1).sum().item()}")
