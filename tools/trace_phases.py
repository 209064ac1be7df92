import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["QQQ_AMD_LIB"] = os.path.join(ROOT, "qqq_amd/libqqq_amd_trace.so")
import bench as Bn
from qqq_amd import ops
dev = torch.device("cuda:0")
layer = Bn.Layer(dev, grouped=(os.environ.get("MODE","pc")=="g128"), nbuf=1)
M = 4096
A, s1 = Bn.make_tokens(dev, M, M)
D = torch.empty((M, Bn.N_FULL), dtype=torch.float16, device=dev)
acc = torch.zeros((M, Bn.N_FULL), dtype=torch.int32, device=dev)
tune = eval(os.environ.get("TUNE", "dict(kernel=2,bm=256,glds=1,stages=5)"))
for _ in range(2):
    layer.C.zero_()
    ops.qqq_gemm_ex(A, layer.Bs[0], layer.C, D, s1, layer.s2, layer.s3, layer.ws, -1, -1, -1, 16, tune=tune)
    torch.cuda.synchronize()
raw = layer.C.flatten()[: 8 * 400 * 4].cpu().numpy().reshape(8, 400, 4)
for w in [int(x) for x in os.environ.get('WAVES', '0,4').split(',')]:
    tags = raw[w, :, 0]; t = (raw[w, :, 1].astype(np.int64) & 0xffffffff) | (raw[w, :, 2].astype(np.int64) << 32)
    n = int((t != 0).sum())
    # steady-state window: entries 120..(n)
    rows = [(int(tags[i]), int(t[i + 1] - t[i])) for i in range(100, min(n - 1, 390))]
    agg = {}
    for tag, d in rows: agg.setdefault(tag, []).append(d)
    print("wave", w, "n", n)
    for tag in sorted(agg): print(f"   after stamp {tag:3d}: mean {np.mean(agg[tag]):7.1f} cycles  (n={len(agg[tag])})")
    tot = sum(np.mean(v) for v in agg.values())
    print("   sum per 128-k block (4 k-steps):", tot, " (2048 = saturated matrix pipe)")
    if os.environ.get("DUMP") == "1":
        print("   raw:", [(int(tags[i]), int(t[i + 1] - t[i])) for i in range(200, 244)])
