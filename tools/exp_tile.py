import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench as Bn
dev = torch.device("cuda:0")
layer = Bn.Layer(dev, grouped=False, nbuf=2)
tunes = eval(os.environ.get("TUNES", "[dict(kernel=2, bm=256, glds=1, stages=2)]"))
for M in (4096,):
    A, s1 = Bn.make_tokens(dev, M, M)
    D = torch.empty((M, Bn.N_FULL), dtype=torch.float16, device=dev)
    for tune in tunes:
        layer.time_calls(A, s1, D, 3, tune=tune)
        t = layer.time_calls(A, s1, D, 8, tune=tune) * 1e3
        print(f"{os.environ.get('QQQ_AMD_LIB','default')[-12:]} M={M} {tune}: {t.mean():.1f} us (min {t.min():.1f}) {Bn.algorithmic_ops(M,Bn.N_FULL,Bn.K_FULL)/t.mean()/1e6:.0f} TOPS")
