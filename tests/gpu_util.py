"""Shared helpers for the -m gpu parity tests (call the product through its C-ABI / operator layer)."""
import numpy as np
import torch

from qqq_amd import ops


def ulp_distance(a: np.ndarray, b: np.ndarray) -> int:
    """max distance in fp16 ulps between two fp16 arrays (monotone integer mapping)."""
    def key(x):
        u = x.view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, 0x8000 - u, u)
    if a.size == 0:
        return 0
    return int(np.abs(key(a) - key(b)).max())


class GemmHarness:
    """Owns the device buffers of one layer (as QuantLinear would) and runs qqq_gemm variants."""

    def __init__(self, B, s2, s3, dev, max_par=16):
        self.dev = dev
        def dv(t, dtype=None):
            if isinstance(t, np.ndarray):
                t = torch.from_numpy(np.ascontiguousarray(t, dtype=dtype))
            return t.to(dev).contiguous()

        self.B = dv(B)
        self.s2 = dv(s2, np.float32)
        if s3 is None or (isinstance(s3, np.ndarray) and s3.size == 0) or (torch.is_tensor(s3) and s3.numel() == 0):
            self.s3 = torch.empty(0, dtype=torch.float16, device=dev)
        else:
            self.s3 = dv(s3)
        self.N = self.B.shape[1] // 2
        self.K = self.B.shape[0] * 16
        self.max_par = max_par
        # poison the scratch: the kernels must not depend on its previous contents
        self.C = torch.full((max_par * 64, self.N), 0x7B7B7B7B, dtype=torch.int32, device=dev)
        self.ws = torch.zeros(max(self.N // 128, 1) * max_par, dtype=torch.int32, device=dev)
        self.W8 = None

    def expand(self):
        """opt-in load-time re-layout (per-group layers): from here on every run() hands the expanded int8 weights to the call"""
        self.W8 = ops.expand_int8(self.B, self.s3)  # (s3 empty: a per-channel layer)
        return self

    def run(self, A, s1, tune=None, want_acc=True, bias=None):
        A = (torch.from_numpy(np.ascontiguousarray(A)) if isinstance(A, np.ndarray) else A).to(self.dev).contiguous()
        s1 = (torch.from_numpy(np.ascontiguousarray(s1, dtype=np.float32)) if isinstance(s1, np.ndarray) else s1).to(self.dev).contiguous()
        M = A.shape[0]
        D = torch.full((M, self.N), float("nan"), dtype=torch.float16, device=self.dev)
        acc = torch.full((M, self.N), -1, dtype=torch.int32, device=self.dev) if want_acc else None
        if bias is not None:
            bias = (torch.from_numpy(np.ascontiguousarray(bias)) if isinstance(bias, np.ndarray) else bias).to(self.dev)
        ops.qqq_gemm_ex(A, self.B, self.C, D, s1, self.s2, self.s3, self.ws, -1, -1, -1, self.max_par,
                        tune=tune, acc_out=acc, bias=bias, W8=self.W8)
        torch.cuda.synchronize()
        assert int(self.ws.abs().sum().item()) == 0, "workspace must be all-zero on return"
        return D.cpu().numpy(), (acc.cpu().numpy() if want_acc else None)


def variants(M, K, N):
    """tuning variants that are valid for this problem"""
    v = [dict(kernel=1, ksplit=1, waves=4), dict(kernel=1, ksplit=1, waves=8)]
    if K // 64 >= 4:
        v += [dict(kernel=1, ksplit=2, waves=4, fused=1), dict(kernel=1, ksplit=2, waves=4, fused=2), dict(kernel=1, ksplit=2, waves=8, fused=3)]
    if K // 64 >= 8:
        v += [dict(kernel=1, ksplit=3, waves=8, fused=1), dict(kernel=1, ksplit=4, waves=4, fused=3)]
    if K // 64 >= 64:  # round 5: in-launch split-K through arrival-order slots (fused=3) with uneven slices (skew in 64-k steps; -1 = even)
        v += [dict(kernel=1, ksplit=4, waves=8, fused=3, skew=5), dict(kernel=1, ksplit=3, waves=4, fused=3, skew=-1), dict(kernel=1, mt=4, ksplit=2, fused=3, skew=200),
              dict(kernel=1, ksplit=7, waves=4, fused=3, skew=1)]
    if M <= 16:
        v += [dict(kernel=1, ksplit=1, waves=16)]
    v += [dict(kernel=1)]  # auto split
    if N % 64 == 0:  # column kernel (decode): 32 columns x all of K per workgroup
        v += [dict(kernel=3), dict(kernel=3, mt=1, pf=4), dict(kernel=3, mt=2, pf=4), dict(kernel=3, mt=1, pf=12, ksplit=2), dict(kernel=3, mt=1, waves=16), dict(kernel=3, mt=1, waves=16, ksplit=2)]
        if K // 64 >= 3:
            v += [dict(kernel=3, mt=2, pf=8, ksplit=3), dict(kernel=3, mt=1, pf=2), dict(kernel=3, mt=1, pf=6)]
    if N % 64 == 0:  # panel kernel: all tokens of an m-block x 128 / 256 columns x a K slice, in-launch split-K
        nst = (K // 64 + 1) // 2
        for mt in (1, 2, 4, 8):
            v.append(dict(kernel=4, mt=mt, ksplit=1))
        v += [dict(kernel=4), dict(kernel=4, bm=256), dict(kernel=4, waves=4), dict(kernel=4, pf=2), dict(kernel=4, bm=256, pf=2, mt=2),
              dict(kernel=4, pf=8, mt=1), dict(kernel=4, pf=3), dict(kernel=4, pf=4, stages=2, mt=4),
              dict(kernel=4, bm=256, mt=8, pw=2), dict(kernel=4, bm=256, mt=8, pw=2, pf=3), dict(kernel=4, bm=256, mt=8, pw=2, pf=4, stages=4)]
        if nst >= 2:
            v += [dict(kernel=4, ksplit=2), dict(kernel=4, bm=256, ksplit=2, mt=4), dict(kernel=4, waves=4, ksplit=2, pf=2),
                  dict(kernel=4, bm=256, mt=8, pw=2, ksplit=2), dict(kernel=4, bm=256, mt=8, pw=2, stages=4, ksplit=2)]
        if nst >= 3:
            v += [dict(kernel=4, ksplit=3, mt=1), dict(kernel=4, bm=256, ksplit=3), dict(kernel=4, bm=256, mt=8, pw=2, pf=3, ksplit=3)]
        # round 5: uneven K slices (the host keeps 4 stages per slice; -1 = even slices, 0 = the automatic skew)
        if nst >= 9:
            v += [dict(kernel=4, ksplit=2, skew=1), dict(kernel=4, mt=4, ksplit=2, skew=63), dict(kernel=4, ksplit=2, skew=-1)]
        if nst >= 18:
            v += [dict(kernel=4, ksplit=4, skew=2), dict(kernel=4, bm=256, ksplit=3, skew=5), dict(kernel=4, bm=256, mt=8, pw=2, ksplit=2, skew=3)]
    if N % 64 == 0:  # wide kernel: 256 tokens x 256 columns per workgroup, four 512-register waves, no split-K
        v += [dict(kernel=5), dict(kernel=5, pf=8), dict(kernel=5, pw=4), dict(kernel=5, mt=8), dict(kernel=5, mt=8, pf=4),
              dict(kernel=5, bm=128), dict(kernel=5, bm=128, pf=8, pw=16)]
        if K % 128 == 0 and K // 128 >= 8:  # in-launch split-K: row-major partial tiles in C, tickets in workspace
            v += [dict(kernel=5, ksplit=2), dict(kernel=5, mt=8, ksplit=2)]
        if K % 128 == 0 and K // 128 >= 12:
            v += [dict(kernel=5, ksplit=3), dict(kernel=5, mt=8, ksplit=3, pf=8), dict(kernel=5, ksplit=2, skew=3), dict(kernel=5, bm=128, ksplit=3, skew=1)]
    if K % 128 == 0:
        for bm in (64, 128, 256):
            v.append(dict(kernel=2, bm=bm, glds=2, ksplit=1))
            for stages in (2, 3, 4) + ((5, 6, 7) if bm == 256 else ()):
                v.append(dict(kernel=2, bm=bm, glds=1, stages=stages, ksplit=1))
        for bm in (258, 259, 130, 131):  # other wave shapes of the 256- and 128-row tiles
            for stages in (2, 3, 5):
                v.append(dict(kernel=2, bm=bm, glds=1, stages=stages, ksplit=1))
        if K // 128 >= 2:
            v.append(dict(kernel=2, bm=258, glds=1, stages=5, ksplit=2))
            v.append(dict(kernel=2, bm=256, glds=1, stages=5, ksplit=2))
            v.append(dict(kernel=2, bm=256, glds=1, stages=6, ksplit=2))
            v.append(dict(kernel=2, bm=256, glds=1, stages=7, ksplit=2))
            v.append(dict(kernel=2, bm=128, glds=1, stages=3, ksplit=2))
            v.append(dict(kernel=2, bm=64, glds=2, ksplit=2))
            # tiled split-K: fused=1 in-launch (slots + tickets; also the default), fused=2 slabs + reduce launch
            v.append(dict(kernel=2, bm=256, glds=1, stages=5, ksplit=2, fused=2))
            v.append(dict(kernel=2, bm=131, glds=2, ksplit=2, fused=1))
            v.append(dict(kernel=2, bm=64, glds=2, ksplit=2, fused=2))
        if K // 128 >= 4:
            v.append(dict(kernel=2, bm=64, glds=1, stages=4, ksplit=3))
            v.append(dict(kernel=2, bm=130, glds=1, stages=4, ksplit=4, fused=1))
            v.append(dict(kernel=2, bm=64, glds=1, stages=4, ksplit=3, fused=2))
    v.append(dict())  # fully automatic (== qqq_w4a8_gemm)
    return v
