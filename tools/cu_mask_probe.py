#!/usr/bin/env python3
"""Which (XCD, CU) does bit i of a hipExtStreamCreateWithCUMask mask enable?  (the `sms` cap assumed neighbouring bits are
neighbouring CUs of one XCD.)  One temporary stream per probe mask, 2048 one-wave workgroups that hold their CU for 20 us, {XCC_ID, HW_ID} each.
usage: python tools/cu_mask_probe.py            (GPU box)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qqq_amd import _dev

DEV = _dev.lib()
dev = torch.device("cuda:0")


def where(bits, nwg=2048, hold=20):
    """set of (xcc, se, sh?, cu) the workgroups of a stream masked to `bits` ran on"""
    words = 8
    mask = (ctypes.c_uint32 * words)()
    for b in bits:
        mask[b >> 5] |= 1 << (b & 31)
    out = torch.zeros(2 * nwg, dtype=torch.int32, device=dev)
    rc = DEV.qqq_dev_probe_placement(mask, words, nwg, hold, out.data_ptr(), 0, None)
    assert rc == 0, _dev.last_error()
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype(np.uint32).reshape(nwg, 2)
    xcc = o[:, 0] & 15
    hw = o[:, 1]
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7  # gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    return sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))


def per_xcc(w):
    per = {}
    for x, se, sh, cu in w:
        per[x] = per.get(x, 0) + 1
    return dict(sorted(per.items()))


if __name__ == "__main__":
    print("single bits (64 workgroups): the XCC whose workgroups all landed on ONE CU is the one the bit belongs to; the other XCCs are unrestricted")
    for b in list(range(0, 18)) + [31, 32, 63, 64, 127, 128, 255]:
        w = where([b], nwg=64, hold=5)
        per = per_xcc(w)
        one = [x for x, n in per.items() if n == 1]
        cu = [(x, se, sh, c) for x, se, sh, c in w if x in one]
        print(f"  bit {b:3d}: restricted XCC {one} -> (xcc, se, sh, cu) {cu};  CUs seen per XCC {per}")
    for name, bits in (("first 8 bits", range(8)), ("first 32 bits", range(32)), ("first 64 bits", range(64)), ("first 100 bits", range(100)), ("every 8th bit (all of XCC 0)", range(0, 256, 8)),
                       ("every 4th bit from 3 (round 5's sms = 64)", range(3, 256, 4)), ("all 256", range(256))):
        w = where(list(bits))
        print(f"{name}: {len(w)} distinct CUs; per XCC {per_xcc(w)}")
