#!/usr/bin/env python3
"""Full-size pins for BASELINE configs (N=8192, K=21760) without committing 89 MB tensors.

Runs only in the build container (needs /root/reference): the REFERENCE's own QuantLinear.pack() is run at full size
on inputs drawn from a generator this repository owns (numpy PCG64, tests/fullsize_inputs.py), and the SHA-256 digests
of its outputs `B`, `s_channel`, `s_group` -- and of the inputs, so that a changed random stream is recognised as
such -- are written to tests/golden/fullsize_pins.json.  tests/test_fullsize_pins_cpu.py regenerates the same inputs
and requires the product packer (qqq_amd.QuantLinear.pack, native qqq_pack_int4) to reproduce the digests; the GPU
test does the same with the on-device packer.

The file also carries the BASELINE configs[0] pin ("single qqq_gemm call M=16 N=8192 K=21760 per-channel, CPU
fake-quant reference"): digests of the oracle's int32 accumulators and fp16 outputs for that call, whose operands
(`B`, `s_channel` from the reference's pack(), `xq`, `s1` from the reference's dynamic_quant()) are reference-produced.

`sweep_per_channel` / `sweep_g128`: the same for BASELINE configs[1] / configs[2] -- M in {1, 16, 128, 1024, 4096} as row
prefixes of one 4096-token draw, activations quantised by the reference's dynamic_quant, per-prefix digests of the oracle's
accumulators and outputs (the -m gpu suite runs the HIP kernels on exactly these operands and compares digests).

usage: python tests/golden/gen_fullsize_pins.py   (~15 minutes on 8 cores: two 4096-token oracle GEMMs)
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    import fullsize_inputs as FI
    from gen_golden import load_reference_module
    from oracle import c_oracle as C

    ref = load_reference_module()
    torch.version.hip = None  # qlinear_marlin.py:56-59
    torch.cuda.get_device_capability = lambda *a, **k: (8, 0)  # :60-63
    out = {"N": FI.N_FULL, "K": FI.K_FULL, "generator": "numpy PCG64, seeds in tests/fullsize_inputs.py",
           "numpy": np.__version__, "torch": torch.__version__}
    for mode, gs in (("per_channel", -1), ("g128", 128)):
        W_fq, scale, s_extra = FI.layer_inputs(gs)
        ent = {"in_W_fq": sha(W_fq), "in_scale": sha(scale)}
        if s_extra is not None:
            ent["in_s_extra"] = sha(s_extra)
        lin = torch.nn.Linear(FI.K_FULL, FI.N_FULL, bias=False).half()
        lin.weight.data = torch.from_numpy(W_fq)
        ql = ref.QuantLinear(4, gs, FI.K_FULL, FI.N_FULL, bias=False)
        ql.pack(lin, torch.from_numpy(scale), None if s_extra is None else torch.from_numpy(s_extra))
        B, s_channel, s_group = ql.B.numpy(), ql.s_channel.numpy(), ql.s_group.numpy()
        ent.update(ref_B=sha(B), ref_s_channel=sha(s_channel), ref_s_group=sha(s_group))
        out[mode] = ent
        print(mode, ent)
        if gs == -1:
            # BASELINE configs[0]: one M=16 call on reference-produced operands
            x = FI.c0_tokens()
            xq, s1 = ql.dynamic_quant(torch.from_numpy(x))
            xq, s1 = xq.numpy(), s1.numpy()
            D, acc = C.qqq_gemm(xq, B, s1, s_channel, None, return_acc=True)
            Dfq = (xq.astype(np.float32) * s1) @ W_fq.astype(np.float32).T
            err = float(np.abs(D.astype(np.float32) - Dfq).max())
            out["config0"] = {"M": 16, "in_x": sha(x), "ref_xq": sha(xq), "ref_s1": sha(s1), "oracle_acc": sha(acc),
                              "oracle_D": sha(D.view(np.uint16)), "max_abs_D": float(np.abs(D.astype(np.float32)).max()),
                              "max_abs_err_vs_fakequant": err}
            print("config0", out["config0"])
        # BASELINE configs[1] / configs[2] on reference-made operands: the sweep's token counts (row prefixes of one
        # 4096-token draw; rows are independent), activations quantised by the REFERENCE's dynamic_quant, weights
        # from the REFERENCE's pack(); digests of the oracle's accumulators / outputs per prefix.  The -m gpu test
        # test_gpu_parity.py::test_baseline_sweep_on_reference_operands_pinned holds the HIP kernels to them.
        xs = FI.sweep_tokens()
        xq, s1 = ql.dynamic_quant(torch.from_numpy(xs))
        xq, s1 = xq.numpy(), s1.numpy()
        D, acc = C.qqq_gemm(xq, B, s1, s_channel, s_group if gs != -1 else None, return_acc=True)
        sw = {"in_x": sha(xs), "ref_xq": sha(xq), "ref_s1": sha(s1), "Ms": list(FI.SWEEP_MS)}
        for M in FI.SWEEP_MS:
            sw[f"oracle_acc_m{M}"] = sha(acc[:M])
            sw[f"oracle_D_m{M}"] = sha(D[:M].view(np.uint16))
        out["sweep_" + mode] = sw
        print("sweep", mode, sw)
        del ql, lin, W_fq
    path = os.path.join(HERE, "fullsize_pins.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
