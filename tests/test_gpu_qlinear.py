"""QuantLinear (module level) and the fused dynamic_quant kernel on the GPU."""
import numpy as np
import pytest
import torch

from gpu_util import ulp_distance

pytestmark = pytest.mark.gpu


def _torch_dynamic_quant(x):
    # the reference's expression, verbatim semantics (qlinear_marlin.py:265-268), evaluated by torch on the GPU
    quant_scale = x.abs().max(dim=-1, keepdim=True)[0].div(127.0).to(torch.float32)
    xq = (x / quant_scale).round().clamp(-128, 127).to(torch.int8)
    return xq, quant_scale


def test_fused_dynamic_quant(dev):
    from oracle import c_oracle as C
    from qqq_amd import dynamic_quant

    g = torch.Generator(device="cpu").manual_seed(0)
    for (M, K) in ((1, 256), (7, 4096), (64, 21760), (33, 11008), (3, 65536)):
        x = (torch.randn((M, K), generator=g) * 1.7).to(torch.float16)
        x[M // 2, K // 3] = 31.5
        xq, s1 = dynamic_quant(x.to(dev))
        torch.cuda.synchronize()
        oq, os1 = C.dynamic_quant(x.numpy(), "recip")
        assert np.array_equal(s1.cpu().numpy().view(np.uint32), os1.view(np.uint32)), (M, K)
        assert np.array_equal(xq.cpu().numpy(), oq), (M, K)
        tq, ts1 = _torch_dynamic_quant(x.to(dev))
        n_s = int((ts1 != s1).sum().item())
        n_q = int((tq != xq).sum().item())
        print(f"dynamic_quant M={M} K={K}: differs from torch-GPU ops in {n_s} scales / {n_q} values")
        assert n_s == 0 and n_q == 0, "fused kernel must equal the reference expression as torch evaluates it on this GPU"


@pytest.mark.parametrize("group_size", [-1, 128])
def test_quantlinear_forward(golden, dev, group_size):
    from oracle import c_oracle as C
    from qqq_amd import QuantLinear

    tag = f"g{group_size}_n128_k256"
    W = golden[f"{tag}/W_fq"]
    N, K = W.shape
    lin = torch.nn.Linear(K, N, bias=True).half()
    lin.weight.data = torch.from_numpy(W.copy())
    lin.bias.data = torch.from_numpy(golden[f"{tag}/bias"].copy())
    ql = QuantLinear(4, group_size, K, N, bias=True)
    se = torch.from_numpy(golden[f"{tag}/s_extra"].copy()) if group_size != -1 else None
    ql.pack(lin, torch.from_numpy(golden[f"{tag}/scale"].copy()), se)
    ql = ql.to(dev)
    for M in golden[f"{tag}/Ms"]:
        x = golden[f"{tag}/m{M}/x"]
        y = ql(torch.from_numpy(x.copy()).to(dev).reshape(1, int(M), K))
        torch.cuda.synchronize()
        assert y.shape == (1, int(M), N) and y.dtype == torch.float16
        xq, s1 = C.dynamic_quant(x, "recip")
        D = C.qqq_gemm(xq, golden[f"{tag}/ref_B"], s1, golden[f"{tag}/ref_s_channel"], golden[f"{tag}/ref_s_group"])
        exp = (torch.from_numpy(D.copy()) + torch.from_numpy(golden[f"{tag}/bias"].copy())).numpy()  # fp16 add, as the reference
        assert ulp_distance(y.cpu().numpy().reshape(int(M), N), exp) == 0


def test_state_dict_roundtrip_and_vllm_wrapper(golden, dev):
    from qqq_amd import QuantLinear, marlin_qqq_gemm

    tag = "g128_n256_k256"
    ql = QuantLinear(4, 128, 256, 256, bias=False)
    sd = {"B": torch.from_numpy(golden[f"{tag}/ref_B"].copy()), "s_channel": torch.from_numpy(golden[f"{tag}/ref_s_channel"].copy()),
          "s_group": torch.from_numpy(golden[f"{tag}/ref_s_group"].copy())}
    ql.load_state_dict(sd)  # a QQQ checkpoint's keys load unchanged
    ql = ql.to(dev)
    M = 16
    xq = torch.from_numpy(golden[f"{tag}/m{M}/ref_xq"].copy()).to(dev)
    s1 = torch.from_numpy(golden[f"{tag}/m{M}/ref_s1"].copy()).to(dev)
    D = marlin_qqq_gemm(xq, ql.B, s1, ql.s_channel, ql.s_group, ql.workspace, M, 256, 256)
    torch.cuda.synchronize()
    assert ulp_distance(D.cpu().numpy(), golden[f"{tag}/m{M}/oracle_D"]) == 0


def test_custom_op_traces_under_torch_compile(golden, dev):
    """north_star: `qqq_gemm` is a real torch custom op (torch.library), so a compiled graph keeps it as one
    node instead of graph-breaking on a pybind function.  aot_eager avoids needing a codegen backend."""
    import torch._dynamo
    from qqq_amd import QuantLinear

    tag = "g-1_n128_k256"
    ql = QuantLinear(4, -1, 256, 128, bias=True)
    ql.load_state_dict({"B": torch.from_numpy(golden[f"{tag}/ref_B"].copy()),
                        "s_channel": torch.from_numpy(golden[f"{tag}/ref_s_channel"].copy()),
                        "s_group": torch.empty(0, dtype=torch.float16),
                        "bias": torch.from_numpy(golden[f"{tag}/bias"].copy())})
    ql = ql.to(dev)
    x = torch.from_numpy(golden[f"{tag}/m16/x"].copy()).to(dev)
    eager = ql(x)
    torch._dynamo.reset()
    compiled = torch.compile(ql, backend="aot_eager", fullgraph=True)
    out = compiled(x)
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


def test_fused_projections_match_separate_layers(golden, dev):
    from qqq_amd import QuantLinear, fuse_quant_linears

    for gs, tags in ((-1, ("g-1_n128_k256", "g-1_n256_k256")), (128, ("g128_n128_k256", "g128_n256_k256"))):
        parts = []
        for tag in tags:
            N, K = golden[f"{tag}/W_fq"].shape
            ql = QuantLinear(4, gs, K, N, bias=True)
            sd = {"B": torch.from_numpy(golden[f"{tag}/ref_B"].copy()), "s_channel": torch.from_numpy(golden[f"{tag}/ref_s_channel"].copy()),
                  "s_group": torch.from_numpy(golden[f"{tag}/ref_s_group"].copy()), "bias": torch.from_numpy(golden[f"{tag}/bias"].copy())}
            ql.load_state_dict(sd)
            parts.append(ql.to(dev))
        fused = fuse_quant_linears(parts)
        x = torch.from_numpy(golden[f"{tags[0]}/m16/x"].copy()).to(dev)
        sep = torch.cat([p(x) for p in parts], dim=-1)
        out = fused(x)
        torch.cuda.synchronize()
        assert torch.equal(out, sep)
