"""Product packer / QuantLinear host logic against the reference's own pack() outputs (goldens) -- CPU."""
import numpy as np
import pytest
import torch

from qqq_amd import QuantLinear, pack as P


def _cases(golden):
    for tag in golden["cases"]:
        tag = str(tag)
        yield tag, int(tag.split("_")[0][1:])


def test_pack_codes_matches_reference_layout(golden):
    from oracle import qqq_ref as R

    for tag, _ in _cases(golden):
        B = golden[f"{tag}/ref_B"]
        grouped = golden[f"{tag}/ref_s_group"].size > 0
        codes = R.unpack_codes(B, grouped)
        assert np.array_equal(P.pack_codes(torch.from_numpy(codes), grouped).numpy(), B)
        assert np.array_equal(P.unpack_codes(torch.from_numpy(B), grouped).numpy(), codes)


def test_quantlinear_pack_matches_reference_pack(golden):
    for tag, gs in _cases(golden):
        W = golden[f"{tag}/W_fq"]
        N, K = W.shape
        lin = torch.nn.Linear(K, N, bias=True).half()
        lin.weight.data = torch.from_numpy(W.copy())
        lin.bias.data = torch.from_numpy(golden[f"{tag}/bias"].copy())
        ql = QuantLinear(4, gs, K, N, bias=True)
        se = torch.from_numpy(golden[f"{tag}/s_extra"].copy()) if f"{tag}/s_extra" in golden else None
        ql.pack(lin, torch.from_numpy(golden[f"{tag}/scale"].copy()), se)
        assert np.array_equal(ql.B.numpy(), golden[f"{tag}/ref_B"]), tag
        assert np.array_equal(ql.s_channel.numpy().view(np.uint32), golden[f"{tag}/ref_s_channel"].view(np.uint32)), tag
        assert np.array_equal(ql.s_group.numpy().view(np.uint16), golden[f"{tag}/ref_s_group"].view(np.uint16)), tag
        assert np.array_equal(ql.bias.numpy().view(np.uint16), golden[f"{tag}/bias"].view(np.uint16))


def test_state_dict_contract():
    """Buffer names / shapes / dtypes / persistence of the reference module (qlinear_marlin.py:97-138)."""
    ql = QuantLinear(4, 128, 256, 512, bias=True)
    sd = ql.state_dict()
    assert set(sd) == {"B", "s_channel", "s_group", "bias"}
    assert sd["B"].shape == (16, 1024) and sd["B"].dtype == torch.int32
    assert sd["s_channel"].shape == (1, 512) and sd["s_channel"].dtype == torch.float32
    assert sd["s_group"].shape == (2, 512) and sd["s_group"].dtype == torch.float16
    assert ql.workspace.shape == (512 // 128 * 16,) and ql.workspace.dtype == torch.int32
    assert ql.reduce_buffer.shape == (1024, 512) and ql.reduce_buffer.dtype == torch.int32
    ql2 = QuantLinear(4, -1, 256, 512, bias=False)
    assert set(ql2.state_dict()) == {"B", "s_channel", "s_group"} and ql2.s_group.numel() == 0
    ql.half()
    assert ql.s_channel.dtype == torch.float32 and ql.s_group.dtype == torch.float16  # _apply pin (:141-145)


def test_constructor_rejections():
    with pytest.raises(ValueError):
        QuantLinear(4, -1, 100, 256, bias=False)
    with pytest.raises(NotImplementedError):
        QuantLinear(8, -1, 256, 256, bias=False)
    with pytest.raises(ValueError):
        QuantLinear(4, 64, 256, 256, bias=False)
    with pytest.raises(NotImplementedError):
        QuantLinear(4, -1, 256, 256, bias=False, trainable=True)


def test_ops_refuse_cpu_tensors():
    """There is no CPU path: calling the operator with CPU tensors must fail loudly, not fall back."""
    from qqq_amd import qqq_gemm

    A = torch.zeros((16, 256), dtype=torch.int8)
    B = torch.zeros((16, 512), dtype=torch.int32)
    C = torch.zeros((1024, 256), dtype=torch.int32)
    D = torch.zeros((16, 256), dtype=torch.float16)
    s1 = torch.ones((16, 1)); s2 = torch.ones((1, 256)); s3 = torch.zeros(0, dtype=torch.float16)
    ws = torch.zeros(32, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="no CPU path"):
        qqq_gemm(A, B, C, D, s1, s2, s3, ws, -1, -1, -1, 16)
    # reference-side checks keep their messages (csrc/qqq_gemm.cu:1066-1075)
    with pytest.raises(RuntimeError, match="s1 dtype must be float32"):
        qqq_gemm(A, B, C, D, s1.half(), s2, s3, ws, -1, -1, -1, 16)
    with pytest.raises(RuntimeError, match="workspace must be of size at least"):
        qqq_gemm(A, B, C, D, s1, s2, s3, ws[:1], -1, -1, -1, 16)
    with pytest.raises(RuntimeError, match="not compatible with 3 groups"):
        qqq_gemm(A, B, C, D, s1, s2, torch.zeros((3, 256), dtype=torch.float16), ws, -1, -1, -1, 16)


def test_fuse_quant_linears_equals_packing_the_concatenated_weight(golden):
    """q/k/v-style fusion: concatenating stored tensors == pack() of the concatenated layer."""
    from qqq_amd import fuse_quant_linears

    for gs, tags in ((-1, ("g-1_n128_k256", "g-1_n256_k256")), (128, ("g128_n128_k256", "g128_n256_k256"))):
        parts, Ws, scs, ses, biases = [], [], [], [], []
        for tag in tags:
            W = golden[f"{tag}/W_fq"]
            N, K = W.shape
            lin = torch.nn.Linear(K, N, bias=True).half()
            lin.weight.data = torch.from_numpy(W.copy())
            lin.bias.data = torch.from_numpy(golden[f"{tag}/bias"].copy())
            ql = QuantLinear(4, gs, K, N, bias=True)
            se = torch.from_numpy(golden[f"{tag}/s_extra"].copy()) if gs != -1 else None
            ql.pack(lin, torch.from_numpy(golden[f"{tag}/scale"].copy()), se)
            parts.append(ql)
            Ws.append(W); scs.append(golden[f"{tag}/scale"]); biases.append(golden[f"{tag}/bias"])
            if gs != -1:
                ses.append(golden[f"{tag}/s_extra"])
        fused = fuse_quant_linears(parts)
        Wc = np.concatenate(Ws, 0)
        lin = torch.nn.Linear(Wc.shape[1], Wc.shape[0], bias=True).half()
        lin.weight.data = torch.from_numpy(Wc.copy())
        lin.bias.data = torch.from_numpy(np.concatenate(biases))
        ref = QuantLinear(4, gs, Wc.shape[1], Wc.shape[0], bias=True)
        ref.pack(lin, torch.from_numpy(np.concatenate(scs, 0)), torch.from_numpy(np.concatenate(ses, 0)) if gs != -1 else None)
        assert torch.equal(fused.B, ref.B) and torch.equal(fused.s_channel, ref.s_channel)
        assert torch.equal(fused.s_group, ref.s_group) and torch.equal(fused.bias, ref.bias)
