"""Pins the CPU oracle (oracle/qqq_ref.py numpy + oracle/qqq_oracle.c) against vectors produced by the
REFERENCE's own python (QuantLinear.pack / dynamic_quant, imported by path in the build container by
tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import qqq_ref as R


def _cases(golden):
    """yields (tag, group_size) with group_size = -1 whenever the reference treats the layer as
    per-channel (group_size == -1 or group_size == infeatures, qlinear_marlin.py:87,92)."""
    for tag in golden["cases"]:
        tag = str(tag)
        gs = int(tag.split("_")[0][1:])
        if golden[f"{tag}/ref_s_group"].size == 0:
            gs = -1
        yield tag, gs


def test_pack_matches_reference_pack(golden):
    for tag, gs in _cases(golden):
        se = golden[f"{tag}/s_extra"] if gs != -1 else None
        # hand pack_from_fakequant the nominal group size: it must detect the group == K edge itself
        nominal = int(tag.split("_")[0][1:])
        B, s2, s3 = R.pack_from_fakequant(golden[f"{tag}/W_fq"], golden[f"{tag}/scale"], se, nominal)
        assert np.array_equal(B, golden[f"{tag}/ref_B"]), tag
        assert np.array_equal(s2.view(np.uint32), golden[f"{tag}/ref_s_channel"].view(np.uint32)), tag
        if gs != -1:
            assert np.array_equal(s3.view(np.uint16), golden[f"{tag}/ref_s_group"].view(np.uint16)), tag
        else:
            assert golden[f"{tag}/ref_s_group"].size == 0


def test_unpack_is_inverse_of_reference_layout(golden):
    for tag, gs in _cases(golden):
        B = golden[f"{tag}/ref_B"]
        grouped = gs != -1
        codes = R.unpack_codes(B, grouped)
        assert np.array_equal(codes, C.unpack(B, grouped))
        assert np.array_equal(R.pack_codes(codes, grouped), B)
        assert np.array_equal(C.pack(codes, grouped), B)
        # the integers the fake-quant weights encode are recovered exactly (qlinear_marlin.py:202)
        W = golden[f"{tag}/W_fq"].astype(np.float32)
        sc = golden[f"{tag}/scale"]
        if grouped:
            sc = np.repeat(sc, gs, axis=1)
            assert np.array_equal(codes.T, np.clip(np.rint(W / sc) + 8, 0, 15))
        else:
            assert np.array_equal(codes.T, np.clip(np.rint(W / sc), -7, 7))


def test_dynamic_quant_matches_reference(golden):
    for tag, gs in _cases(golden):
        for M in golden[f"{tag}/Ms"]:
            x = golden[f"{tag}/m{M}/x"]
            for impl in (R, C):
                xq, s1 = impl.dynamic_quant(x, "div")
                assert np.array_equal(xq, golden[f"{tag}/m{M}/ref_xq"])
                assert np.array_equal(s1.view(np.uint32), golden[f"{tag}/m{M}/ref_s1"].view(np.uint32))
            a = R.dynamic_quant(x, "recip")
            b = C.dynamic_quant(x, "recip")
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_gemm_c_equals_numpy_equals_pins(golden):
    for tag, gs in _cases(golden):
        B, s2, s3 = golden[f"{tag}/ref_B"], golden[f"{tag}/ref_s_channel"], golden[f"{tag}/ref_s_group"]
        for M in golden[f"{tag}/Ms"]:
            xq, s1 = golden[f"{tag}/m{M}/ref_xq"], golden[f"{tag}/m{M}/ref_s1"]
            Dn, accn = R.qqq_gemm(xq, B, s1, s2, s3, return_acc=True)
            Dc, accc = C.qqq_gemm(xq, B, s1, s2, s3, return_acc=True)
            assert np.array_equal(accn, accc)
            assert np.array_equal(Dn.view(np.uint16), Dc.view(np.uint16))
            assert np.array_equal(accn, golden[f"{tag}/m{M}/oracle_acc"])
            assert np.array_equal(Dn.view(np.uint16), golden[f"{tag}/m{M}/oracle_D"].view(np.uint16))


def test_gemm_agrees_with_reference_fake_quant_path(golden):
    """The reference defines the fake-quant float path D ~= (xq*s1) @ W_fq.T.  It can only be
    an ABSOLUTE-tolerance check (near-zero outputs differ by many ulps); what it pins bit-exactly are
    the integer operands, checked in test_unpack_is_inverse_of_reference_layout."""
    for tag, gs in _cases(golden):
        W = golden[f"{tag}/W_fq"].astype(np.float32)
        K = W.shape[1]
        for M in golden[f"{tag}/Ms"]:
            xq, s1 = golden[f"{tag}/m{M}/ref_xq"], golden[f"{tag}/m{M}/ref_s1"]
            D = golden[f"{tag}/m{M}/oracle_D"].astype(np.float32)
            Dfq = (xq.astype(np.float32) * s1) @ W.T
            # per-channel: fp16 half-ulp of D + 2^-11 relative rounding of W_fq;
            # per-group adds the second-level int8 rounding (<= 0.5*s_extra per weight)
            tol = 8e-3 if gs == -1 else 0.15
            assert np.abs(D - Dfq).max() <= tol * max(1.0, np.sqrt(K / 256.0)), (tag, M)


def test_per_group_dequant_exhaustive():
    """dequant_per_group restatement == rint((u-8)*s) wherever the product is in [-128, 127.5);
    all 16 nibbles x all finite non-negative fp16 scales."""
    u = np.repeat(np.arange(16, dtype=np.int8), 31744).reshape(16, 31744)
    s = np.tile(np.arange(31744, dtype=np.uint16), (16, 1)).view(np.float16)
    w8 = R.dequant_per_group_faithful(u, s)
    prod = (u.astype(np.float64) - 8) * s.astype(np.float64)
    inr = (prod >= -128.0) & (prod < 127.5)
    assert np.array_equal(w8[inr], np.rint(prod[inr]).astype(np.int8))
    # the C oracle takes the same path: run it through a synthetic packed tensor
    K, N = 128, 64
    rng = np.random.default_rng(3)
    codes = rng.integers(0, 16, size=(K, N), dtype=np.int8)
    s3 = rng.integers(0, 31744, size=(1, N)).astype(np.uint16).view(np.float16)
    B = R.pack_codes(codes, True)
    assert np.array_equal(C.weight_operand(B, s3, 128), R.weight_operand(B, s3, True))
