"""Round 6: SURVEY 8 f-3's opt-in load-time re-layout -- per-group weights expanded ONCE to int8 (qqq_expand_int8) and read by the wide
kernel instead of being re-quantised inside its loop.  Everything through the product's operator layer / C-ABI, against the CPU oracle:

  * the expander's bytes vs the oracle's restatement (values = dequant_per_group, csrc/qqq_gemm.cu:167-210; layout = include/qqq_amd.h),
    random layers with scales in the wrap region, and EXHAUSTIVELY: all 16 nibbles in every nibble position x every finite fp16 scale;
  * the GEMM through the expanded weights: every instantiation of the wide kernel that reads them (three tile shapes x one / two / three
    K slices, the persistent tile walk), every K tail, ragged m and n, int32 accumulators and fp16 outputs
    bit-exact, and bit-identical to the same call WITHOUT the expanded weights;
  * QuantLinear.expand_for_prefill(): forward() unchanged bit for bit, the buffer is non-persistent, follows .to(), drop_expanded().
"""
import numpy as np
import pytest
import torch

from gpu_util import GemmHarness, ulp_distance

pytestmark = pytest.mark.gpu


def _layer(rng, K, N, smax=15.0):
    from oracle import qqq_ref as R

    codes = rng.integers(0, 16, size=(K, N)).astype(np.int8)
    B = R.pack_codes(codes, True)
    s2 = rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5
    s3 = (rng.random((K // 128, N), dtype=np.float32) * smax + 0.5).astype(np.float16)
    return B, s2, s3


def test_expand_int8_matches_the_oracle(dev):
    from oracle import qqq_ref as R
    from qqq_amd import ops

    rng = np.random.default_rng(61)
    for (K, N, smax) in ((128, 64, 15.0), (512, 256, 15.0), (1024, 448, 40.0), (384, 8192, 15.0)):
        B, _, s3 = _layer(rng, K, N, smax)
        if smax > 20:
            s3[0, :8] = np.float16(6e-8)      # subnormal scales
            s3[1, :8] = np.float16(60000.0)   # products overflow fp16
            s3[2, 8:16] = np.float16(-3.5)    # negative scales
        got = ops.expand_int8(torch.from_numpy(B).to(dev), torch.from_numpy(s3).to(dev))
        torch.cuda.synchronize()
        assert got.dtype == torch.int8 and got.numel() == K * N
        exp = R.expand_int8(B, s3)
        assert np.array_equal(got.cpu().numpy(), exp), (K, N)
        # a per-channel layer of the same size: the operand is 16 * w4 (every signed nibble, -8 included)
        codes = rng.integers(-8, 8, size=(K, N)).astype(np.int8)
        Bc = R.pack_codes(codes, False)
        got = ops.expand_int8(torch.from_numpy(Bc).to(dev), torch.empty(0, dtype=torch.float16, device=dev)).cpu().numpy()
        assert np.array_equal(got, R.expand_int8(Bc)), (K, N)
        assert np.array_equal(got.astype(np.int16).reshape(K // 64, N // 64, 4, 64, 16)[0, 0, 0, 0], 16 * codes[:16, 0].astype(np.int16))


def test_expand_int8_exhaustive_every_nibble_every_finite_scale(dev):
    """One 128-k group, 63 488 columns: column n carries the n-th finite fp16 scale (either sign, subnormals, values whose products
    wrap or overflow), and down each column every nibble value appears in every nibble position of the packed words -- the expander
    must equal the bit-faithful restatement of dequant_per_group everywhere (as test_per_group_dequant_exhaustive_on_device does
    for the in-loop re-quantiser)."""
    from oracle import qqq_ref as R
    from qqq_amd import ops

    bits = np.arange(65536, dtype=np.uint32)
    finite = bits[(bits & 0x7C00) != 0x7C00].astype(np.uint16)
    N, K = finite.size, 128
    assert N % 64 == 0
    k = np.arange(K)[:, None]
    n = np.arange(N)[None, :]
    codes = ((k * 5 + (k // 16) * 3 + n // 64 + n) % 16).astype(np.int8)
    for u in range(16):  # every nibble value under every scale
        assert (codes == u).any(axis=0).all()
    B = R.pack_codes(codes, True)
    s_log = finite.view(np.float16).reshape(1, N)
    s3 = np.empty_like(s_log)
    s3[:, R.s_group_stored_index(np.arange(N))] = s_log  # stored order
    got = ops.expand_int8(torch.from_numpy(B).to(dev), torch.from_numpy(s3).to(dev)).cpu().numpy()
    exp = R.expand_int8(B, s3)
    assert np.array_equal(got, exp)
    prod = (np.arange(16)[:, None] - 8.0) * finite.view(np.float16).astype(np.float64)[None, :]
    assert (prod >= 127.5).any() and (prod < -128).any()


def test_gemm_through_expanded_weights_every_instantiation_every_k_tail(dev):
    """As test_wide_every_instantiation_every_k_tail, for the instantiations that read the expanded weights."""
    from oracle import c_oracle as C
    from qqq_amd import _lib

    rng = np.random.default_rng(2026)
    N, M = 448, 300
    for st in (1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 13):
        K = 128 * st
        B, s2, s3 = _layer(rng, K, N, 40.0 if st == 5 else 15.0)
        h = GemmHarness(B, s2, s3, dev)
        A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
        s1 = rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001
        eD, eacc = C.qqq_gemm(A, B, s1, s2, s3, return_acc=True)
        D0, acc0 = h.run(A, s1, dict(kernel=5))  # the in-loop re-quantiser
        assert np.array_equal(acc0, eacc) and ulp_distance(D0, eD) == 0
        h.expand()
        for shape in (dict(), dict(mt=8), dict(bm=128)):
            for ks in ((1, 2, 3) if st >= 12 else (1, 2) if st >= 8 else (1,)):
                tune = dict(kernel=5, ksplit=ks, **shape)
                pl = _lib.plan(M, N, K, 128, 16, tune=dict(w8=1, **tune))
                assert pl["w8"] == 1 and pl["pf"] == 4, pl  # (expanded weights: one ring depth)
                D, acc = h.run(A, s1, tune)
                assert np.array_equal(acc, eacc), (K, tune)
                assert np.array_equal(D.view(np.uint16), D0.view(np.uint16)), (K, tune)
        # tune.w8 = -1: the call has the expanded weights and ignores them
        assert _lib.plan(M, N, K, 128, 16, tune=dict(kernel=5, w8=-1))["w8"] == 0
        D, acc = h.run(A, s1, dict(kernel=5, w8=-1))
        assert np.array_equal(acc, eacc) and np.array_equal(D.view(np.uint16), D0.view(np.uint16))
        # every other family ignores them
        for tune in (dict(kernel=1), dict(kernel=4), dict(kernel=3, mt=2), dict(kernel=2, bm=128)):
            D, acc = h.run(A[:48], s1[:48], tune)
            assert np.array_equal(acc, eacc[:48]), (K, tune)
        # a per-channel layer through its expanded weights (16 * w4 as int8)
        from oracle import qqq_ref as R
        codes = rng.integers(-8, 8, size=(K, N)).astype(np.int8)
        Bc = R.pack_codes(codes, False)
        hc = GemmHarness(Bc, s2, None, dev)
        eD, eacc = C.qqq_gemm(A, Bc, s1, s2, None, return_acc=True)
        D0, acc0 = hc.run(A, s1, dict(kernel=5))
        assert np.array_equal(acc0, eacc) and ulp_distance(D0, eD) == 0
        hc.expand()
        for shape in (dict(), dict(mt=8), dict(bm=128)):
            for ks in ((1, 2) if st >= 8 else (1,)):
                tune = dict(kernel=5, ksplit=ks, **shape)
                assert _lib.plan(M, N, K, -1, 16, tune=dict(w8=1, **tune))["w8"] == 1
                D, acc = hc.run(A, s1, tune)
                assert np.array_equal(acc, eacc), (K, tune)
                assert np.array_equal(D.view(np.uint16), D0.view(np.uint16)), (K, tune)


def test_tile_walk_through_expanded_weights(dev):
    """The persistent tile walk on expanded weights: > 3 tiles per workgroup, ragged edges, a seam behind every stage position,
    with a bias (the production flush)."""
    from oracle import c_oracle as C
    from qqq_amd import _lib

    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    rng = np.random.default_rng(40)
    N = 4160
    for st in (8, 9, 10, 11):
        K = 128 * st
        M = 256 * (-(-3 * cus // 17) + 1) + 37
        B, s2, s3 = _layer(rng, K, N)
        bias = (rng.standard_normal(N) * 0.1).astype(np.float16)
        h = GemmHarness(B, s2, s3, dev).expand()
        A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
        s1 = rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001
        eD, eacc = C.qqq_gemm(A, B, s1, s2, s3, return_acc=True)
        eDb = (torch.from_numpy(eD.copy()) + torch.from_numpy(bias)).numpy()
        tune = dict(kernel=5, glds=2)
        pl = _lib.plan(M, N, K, 128, 16, tune=dict(w8=1, **tune))
        assert pl["kernel"] == 5 and pl["glds"] == 2 and pl["w8"] == 1 and pl["pf"] == 4, pl
        D, acc = h.run(A, s1, tune)
        assert np.array_equal(acc, eacc), (K, tune)
        assert ulp_distance(D, eD) == 0, (K, tune)
        Db, _ = h.run(A, s1, tune, want_acc=False, bias=bias)
        assert ulp_distance(Db, eDb) == 0, (K, tune)


def test_automatic_dispatch_uses_expanded_weights_only_where_the_wide_kernel_runs(dev):
    from oracle import c_oracle as C
    from qqq_amd import _lib

    rng = np.random.default_rng(8)
    N, K = 2048, 2048
    B, s2, s3 = _layer(rng, K, N)
    h = GemmHarness(B, s2, s3, dev).expand()
    for M in (1, 16, 100, 700, 1500, 4099):
        pl = _lib.plan(M, N, K, 128, 16, tune=dict(w8=1))
        assert (pl["w8"] == 1) == (pl["kernel"] == 5), (M, pl)
        A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
        s1 = rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001
        eD, eacc = C.qqq_gemm(A, B, s1, s2, s3, return_acc=True)
        D, acc = h.run(A, s1, None)
        assert np.array_equal(acc, eacc), M
        assert ulp_distance(D, eD) == 0, M
    assert _lib.plan(4099, N, K, 128, 16, tune=dict(w8=1))["w8"] == 1


def test_quantlinear_expand_for_prefill(golden, dev):
    from qqq_amd import QuantLinear

    torch.manual_seed(3)
    K, N = 1024, 768
    for gs in (128, -1):
        ql = QuantLinear(4, gs, K, N, bias=True)
        lin = torch.nn.Linear(K, N, bias=True).half()
        W = lin.weight.data.float()
        if gs == 128:
            Wg = W.reshape(N, K // 128, 128)
            sc = (2 * Wg.abs().amax(-1) / 15).clamp_min(1e-5)
            q = torch.clamp(torch.round(Wg / sc[..., None]) + 8, 0, 15)
            Wfq = ((q - 8) * sc[..., None]).reshape(N, K)
            s_extra = (Wfq.abs().amax(-1, keepdim=True) / 127).clamp_min(1e-8)
            lin.weight.data = Wfq.half()
            ql.pack(lin, sc, s_extra)
        else:
            sc = (W.abs().amax(-1, keepdim=True) / 7).clamp_min(1e-5)
            lin.weight.data = (torch.clamp(torch.round(W / sc), -7, 7) * sc).half()
            ql.pack(lin, sc)
        ql = ql.to(dev)
        xs = [torch.randn(m, K, device=dev).half() for m in (3, 400, 1300)]
        before = [ql(x).clone() for x in xs]
        keys = set(ql.state_dict().keys())
        assert ql.expand_for_prefill() is ql
        if gs == 128:
            assert ql.W8 is not None and ql.W8.dtype == torch.int8 and ql.W8.numel() == K * N and ql.W8.device.type == "cuda"
        else:
            assert ql.W8 is None  # per-channel: a no-op unless asked for
            ql.expand_for_prefill(per_channel=True)
            assert ql.W8 is not None and ql.W8.numel() == K * N
        assert set(ql.state_dict().keys()) == keys  # non-persistent: never in a checkpoint
        for x, b in zip(xs, before):
            assert torch.equal(ql(x), b)
        ql.expand_for_prefill()  # again (after new weights): replaces the buffer
        for x, b in zip(xs, before):
            assert torch.equal(ql(x), b)
        ql.drop_expanded()
        assert ql.W8 is None
        for x, b in zip(xs, before):
            assert torch.equal(ql(x), b)


def test_expanded_weights_trace_under_torch_compile(dev):
    from qqq_amd import ops

    rng = np.random.default_rng(12)
    K, N, M = 1024, 512, 600
    B, s2, s3 = _layer(rng, K, N)
    h = GemmHarness(B, s2, s3, dev).expand()
    x = torch.randn(M, K, device=dev).half()

    def f(x):
        return ops.quantlinear_forward(x, h.B, h.C, h.s2, h.s3, h.ws, None, 16, W8=h.W8)

    eager = f(x)
    comp = torch.compile(f, fullgraph=True)(x)
    assert torch.equal(eager, comp)
    assert torch.equal(eager, ops.quantlinear_forward(x, h.B, h.C, h.s2, h.s3, h.ws, None, 16))
