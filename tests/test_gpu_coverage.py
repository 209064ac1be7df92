"""Round-2 coverage on the GPU (all through the product's operator layer / C-ABI, checked against the CPU oracle):

  * f-3  on-device packer vs the oracle's pack, the reference's golden `ref_B`, and the full-size reference digests;
  * f-4  fused q/k/v and gate/up projections at the real Llama-2-7B shapes vs the oracle of the SEPARATE layers;
  * BASELINE configs[3]: the three distinct Llama-2-7B (N,K) x M in {1024, 8192, 32768} x {per-channel, g128}
    end to end through QuantLinear.forward (fused quant + GEMM + bias), oracle on a row subsample (rows are
    independent);
  * a-2  the kernels' per-group re-quantiser on the device over all 16 nibbles x every finite fp16 scale of either
    sign (wrap region, subnormals and overflow included) vs the bit-faithful restatement of dequant_per_group;
  * module hygiene: bias pinned to fp16 across .to(bf16/.float()), bias registered by pack(), any-rank dynamic_quant.
"""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import ulp_distance

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------------------------------------
# f-3: packer on the device
# ------------------------------------------------------------------------------------------------
def test_device_packer_matches_oracle_and_reference_goldens(golden, dev):
    from oracle import c_oracle as C
    from oracle import qqq_ref as R
    from qqq_amd import pack as P

    for tag in golden["cases"]:
        tag = str(tag)
        B = golden[f"{tag}/ref_B"]  # produced by the reference's own pack()
        grouped = golden[f"{tag}/ref_s_group"].size > 0
        codes = R.unpack_codes(B, grouped)
        got = P.pack_codes(torch.from_numpy(codes).to(dev), grouped)
        assert got.is_cuda and np.array_equal(got.cpu().numpy(), B), tag
        back = P.unpack_codes(torch.from_numpy(B).to(dev), grouped)
        assert back.is_cuda and np.array_equal(back.cpu().numpy(), codes), tag
    rng = np.random.default_rng(11)
    for (K, N) in ((64, 64), (1360, 320), (4096, 11008), (21760, 8192)):
        for grouped in (False, True):
            codes = rng.integers(0, 16, size=(K, N), dtype=np.int8) if grouped else rng.integers(-8, 8, size=(K, N), dtype=np.int8)
            want = C.pack(codes, grouped)
            got = P.pack_codes(torch.from_numpy(codes).to(dev), grouped)
            assert np.array_equal(got.cpu().numpy(), want), (K, N, grouped)
            assert np.array_equal(P.unpack_codes(got, grouped).cpu().numpy(), codes), (K, N, grouped)
            # int32 code tensors (what torch.round(...).int() yields in QuantLinear.pack) take the same path
            if K <= 4096:
                got32 = P.pack_codes(torch.from_numpy(codes.astype(np.int32)).to(dev), grouped)
                assert torch.equal(got32, got)


@pytest.mark.parametrize("gs,mode", [(-1, "per_channel"), (128, "g128")])
def test_device_quantlinear_pack_reproduces_fullsize_reference_digests(gs, mode, dev):
    """N=8192, K=21760: QuantLinear.pack with every tensor on the GPU must reproduce the SHA-256 digests of the
    reference's own pack() outputs (tests/golden/fullsize_pins.json, tests/golden/gen_fullsize_pins.py)."""
    import fullsize_inputs as FI
    from qqq_amd import QuantLinear

    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_pins.json")))
    W_fq, scale, s_extra = FI.layer_inputs(gs)
    if sha(W_fq) != pins[mode]["in_W_fq"] or sha(scale) != pins[mode]["in_scale"]:
        pytest.skip("numpy draws a different PCG64 normal stream than the pinned one: the pins do not apply")
    lin = torch.nn.Linear(FI.K_FULL, FI.N_FULL, bias=False).half()
    lin.weight.data = torch.from_numpy(W_fq)
    lin = lin.to(dev)
    ql = QuantLinear(4, gs, FI.K_FULL, FI.N_FULL, bias=False).to(dev)
    ql.pack(lin, torch.from_numpy(scale).to(dev), None if s_extra is None else torch.from_numpy(s_extra).to(dev))
    torch.cuda.synchronize()
    assert ql.B.is_cuda
    assert sha(ql.B.cpu().numpy()) == pins[mode]["ref_B"]
    assert sha(ql.s_channel.cpu().numpy()) == pins[mode]["ref_s_channel"]
    assert sha(ql.s_group.cpu().numpy()) == pins[mode]["ref_s_group"]


# ------------------------------------------------------------------------------------------------
# helpers: random layers with the stored tensors on the device
# ------------------------------------------------------------------------------------------------
def _random_layer(N, K, grouped, dev, seed, bias=True):
    from qqq_amd import QuantLinear, pack as P

    g = torch.Generator(device="cpu").manual_seed(seed)
    ql = QuantLinear(4, 128 if grouped else -1, K, N, bias=bias).to(dev)
    if grouped:
        codes = torch.randint(0, 16, (K, N), generator=g, dtype=torch.int8)
        ql.s_group.copy_((torch.rand((K // 128, N), generator=g) * 15.0 + 0.5).to(torch.float16))
    else:
        codes = torch.randint(-7, 8, (K, N), generator=g, dtype=torch.int8)
    ql.B.copy_(P.pack_codes(codes.to(dev), grouped))
    ql.s_channel.copy_((torch.rand((1, N), generator=g) * 2e-4 + 1e-5).to(torch.float32))
    if bias:
        ql.bias.copy_((torch.randn((N,), generator=g) * 0.1).to(torch.float16))
    return ql


def _oracle_forward(x_rows, ql):
    """reference chain on the CPU oracle: dynamic_quant (torch-GPU semantics) -> qqq_gemm -> fp16 bias add"""
    from oracle import c_oracle as C

    xq, s1 = C.dynamic_quant(x_rows, "recip")
    s3 = ql.s_group.cpu().numpy() if ql.s_group.numel() else None
    D = C.qqq_gemm(xq, ql.B.cpu().numpy(), s1, ql.s_channel.cpu().numpy(), s3)
    if ql.bias is not None:
        D = (torch.from_numpy(D.copy()) + ql.bias.cpu()).numpy()
    return D


def _tokens(M, K, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randn((M, K), generator=g, device=dev, dtype=torch.float32).to(torch.float16)


def _sample_rows(M, n, seed):
    if M <= n:
        return np.arange(M)
    return np.unique(np.r_[0, M - 1, np.random.default_rng(seed).integers(0, M, n - 2)])


# ------------------------------------------------------------------------------------------------
# f-4: fused projections at the real Llama-2-7B shapes vs the oracle of the separate layers
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("grouped", [False, True])
def test_fused_projections_llama_shapes_against_oracle(grouped, dev):
    """q/k/v (3 x 4096) and gate/up (2 x 11008) at K=4096 (gptq/models/llama.py:202-229, :275-283): ONE W4A8 GEMM with
    N = 12288 / 22016 must equal the reference chain evaluated layer by layer on the CPU oracle."""
    from qqq_amd import fuse_quant_linears

    K = 4096
    for name, widths in (("qkv", (4096, 4096, 4096)), ("gate_up", (11008, 11008))):
        parts = [_random_layer(n, K, grouped, dev, seed=100 * i + n + (7 if grouped else 0)) for i, n in enumerate(widths)]
        fused = fuse_quant_linears(parts)
        assert fused.outfeatures == sum(widths)
        for M in (1, 1024):
            x = _tokens(M, K, dev, seed=M)
            y = fused(x)
            torch.cuda.synchronize()
            assert y.shape == (M, sum(widths)) and y.dtype == torch.float16
            rows = _sample_rows(M, 24, M)
            xr = x[torch.from_numpy(rows).to(dev)].cpu().numpy()
            want = np.concatenate([_oracle_forward(xr, p) for p in parts], axis=1)
            got = y.cpu().numpy()[rows]
            assert ulp_distance(got, want) == 0, (name, grouped, M)
            assert int(fused.workspace.abs().sum().item()) == 0


# ------------------------------------------------------------------------------------------------
# BASELINE configs[3]: Llama-2-7B linear shapes x batch in {1, 8, 32} x seq 1024, end to end
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("grouped", [False, True])
@pytest.mark.parametrize("N,K", [(4096, 4096), (11008, 4096), (4096, 11008)])
def test_llama7b_quantlinear_forward_matrix(N, K, grouped, dev):
    """The 7 linears of a Llama-2-7B block are 3 distinct (N,K) (q/k/v/o, gate/up, down: gptq/models/llama.py:202-283);
    batch x seq = 1024 / 8192 / 32768 tokens; QuantLinear.forward = fused dynamic quant + W4A8 GEMM + fp16 bias
    (qlinear_marlin.py:270-288).  Oracle on <= 40 sampled rows per case."""
    ql = _random_layer(N, K, grouped, dev, seed=N + 3 * K + (1 if grouped else 0))
    for M in (1024, 8192, 32768):
        x = _tokens(M, K, dev, seed=M + N)
        y = ql(x.reshape(M // 1024, 1024, K))  # [batch, seq, hidden] as the model calls it
        torch.cuda.synchronize()
        assert y.shape == (M // 1024, 1024, N) and y.dtype == torch.float16
        rows = _sample_rows(M, 40, M + K)
        xr = x[torch.from_numpy(rows).to(dev)].cpu().numpy()
        want = _oracle_forward(xr, ql)
        got = y.reshape(M, N)[torch.from_numpy(rows).to(dev)].cpu().numpy()
        assert ulp_distance(got, want) == 0, (N, K, grouped, M)
        assert not torch.isnan(y).any()
        assert int(ql.workspace.abs().sum().item()) == 0
        del y, x


# ------------------------------------------------------------------------------------------------
# a-2: the per-group re-quantiser on the device, exhaustively
# ------------------------------------------------------------------------------------------------
def test_per_group_dequant_exhaustive_on_device(dev):
    """16 nibbles (in each of the 8 nibble positions of a packed word) x all 63 488 finite fp16 scales of either sign:
    the kernels' unpack_pair<GROUPED> must equal the bit-faithful restatement of dequant_per_group
    (csrc/qqq_gemm.cu:167-210) everywhere -- in range, in the wrap region (product >= 127.5 or < -128), for
    subnormal scales and for products that overflow fp16."""
    from oracle import qqq_ref as R
    from qqq_amd import _dev

    L = _dev.lib()
    bits = np.arange(65536, dtype=np.uint32)
    finite = bits[(bits & 0x7C00) != 0x7C00].astype(np.uint16)  # 63 488 patterns
    S = finite.size
    # word j (0..15): nibble position p holds u = (j + 3*p) % 16 -> every position sees every value
    words = np.zeros(16, np.uint32)
    for j in range(16):
        for p in range(8):
            words[j] |= np.uint32(((j + 3 * p) % 16) << (4 * p))
    q = np.repeat(words, S)
    s0 = np.tile(finite, 16)
    s1 = np.tile(finite[::-1].copy(), 16)  # a different scale for the b = 1 half
    n = q.size
    tq, t0, t1 = (torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a.view(np.int16)).to(dev) for a in (q, s0, s1))
    out = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    rc = L.qqq_dev_probe_dequant(tq.data_ptr(), t0.data_ptr(), t1.data_ptr(), out.data_ptr(), n, 0,
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, _dev.last_error()
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32).reshape(n, 2)
    # expected: b = 0 quadruple = nibbles (p0, p4, p1, p5) with s0, b = 1 quadruple = (p2, p6, p3, p7) with s1
    for half, (scales, pos) in enumerate(((s0, (0, 4, 1, 5)), (s1, (2, 6, 3, 7)))):
        exp = np.zeros(n, np.uint32)
        sc = scales.view(np.float16)
        for byte, p in enumerate(pos):
            u = ((q >> np.uint32(4 * p)) & np.uint32(0xF)).astype(np.int8)
            w8 = R.dequant_per_group_faithful(u, sc).view(np.uint8).astype(np.uint32)
            exp |= w8 << np.uint32(8 * byte)
        bad = np.nonzero(got[:, half] != exp)[0]
        assert bad.size == 0, (half, bad[:5], [hex(int(x)) for x in q[bad[:5]]], [hex(int(x)) for x in scales[bad[:5]]],
                               [hex(int(x)) for x in got[bad[:5], half]], [hex(int(x)) for x in exp[bad[:5]]])
    # and the wrap region is really inside what was covered: some products are far outside [-128, 127.5)
    prod = (np.arange(16)[:, None] - 8.0) * finite.view(np.float16).astype(np.float64)[None, :]
    assert (prod >= 127.5).any() and (prod < -128).any()


def test_per_group_gemm_with_scales_in_the_wrap_region(dev):
    """End to end through the GEMM kernels with per-group scales up to 40 (|(u-8)*s| up to 320: the low byte wraps
    exactly like the reference kernel's does) -- every kernel family, int32 accumulators bit-exact vs the oracle."""
    from gpu_util import GemmHarness
    from oracle import c_oracle as C
    from oracle import qqq_ref as R

    rng = np.random.default_rng(5)
    M, N, K = 48, 256, 512
    codes = rng.integers(0, 16, size=(K, N), dtype=np.int8)
    s3 = (rng.random((K // 128, N), dtype=np.float32) * 40.0).astype(np.float16)
    s3[0, :8] = np.float16(6e-8)  # subnormal scales
    s3[1, :8] = np.float16(60000.0)  # products overflow fp16
    B = R.pack_codes(codes, True)
    A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    s1 = (rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001)
    s2 = (rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5)
    eD, eacc = C.qqq_gemm(A, B, s1, s2, s3, return_acc=True)
    assert np.array_equal(eacc, R.gemm_int32(A, R.weight_operand(B, s3, True)))
    h = GemmHarness(B, s2, s3, dev)
    for tune in (None, dict(kernel=1), dict(kernel=3, mt=2), dict(kernel=2, bm=64), dict(kernel=2, bm=130, glds=1, stages=3),
                 dict(kernel=2, bm=258, glds=1, stages=3), dict(kernel=2, bm=256, glds=1, stages=6), dict(kernel=4), dict(kernel=4, bm=256, ksplit=2)):
        D, acc = h.run(A, s1, tune)
        assert np.array_equal(acc, eacc), tune
        assert ulp_distance(D, eD) == 0, tune


# ------------------------------------------------------------------------------------------------
# module hygiene
# ------------------------------------------------------------------------------------------------
def test_bias_survives_model_wide_dtype_casts(dev):
    """The reference's prepare_for_inference does model.to(bfloat16/float32); _apply pins the scale dtypes
    (qlinear_marlin.py:141-145).  The fused epilogue reads the bias as fp16 bits, so the bias is pinned too."""
    ql = _random_layer(256, 256, False, dev, seed=3)
    x = _tokens(9, 256, dev, seed=1)
    want = _oracle_forward(x.cpu().numpy(), ql)
    for cast in (lambda m: m.to(torch.bfloat16), lambda m: m.float(), lambda m: m.half()):
        ql = cast(ql)
        assert ql.bias.dtype == torch.float16 and ql.s_channel.dtype == torch.float32 and ql.B.dtype == torch.int32
        y = ql(x)
        torch.cuda.synchronize()
        assert ulp_distance(y.cpu().numpy(), want) == 0
    # a wrong-dtype bias handed straight to the operator is rejected, not reinterpreted
    from qqq_amd import ops

    with pytest.raises(RuntimeError, match="bias must be a contiguous fp16"):
        ops.quantlinear_forward(x, ql.B, ql.reduce_buffer, ql.s_channel, ql.s_group, ql.workspace, ql.bias.float())
    with pytest.raises(RuntimeError, match="s2"):  # eager check or, on the dispatcher path, the custom op's own
        ops.quantlinear_forward(x, ql.B, ql.reduce_buffer, ql.s_channel[:, :128].contiguous(), ql.s_group, ql.workspace, None)


def test_pack_registers_bias_of_a_layer_built_without_one(golden, dev):
    from qqq_amd import QuantLinear

    tag = "g-1_n128_k256"
    W = golden[f"{tag}/W_fq"]
    N, K = W.shape
    lin = torch.nn.Linear(K, N, bias=True).half()
    lin.weight.data = torch.from_numpy(W.copy())
    lin.bias.data = torch.from_numpy(golden[f"{tag}/bias"].copy())
    ql = QuantLinear(4, -1, K, N, bias=False)
    ql.pack(lin, torch.from_numpy(golden[f"{tag}/scale"].copy()))
    ql = ql.to(dev)  # the bias must move with the module (the reference leaves a plain attribute behind)
    assert ql.bias.is_cuda and "bias" in ql.state_dict()
    x = torch.from_numpy(golden[f"{tag}/m16/x"].copy()).to(dev)
    y = ql(x)
    torch.cuda.synchronize()
    assert ulp_distance(y.cpu().numpy(), _oracle_forward(golden[f"{tag}/m16/x"], ql)) == 0


def test_dynamic_quant_any_rank(dev):
    from oracle import c_oracle as C
    from qqq_amd import dynamic_quant

    g = torch.Generator(device="cpu").manual_seed(4)
    x = (torch.randn((2, 5, 512), generator=g) * 2.0).to(torch.float16)
    xq, s1 = dynamic_quant(x.to(dev))
    assert xq.shape == (2, 5, 512) and s1.shape == (2, 5, 1) and s1.dtype == torch.float32
    oq, os1 = C.dynamic_quant(x.reshape(10, 512).numpy(), "recip")
    assert np.array_equal(xq.cpu().numpy().reshape(10, 512), oq) and np.array_equal(s1.cpu().numpy().reshape(10, 1), os1)
    x1 = x[0, 0]
    xq1, s11 = dynamic_quant(x1.to(dev))
    assert xq1.shape == (512,) and s11.shape == (1,) and np.array_equal(xq1.cpu().numpy(), oq[0])


def test_dispatcher_path_is_taken_under_dispatch_modes(golden, dev):
    """Eager calls bypass the torch dispatcher only for plain tensors with no dispatch mode active; under a
    TorchDispatchMode the registered custom ops must be what runs (visible to dispatch-based tooling)."""
    from torch.utils._python_dispatch import TorchDispatchMode

    from qqq_amd import QuantLinear

    seen = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            seen.append(str(func))
            return func(*args, **(kwargs or {}))

    tag = "g-1_n128_k256"
    ql = QuantLinear(4, -1, 256, 128, bias=True)
    ql.load_state_dict({"B": torch.from_numpy(golden[f"{tag}/ref_B"].copy()),
                        "s_channel": torch.from_numpy(golden[f"{tag}/ref_s_channel"].copy()),
                        "s_group": torch.empty(0, dtype=torch.float16),
                        "bias": torch.from_numpy(golden[f"{tag}/bias"].copy())})
    ql = ql.to(dev)
    x = torch.from_numpy(golden[f"{tag}/m16/x"].copy()).to(dev)
    eager = ql(x)
    with Spy():
        out = ql(x)
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    assert any("qqq_amd.qqq_gemm" in s for s in seen) and any("qqq_amd.dynamic_quant" in s for s in seen), seen
