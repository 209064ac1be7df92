"""Parity of the HIP path (through the C-ABI) with the CPU oracle: int32 accumulators bit-exact,
fp16 outputs within 1 ulp (tolerance from BASELINE.json north_star; 0 ulp is what we expect and
report).  Small cases: committed golden fixtures, every kernel variant.  BASELINE sizes: C oracle on
full outputs (m <= 128) or a row subsample (rows are independent), plus size-independent properties."""
import numpy as np
import pytest
import torch

from conftest import gpu_dump
from gpu_util import GemmHarness, ulp_distance, variants
from qqq_amd import ops

pytestmark = pytest.mark.gpu
MAX_ULP = 1  # north_star: "fp16 outputs within 1 ulp of the reference dequant"


def _cases(golden):
    for tag in golden["cases"]:
        yield str(tag)


def test_golden_fixtures_all_variants(golden, dev):
    worst = 0
    failures = []
    for tag in _cases(golden):
        h = GemmHarness(golden[f"{tag}/ref_B"], golden[f"{tag}/ref_s_channel"], golden[f"{tag}/ref_s_group"], dev)
        for M in golden[f"{tag}/Ms"]:
            xq, s1 = golden[f"{tag}/m{M}/ref_xq"], golden[f"{tag}/m{M}/ref_s1"]
            eacc, eD = golden[f"{tag}/m{M}/oracle_acc"], golden[f"{tag}/m{M}/oracle_D"]
            for tune in variants(int(M), h.K, h.N):
                D, acc = h.run(xq, s1, tune)
                ok_acc = np.array_equal(acc, eacc)
                ulp = ulp_distance(D, eD) if not np.isnan(D.astype(np.float32)).any() else 10**6
                worst = max(worst, ulp if ok_acc else 0)
                if not ok_acc or ulp > MAX_ULP:
                    failures.append((tag, int(M), tune, bool(ok_acc), ulp))
                    if len(failures) <= 6:
                        gpu_dump(f"parity_{tag}_m{M}_{len(failures)}", acc=acc, eacc=eacc, D=D, eD=eD,
                                 tune=np.array(str(tune)))
    print("golden parity: worst fp16 ulp distance =", worst, "failures =", len(failures))
    assert not failures, failures[:12]
    assert worst == 0  # stronger than the stated tolerance: the epilogue is bit-identical


def test_scratch_reuse_and_repeated_calls(golden, dev):
    """Same layer buffers (C, workspace) reused across calls with different tokens, split-K in-launch
    reduction included: no call may see stale partial sums of the previous one."""
    tag = "g-1_n128_k256"
    h = GemmHarness(golden[f"{tag}/ref_B"], golden[f"{tag}/ref_s_channel"], golden[f"{tag}/ref_s_group"], dev)
    Ms = [int(m) for m in golden[f"{tag}/Ms"]]
    for rep in range(3):
        for M in Ms + Ms[::-1]:
            for tune in (dict(kernel=1, ksplit=2, waves=4, fused=1), dict(kernel=1, ksplit=2, waves=4, fused=2), dict(kernel=1, ksplit=2, waves=4, fused=3),
                         dict(kernel=2, bm=64, glds=1, ksplit=2)):
                D, acc = h.run(golden[f"{tag}/m{M}/ref_xq"], golden[f"{tag}/m{M}/ref_s1"], tune)
                assert np.array_equal(acc, golden[f"{tag}/m{M}/oracle_acc"]), (rep, M, tune)
                assert ulp_distance(D, golden[f"{tag}/m{M}/oracle_D"]) == 0


def test_fused_bias_every_path(golden, dev):
    """bias is added in fp16 AFTER the fp16 round (the reference's `D + self.bias`, qlinear_marlin.py:287), in
    the direct, LDS-transposed, in-launch-reduce and reduce-kernel epilogues alike."""
    for tag in ("g-1_n128_k256", "g128_n256_k512"):
        h = GemmHarness(golden[f"{tag}/ref_B"], golden[f"{tag}/ref_s_channel"], golden[f"{tag}/ref_s_group"], dev)
        bias = golden[f"{tag}/bias"]
        for M in golden[f"{tag}/Ms"]:
            exp = (torch.from_numpy(golden[f"{tag}/m{M}/oracle_D"].copy()) + torch.from_numpy(bias.copy())).numpy()
            for tune in variants(int(M), h.K, h.N):
                D, _ = h.run(golden[f"{tag}/m{M}/ref_xq"], golden[f"{tag}/m{M}/ref_s1"], tune, want_acc=False, bias=bias)
                assert ulp_distance(D, exp) == 0, (tag, int(M), tune)


def test_empty_and_error_behaviour(golden, dev):
    from qqq_amd import qqq_gemm

    tag = "g-1_n128_k256"
    h = GemmHarness(golden[f"{tag}/ref_B"], golden[f"{tag}/ref_s_channel"], golden[f"{tag}/ref_s_group"], dev)
    A = torch.zeros((0, h.K), dtype=torch.int8, device=dev)
    D = torch.zeros((0, h.N), dtype=torch.float16, device=dev)
    s1 = torch.zeros((0, 1), dtype=torch.float32, device=dev)
    qqq_gemm(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, -1, -1, -1, 16)  # m == 0: success, nothing launched
    A = torch.zeros((4, h.K), dtype=torch.int8, device=dev)
    D = torch.zeros((4, h.N), dtype=torch.float16, device=dev)
    s1 = torch.ones((4, 1), dtype=torch.float32, device=dev)
    with pytest.raises(RuntimeError, match="not compatible with thread_k=32"):
        qqq_gemm(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, 32, 128, -1, 16)
    with pytest.raises(RuntimeError, match="No kernel implementation for thread_k=64, thread_n=128"):
        qqq_gemm(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, 64, 128, -1, 16)
    qqq_gemm(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, 128, 128, -1, 16)  # a reference-valid user config
    torch.cuda.synchronize()


def test_random_shapes_against_oracle(dev):
    """Ragged shapes the reference admits (qlinear_marlin.py:65-77: K%64==0 & N%256==0, or (128,128),
    (128,64), (64,128) multiples): N%128==64 / N%256 in {64,128,192} column edges, K%128==64 (stream kernel
    only), m not a multiple of any tile, both modes, automatic dispatch and forced kernels."""
    from oracle import c_oracle as C
    from oracle import qqq_ref as R

    rng = np.random.default_rng(2024)
    shapes = [(1, 64, 128), (3, 192, 256), (17, 320, 192), (33, 448, 384), (65, 576, 640), (129, 704, 256),
              (130, 832, 1152), (257, 1088, 512), (300, 1344, 320), (513, 2112, 768), (40, 11008 // 43 * 2, 4096 // 8)]
    for (M, N, K) in shapes:
        for grouped in (False, True):
            if grouped and K % 128:
                continue
            if not any(K % tk == 0 and N % tn == 0 for tk, tn in [(64, 256), (128, 128), (128, 64), (64, 128)]):
                continue
            if grouped:
                codes = rng.integers(0, 16, size=(K, N), dtype=np.int8)
                s3 = (rng.random((K // 128, N), dtype=np.float32) * 15 + 0.5).astype(np.float16)
            else:
                codes = rng.integers(-8, 8, size=(K, N), dtype=np.int8)  # -8 included: the kernel path is generic
                s3 = np.zeros((0,), np.float16)
            B = R.pack_codes(codes, grouped)
            A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
            s1 = (rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001)
            s2 = (rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5)
            eD, eacc = C.qqq_gemm(A, B, s1, s2, s3 if grouped else None, return_acc=True)
            h = GemmHarness(B, s2, s3, dev)
            tunes = [None, dict(kernel=1), dict(kernel=1, ksplit=2, fused=1), dict(kernel=3), dict(kernel=3, mt=2, ksplit=2),
                     dict(kernel=4), dict(kernel=4, bm=256, ksplit=2), dict(kernel=4, waves=4, mt=2, pf=2), dict(kernel=4, mt=8, ksplit=3),
                     dict(kernel=4, bm=256, mt=8, pw=2), dict(kernel=4, bm=256, mt=8, pw=2, ksplit=2, pf=3),
                     dict(kernel=4, mt=8, ksplit=2, skew=2), dict(kernel=4, mt=4, ksplit=3, skew=1)]
            if K % 128 == 0:
                tunes += [dict(kernel=2), dict(kernel=2, bm=64, glds=1, stages=3), dict(kernel=2, bm=130, glds=1, stages=5),
                          dict(kernel=2, bm=258, glds=1, stages=3, ksplit=2)]
            for tune in tunes:
                D, acc = h.run(A, s1, tune)
                assert np.array_equal(acc, eacc), (M, N, K, grouped, tune)
                assert ulp_distance(D, eD) == 0, (M, N, K, grouped, tune)


def test_tiled_inlaunch_splitk_slot_chains(dev):
    """In-launch split-K of the tiled kernel: K slices of a tile meet in tile-sized int32 slots of C, in
    arrival order.  Shrinking max_par (rows of C = max_par*64, tickets = n/128*max_par) forces the
    single-slot case where deposits chain (read-add-write) and, below one slot, the fall-back to slabs;
    every combination must stay bit-exact, leave the workspace zero, and survive repeated calls."""
    from oracle import c_oracle as C
    from oracle import qqq_ref as R

    rng = np.random.default_rng(77)
    for (M, N, K, max_pars) in [(64, 256, 1024, (1, 2, 16)), (300, 768, 2048, (5, 8, 16)), (513, 320, 1536, (9, 16)),
                                (200, 1024, 4096, (4, 16))]:
        for grouped in (False, True):
            if grouped:
                codes = rng.integers(0, 16, size=(K, N), dtype=np.int8)
                s3 = (rng.random((K // 128, N), dtype=np.float32) * 15 + 0.5).astype(np.float16)
            else:
                codes = rng.integers(-8, 8, size=(K, N), dtype=np.int8)
                s3 = np.zeros((0,), np.float16)
            B = R.pack_codes(codes, grouped)
            A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
            s1 = (rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001)
            s2 = (rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5)
            eD, eacc = C.qqq_gemm(A, B, s1, s2, s3 if grouped else None, return_acc=True)
            for max_par in max_pars:
                h = GemmHarness(B, s2, s3, dev, max_par=max_par)
                for bm, extra in ((64, dict(glds=2)), (131, dict(glds=2)), (130, dict(glds=1, stages=4)),
                                  (256, dict(glds=1, stages=5)), (258, dict(glds=1, stages=3))):
                    for ks in (2, 3, 5, 8):
                        for fused in (1, 2):
                            tune = dict(kernel=2, bm=bm, ksplit=ks, fused=fused, **extra)
                            for rep in range(2):
                                D, acc = h.run(A, s1, tune)
                                assert np.array_equal(acc, eacc), (M, N, K, grouped, max_par, tune, rep)
                                assert ulp_distance(D, eD) == 0, (M, N, K, grouped, max_par, tune, rep)


# ------------------------------------------------------------------------------------------------
# BASELINE sizes: N=8192, K=21760
# ------------------------------------------------------------------------------------------------
N_FULL, K_FULL = 8192, 21760


def _full_layer(dev, grouped, seed=0):
    from qqq_amd import pack as P

    g = torch.Generator(device="cpu").manual_seed(seed)
    if grouped:
        codes = torch.randint(0, 16, (K_FULL, N_FULL), generator=g, dtype=torch.int8)
        s3 = (torch.rand((K_FULL // 128, N_FULL), generator=g) * 15.0 + 0.5).to(torch.float16)  # |(u-8)*s| <= 124
    else:
        codes = torch.randint(-7, 8, (K_FULL, N_FULL), generator=g, dtype=torch.int8)
        s3 = None
    s2 = (torch.rand((1, N_FULL), generator=g) * 2e-4 + 1e-5).to(torch.float32)
    B = P.pack_codes(codes.to(dev), grouped)
    return B, s2, s3


def _tokens(M, seed):
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    A = torch.randint(-128, 128, (M, K_FULL), generator=g, dtype=torch.int8)
    s1 = (torch.rand((M, 1), generator=g) * 0.05 + 0.001).to(torch.float32)
    return A, s1


@pytest.mark.parametrize("grouped", [False, True])
def test_baseline_sizes_against_oracle(grouped, dev):
    from oracle import c_oracle as C

    B, s2, s3 = _full_layer(dev, grouped)
    h = GemmHarness(B, s2, None if s3 is None else s3.to(dev), dev)
    Bn = B.cpu().numpy()
    s2n = s2.numpy()
    s3n = None if s3 is None else s3.numpy()
    ref_rows = {}
    for M in (1, 16, 128, 1024, 4096):
        A, s1 = _tokens(M, M)
        D, acc = h.run(A.numpy(), s1.numpy(), None)  # fully automatic dispatch, as a caller would
        rows = np.arange(M) if M <= 128 else np.unique(np.r_[0, M - 1, np.random.default_rng(M).integers(0, M, 46)])
        eD, eacc = C.qqq_gemm(A.numpy()[rows], Bn, s1.numpy()[rows], s2n, s3n, return_acc=True)
        assert np.array_equal(acc[rows], eacc), (grouped, M)
        assert ulp_distance(D[rows], eD) == 0, (grouped, M)
        assert not np.isnan(D.astype(np.float32)).any()
        ref_rows[M] = (A, s1, D, acc)
    # property: the two kernel families agree bit-for-bit on the same tokens at full size
    A, s1, D, acc = ref_rows[4096]
    for lo in (0, 2048, 4096 - 16):
        Ds, accs = h.run(A[lo : lo + 16].numpy(), s1[lo : lo + 16].numpy(), dict(kernel=1))
        assert np.array_equal(accs, acc[lo : lo + 16]) and np.array_equal(Ds.view(np.uint16), D[lo : lo + 16].view(np.uint16))
    # property: the decode (column) kernel and the stream kernel agree bit-for-bit at full size, any m <= 16
    A, s1, D, acc = ref_rows[16]
    for m in (1, 5, 16):
        for tune in (dict(kernel=3), dict(kernel=3, pf=8, ksplit=2), dict(kernel=1), dict(kernel=4), dict(kernel=4, bm=256, ksplit=8)):
            Dm, accm = h.run(A[:m].numpy(), s1[:m].numpy(), tune)
            assert np.array_equal(accm, acc[:m]) and np.array_equal(Dm.view(np.uint16), D[:m].view(np.uint16)), (m, tune)
    # property: the panel kernel (every shape) agrees with the automatic dispatch at full size for m = 16 .. 128
    for m_src, ms in ((128, (128, 100, 64, 33)), (16, (16, 9))):
        A, s1, D, acc = ref_rows[m_src]
        for m in ms:
            for tune in (dict(kernel=4), dict(kernel=4, bm=256), dict(kernel=4, waves=4), dict(kernel=4, ksplit=1, mt=2), dict(kernel=4, pf=2, ksplit=5)):
                Dm, accm = h.run(A[:m].numpy(), s1[:m].numpy(), tune)
                assert np.array_equal(accm, acc[:m]) and np.array_equal(Dm.view(np.uint16), D[:m].view(np.uint16)), (m, tune)
    # property: token order does not matter (rows independent) -- permute the M=128 batch
    A, s1, D, acc = ref_rows[128]
    perm = np.random.default_rng(3).permutation(128)
    Dp, accp = h.run(A.numpy()[perm], s1.numpy()[perm], None)
    assert np.array_equal(accp, acc[perm]) and np.array_equal(Dp.view(np.uint16), D[perm].view(np.uint16))
    # property: power-of-two rescaling of the token scales is exact in fp16 away from overflow/subnormals
    A, s1, D, acc = ref_rows[16]
    D2, _ = h.run(A.numpy(), (s1 * 0.5).numpy(), None)
    fin = np.abs(D.astype(np.float32)) > 2e-4
    assert np.array_equal((D2.astype(np.float32) * 2)[fin], D.astype(np.float32)[fin])


@pytest.mark.parametrize("gs,mode", [(-1, "per_channel"), (128, "g128")])
def test_baseline_sweep_on_reference_operands_pinned(gs, mode, dev):
    """BASELINE configs[1] / configs[2] on REFERENCE-MADE operands at full size (N=8192, K=21760): weights packed by the
    reference's pack(), activations quantised by the reference's dynamic_quant() (digests in tests/golden/fullsize_pins.json,
    made by tests/golden/gen_fullsize_pins.py in the build container); the HIP kernels -- automatic dispatch, plus the other
    large-m families where they apply -- must reproduce the committed SHA-256 digests of the oracle's int32 accumulators
    and fp16 outputs for every token count of the sweep."""
    import hashlib
    import json
    import os

    import fullsize_inputs as FI
    from oracle import c_oracle as C
    from qqq_amd import QuantLinear

    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    pins = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize_pins.json")))
    if "sweep_" + mode not in pins:
        pytest.skip("fullsize_pins.json carries no sweep pins")
    W_fq, scale, s_extra = FI.layer_inputs(gs)
    xs = FI.sweep_tokens()
    sw = pins["sweep_" + mode]
    if sha(W_fq) != pins[mode]["in_W_fq"] or sha(scale) != pins[mode]["in_scale"] or sha(xs) != sw["in_x"]:
        pytest.skip("numpy draws a different PCG64 normal stream than the pinned one: the pins do not apply")
    lin = torch.nn.Linear(FI.K_FULL, FI.N_FULL, bias=False).half()
    lin.weight.data = torch.from_numpy(W_fq)
    ql = QuantLinear(4, gs, FI.K_FULL, FI.N_FULL, bias=False).to(dev)
    ql.pack(lin.to(dev), torch.from_numpy(scale).to(dev), None if s_extra is None else torch.from_numpy(s_extra).to(dev))
    assert sha(ql.B.cpu().numpy()) == pins[mode]["ref_B"] and sha(ql.s_channel.cpu().numpy()) == pins[mode]["ref_s_channel"]
    assert sha(ql.s_group.cpu().numpy()) == pins[mode]["ref_s_group"]
    xq, s1 = C.dynamic_quant(xs, "div")  # the reference expression as the CPU evaluates it -- checked against the reference's run
    assert sha(xq) == sw["ref_xq"] and sha(s1) == sw["ref_s1"]
    h = GemmHarness(ql.B, ql.s_channel, ql.s_group if gs != -1 else None, dev)
    for M in sw["Ms"]:
        tunes = [None]
        if M >= 1024:
            tunes += [dict(kernel=2), dict(kernel=4, bm=256, mt=8, pw=2), dict(kernel=5)]
        elif M >= 128:
            tunes += [dict(kernel=1), dict(kernel=4, bm=256)]
        else:
            tunes += [dict(kernel=1), dict(kernel=3)]
        for tune in tunes:
            D, acc = h.run(xq[:M], s1[:M], tune)
            assert sha(acc) == sw[f"oracle_acc_m{M}"], (mode, M, tune)
            assert sha(D.view(np.uint16)) == sw[f"oracle_D_m{M}"], (mode, M, tune)


def test_llama7b_linear_shapes(dev):
    """BASELINE config 4 shapes (llama-2-7b linears), batch*seq rows subsampled for the CPU oracle."""
    from oracle import c_oracle as C
    from qqq_amd import pack as P

    for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008)):
        g = torch.Generator(device="cpu").manual_seed(N + K)
        codes = torch.randint(-7, 8, (K, N), generator=g, dtype=torch.int8)
        B = P.pack_codes(codes.to(dev), False)
        s2 = (torch.rand((1, N), generator=g) * 2e-4 + 1e-5).to(torch.float32)
        h = GemmHarness(B, s2, None, dev)
        for M in (8, 1024):
            A = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8)
            s1 = (torch.rand((M, 1), generator=g) * 0.05 + 0.001).to(torch.float32)
            D, acc = h.run(A.numpy(), s1.numpy(), None)
            rows = np.arange(M) if M <= 64 else np.random.default_rng(M).integers(0, M, 32)
            eD, eacc = C.qqq_gemm(A.numpy()[rows], B.cpu().numpy(), s1.numpy()[rows], s2.numpy(), None, return_acc=True)
            assert np.array_equal(acc[rows], eacc), (N, K, M)
            assert ulp_distance(D[rows], eD) == 0


def test_maximum_batch_and_grouped_llama_shape(dev):
    """BASELINE config 4 at its largest batch (32 x 1024 = 32768 tokens) and a per-group layer at 8192 tokens:
    many tile rounds, grid.y > 1 never needed; rows subsampled for the CPU oracle."""
    from oracle import c_oracle as C
    from qqq_amd import pack as P

    for (N, K, M, grouped) in ((4096, 4096, 32768, False), (11008, 4096, 8192, True)):
        g = torch.Generator(device="cpu").manual_seed(7 * N + K + M)
        if grouped:
            codes = torch.randint(0, 16, (K, N), generator=g, dtype=torch.int8)
            s3 = (torch.rand((K // 128, N), generator=g) * 15.0 + 0.5).to(torch.float16)
        else:
            codes = torch.randint(-7, 8, (K, N), generator=g, dtype=torch.int8)
            s3 = None
        B = P.pack_codes(codes.to(dev), grouped)
        s2 = (torch.rand((1, N), generator=g) * 2e-4 + 1e-5).to(torch.float32)
        h = GemmHarness(B, s2, None if s3 is None else s3.to(dev), dev)
        A = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8)
        s1 = (torch.rand((M, 1), generator=g) * 0.05 + 0.001).to(torch.float32)
        D, acc = h.run(A.numpy(), s1.numpy(), None)
        rows = np.unique(np.r_[0, M - 1, np.random.default_rng(M).integers(0, M, 30)])
        eD, eacc = C.qqq_gemm(A.numpy()[rows], B.cpu().numpy(), s1.numpy()[rows], s2.numpy(),
                              None if s3 is None else s3.numpy(), return_acc=True)
        assert np.array_equal(acc[rows], eacc), (N, K, M, grouped)
        assert ulp_distance(D[rows], eD) == 0
        assert not np.isnan(D.astype(np.float32)).any()


HANDOFFS = pytest.mark.parametrize("hand", [0, 4, 8, 12, 16], ids=["as-shipped", "acquire-fence", "release-publish", "both", "write-through-only"])


def _with_handoff(tune, hand):
    """tune.fused bits 2 / 3 / 4: the formal ends of the in-launch split-K hand-off (qqq_common.hip.h) -- an agent-scope acquire fence
    in front of the fold, an agent-scope release on the depositor's completion count -- and the wide kernel's XCD-local deposits
    switched off (every deposit written through, as in round 3).  The same library runs every variant."""
    t = dict(tune)
    t["fused"] = (t.get("fused", 1) & (3 | 64)) | hand  # (bit 6: the wide kernel's two-slice split WITHOUT the exchange hand-off)
    return t


@HANDOFFS
def test_inlaunch_splitk_stress_two_streams(dev, hand):
    """The slot / ticket hand-off of the tiled kernel's in-launch split-K under load: two layers (own scratch each, as
    two QuantLinear modules would have) hammered back to back from two streams, 300 calls with varying m and K
    splits; every result must equal the unsplit kernel's bit for bit and the workspaces must end all-zero."""
    from qqq_amd import pack as P

    g = torch.Generator(device="cpu").manual_seed(99)
    N, K = 2048, 4096
    layers = []
    for i in range(2):
        codes = torch.randint(-7, 8, (K, N), generator=g, dtype=torch.int8)
        B = P.pack_codes(codes.to(dev), False)
        s2 = (torch.rand((1, N), generator=g) * 2e-4 + 1e-5).to(torch.float32)
        layers.append(GemmHarness(B, s2, None, dev))
    Ms = [129, 200, 256, 300, 512, 777, 1024]
    toks, want = {}, {}
    for M in Ms:
        A = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
        s1 = (torch.rand((M, 1), generator=g) * 0.05 + 0.001).to(torch.float32).to(dev)
        toks[M] = (A, s1)
        for li, h in enumerate(layers):
            D = torch.empty((M, N), dtype=torch.float16, device=dev)
            ops.qqq_gemm_ex(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, -1, -1, -1, 16, tune=dict(kernel=2, bm=64, ksplit=1))
            want[(li, M)] = D
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for trial in range(3):  # (the first pass of a fresh process is the gentlest one: repeat)
        outs = []
        for it in range(150):
            for li, h in enumerate(layers):
                M = Ms[(it + 3 * li) % len(Ms)]
                A, s1 = toks[M]
                tune = [None, dict(kernel=2, bm=256, ksplit=2 + it % 3), dict(kernel=2, bm=131, ksplit=2 + it % 4),
                        dict(kernel=2, bm=64, ksplit=2 + it % 3)][it % 4]
                D = torch.empty((M, N), dtype=torch.float16, device=dev)
                with torch.cuda.stream(streams[li]):
                    ops.qqq_gemm_ex(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, -1, -1, -1, 16, tune=(_with_handoff(tune, hand) if tune else tune))
                outs.append((li, M, D, tune))
        torch.cuda.synchronize()
        for li, M, D, tune in outs:
            assert torch.equal(D.view(torch.int16), want[(li, M)].view(torch.int16)), (trial, li, M, tune)
        for h in layers:
            assert int(h.ws.abs().sum().item()) == 0


@HANDOFFS
def test_panel_inlaunch_splitk_stress_two_streams(dev, hand):
    """The panel kernel's ticket / slot hand-off under load: two layers (own scratch each) hammered from two streams, varying m
    (one to three m-blocks), K splits and shapes; every result must equal the unsplit stream kernel's bit for bit, the
    workspaces must end all-zero, and the poisoned reduce buffer must never leak into an output."""
    from qqq_amd import pack as P

    g = torch.Generator(device="cpu").manual_seed(123)
    N, K = 2048, 4096 + 64  # K % 128 == 64: the trailing half stage is exercised as well
    layers = []
    for i in range(2):
        codes = torch.randint(-7, 8, (K, N), generator=g, dtype=torch.int8)
        B = P.pack_codes(codes.to(dev), False)
        s2 = (torch.rand((1, N), generator=g) * 2e-4 + 1e-5).to(torch.float32)
        layers.append(GemmHarness(B, s2, None, dev))
    Ms = [33, 64, 100, 128, 200, 256, 300]
    toks, want = {}, {}
    for M in Ms:
        A = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
        s1 = (torch.rand((M, 1), generator=g) * 0.05 + 0.001).to(torch.float32).to(dev)
        toks[M] = (A, s1)
        for li, h in enumerate(layers):
            D = torch.empty((M, N), dtype=torch.float16, device=dev)
            ops.qqq_gemm_ex(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, -1, -1, -1, 16, tune=dict(kernel=1, ksplit=1))
            want[(li, M)] = D
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    # Repeated: light workgroups (4 waves, 64-token m-blocks) share CUs two and three at a time here, which is what
    # exposed a missing s_nop behind the inline-asm 16-byte deposit stores (data registers overwritten before the store
    # had read them: garbage in a slot) -- the first pass of a fresh process alone had always been green.
    for trial in range(4):
        outs = []
        for it in range(150):
            for li, h in enumerate(layers):
                M = Ms[(it + 2 * li) % len(Ms)]
                A, s1 = toks[M]
                tune = [dict(kernel=4), dict(kernel=4, ksplit=2 + it % 3), dict(kernel=4, bm=256, ksplit=2 + it % 2, pf=3),
                        dict(kernel=4, waves=4, ksplit=4, pf=2), dict(kernel=4, mt=4, ksplit=3), dict(kernel=4, bm=256, mt=8, pw=2, ksplit=2 + it % 2),
                        dict(kernel=4, waves=4, ksplit=3, pf=3, mt=2)][it % 7]
                D = torch.empty((M, N), dtype=torch.float16, device=dev)
                with torch.cuda.stream(streams[li]):
                    ops.qqq_gemm_ex(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, -1, -1, -1, 16, tune=(_with_handoff(tune, hand) if tune else tune))
                outs.append((li, M, D, tune))
        torch.cuda.synchronize()
        for li, M, D, tune in outs:
            assert torch.equal(D.view(torch.int16), want[(li, M)].view(torch.int16)), (trial, li, M, tune)
    for h in layers:
        assert int(h.ws.abs().sum().item()) == 0


@HANDOFFS
def test_wide_inlaunch_splitk_stress_two_streams(dev, hand):
    """The wide kernel's ticket / slot hand-off under load (row-major partial tiles written through to C, folded by the last
    arrival with agent-scope loads, no acquire fence): three layers with their own scratch hammered from two streams, both tile
    heights, 2-3 K slices, ragged m; every result equal to the unsplit stream kernel's bit for bit, workspaces all-zero after.
    Round 6: with fused bit 64 two slices of 256-column tiles EXCHANGE row halves (the first arrival decides from the partner's
    started bit; under this load some pairs fall back to the classic fold); the third layer is per-group with expanded int8
    weights (the loop that reads them has the same epilogue)."""
    from qqq_amd import pack as P

    g = torch.Generator(device="cpu").manual_seed(321)
    N, K = 2048, 4096
    layers = []
    for i in range(3):
        grouped = i >= 1
        codes = torch.randint(0 if grouped else -7, 16 if grouped else 8, (K, N), generator=g, dtype=torch.int8)
        B = P.pack_codes(codes.to(dev), grouped)
        s2 = (torch.rand((1, N), generator=g) * 2e-4 + 1e-5).to(torch.float32)
        s3 = (torch.rand((K // 128, N), generator=g) * 15.0 + 0.5).to(torch.float16) if grouped else None
        layers.append(GemmHarness(B, s2, s3, dev))
    layers[2].expand()
    Ms = [129, 256, 300, 512, 700, 1024]
    toks, want = {}, {}
    for M in Ms:
        A = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
        s1 = (torch.rand((M, 1), generator=g) * 0.05 + 0.001).to(torch.float32).to(dev)
        toks[M] = (A, s1)
        for li, h in enumerate(layers):
            D = torch.empty((M, N), dtype=torch.float16, device=dev)
            ops.qqq_gemm_ex(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, -1, -1, -1, 16, tune=dict(kernel=1, ksplit=1))
            want[(li, M)] = D
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for trial in range(3):
        outs = []
        for it in range(100):
            for li, h in enumerate(layers):
                M = Ms[(it + 2 * li) % len(Ms)]
                A, s1 = toks[M]
                tune = [dict(kernel=5, ksplit=2), dict(kernel=5, mt=8, ksplit=2), dict(kernel=5, ksplit=3), dict(kernel=5, mt=8, ksplit=3, pf=8),
                        dict(kernel=5, ksplit=2, pw=4), dict(kernel=5, bm=128, ksplit=2), dict(kernel=5, ksplit=2, fused=65), dict(kernel=5, mt=8, ksplit=2, fused=65, skew=2)][it % 8]
                D = torch.empty((M, N), dtype=torch.float16, device=dev)
                with torch.cuda.stream(streams[li % 2]):
                    ops.qqq_gemm_ex(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, -1, -1, -1, 16, tune=(_with_handoff(tune, hand) if tune else tune), W8=h.W8)
                outs.append((li, M, D, tune))
        torch.cuda.synchronize()
        for li, M, D, tune in outs:
            assert torch.equal(D.view(torch.int16), want[(li, M)].view(torch.int16)), (trial, li, M, tune)
    for h in layers:
        assert int(h.ws.abs().sum().item()) == 0


def test_panel_two_workgroups_per_cu(dev):
    """More panel workgroups than CUs, light enough (64-token m-blocks) for two to share a CU: the waves of a workgroup
    then drift apart behind a barrier.  Regression for a race of the two-buffer activation ring (the first step's fragment
    reads against stage 0's re-fill of the same buffer); per-group and per-channel, against the oracle."""
    from oracle import c_oracle as C
    from oracle import qqq_ref as R

    rng = np.random.default_rng(77)
    N, K = 16384, 2048
    for grouped in (True, False):
        codes = rng.integers(0 if grouped else -8, 16 if grouped else 8, size=(K, N)).astype(np.int8)
        B = R.pack_codes(codes, grouped)
        s2 = rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5
        s3 = (rng.random((K // 128, N), dtype=np.float32) * 15 + 0.5).astype(np.float16) if grouped else None
        h = GemmHarness(B, s2, s3, dev)
        for M in (64, 48, 128):
            A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
            s1 = rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001
            eD, eacc = C.qqq_gemm(A, B, s1, s2, s3, return_acc=True)
            for tune in (dict(kernel=4, ksplit=4), dict(kernel=4, ksplit=3, pf=2), dict(kernel=4, ksplit=2, pf=3), dict(kernel=4, mt=4, ksplit=4)):
                for rep in range(3):
                    D, acc = h.run(A, s1, tune)
                    assert np.array_equal(acc, eacc), (grouped, M, tune, rep)
                    assert ulp_distance(D, eD) == 0, (grouped, M, tune, rep)


def test_panel_slices_of_a_tile_on_one_xcd_any_strip_count(dev):
    """Round 5: for a split K the panel kernel walks its grid 8 strips at a time through all their K slices (every tile's slices on ONE XCD whatever the strip count;
    tune.fused bit 5 keeps the plain grid order).  The order must not matter: strip counts below 8 (the whole grid is the plain-order tail), exact multiples of 8, and
    ragged ones (9, 11, 18, 43 strips of 128 columns; 256-column strips halve them), 2 ... 4 slices, several m-blocks, both modes and both orders -- int32 accumulators and
    fp16 outputs against the CPU oracle, and the two orders bit-identical to each other."""
    from oracle import c_oracle as C
    from oracle import qqq_ref as R

    rng = np.random.default_rng(2025)
    K = 2048
    for N in (320, 1024, 1152, 1408, 2304, 5504):
        for grouped in (False, True):
            codes = rng.integers(0 if grouped else -8, 16 if grouped else 8, size=(K, N)).astype(np.int8)
            B = R.pack_codes(codes, grouped)
            s2 = rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5
            s3 = (rng.random((K // 128, N), dtype=np.float32) * 15 + 0.5).astype(np.float16) if grouped else None
            h = GemmHarness(B, s2, s3, dev)
            for M in (48, 200):
                A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
                s1 = rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001
                eD, eacc = C.qqq_gemm(A, B, s1, s2, s3, return_acc=True)
                for tune in (dict(kernel=4, ksplit=4), dict(kernel=4, ksplit=3), dict(kernel=4, bm=256, ksplit=2), dict(kernel=4, mt=4, ksplit=4, skew=2)):
                    got = []
                    for order in (0, 32):
                        D, acc = h.run(A, s1, dict(tune, fused=order))
                        assert np.array_equal(acc, eacc), (N, grouped, M, tune, order)
                        assert ulp_distance(D, eD) == 0, (N, grouped, M, tune, order)
                        got.append(D)
                    assert np.array_equal(got[0].view(np.uint16), got[1].view(np.uint16)), (N, grouped, M, tune)


def test_wide_every_instantiation_every_k_tail(dev):
    """The wide kernel stages by LDS-DMA and counts its own waits: every one of its twelve instantiations
    (per-channel / per-group x 256 x 256, 128 x 256, 256 x 128 tiles x ring of 4 / 8 steps), alone and in two K slices, on
    K = 1 ... 11 stages of 128 -- no loop trip at all, whole trips of four stages, and every length of ragged tail behind them
    (where the first LDS-DMA build let hipcc reuse the destination of a dead load) -- against the oracle, ragged m and n."""
    from oracle import c_oracle as C
    from oracle import qqq_ref as R

    rng = np.random.default_rng(2024)
    N, M = 448, 300  # 1.75 strips of 256 / 3.5 of 128; 1.2 / 2.3 m-tiles
    for grouped in (False, True):
        for st in (1, 2, 3, 4, 5, 6, 7, 8, 9, 11):
            K = 128 * st
            codes = rng.integers(0 if grouped else -8, 16 if grouped else 8, size=(K, N)).astype(np.int8)
            B = R.pack_codes(codes, grouped)
            s2 = rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5
            s3 = (rng.random((K // 128, N), dtype=np.float32) * 15 + 0.5).astype(np.float16) if grouped else None
            h = GemmHarness(B, s2, s3, dev)
            A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
            s1 = rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001
            eD, eacc = C.qqq_gemm(A, B, s1, s2, s3, return_acc=True)
            for shape in (dict(), dict(mt=8), dict(bm=128)):
                for pf in (4, 8):
                    for ks in (1, 2) if st >= 8 else (1,):
                        tune = dict(kernel=5, pf=pf, ksplit=ks, **shape)
                        D, acc = h.run(A, s1, tune)
                        assert np.array_equal(acc, eacc), (grouped, K, tune)
                        assert ulp_distance(D, eD) == 0, (grouped, K, tune)


def test_tile_walk_three_tiles_per_workgroup_every_stage_phase(dev):
    """The persistent tile walk of the wide kernel (tune glds=2): one workgroup per CU walks its run of tiles, the
    loads of a tile's last stages already fetch the next tile, the seam flushes the accumulators without touching LDS.  More than
    three tiles per workgroup, a ragged last m-tile, a last strip that overhangs n, and K = 8 ... 11 stages of 128: the flat stage
    sequence keeps rotating through the four LDS buffers / the register ring across seams, so a seam falls behind every stage
    position of the unrolled trip (K % 512 = 0, 128, 256, 384) -- all six instantiations against the oracle, int32 and fp16,
    with a bias; the plan must really be the tile walk."""
    from oracle import c_oracle as C
    from oracle import qqq_ref as R
    from qqq_amd import _lib

    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    rng = np.random.default_rng(4)
    N = 4160                                            # 16.25 strips of 256 / 32.5 of 128
    for grouped in (False, True):
        for st in (8, 9, 10, 11):
            K = 128 * st
            M = 256 * (-(-3 * cus // 17) + 1) + 37      # > 3 tiles of 256 x 256 per CU, ragged last m-tile
            codes = rng.integers(0 if grouped else -8, 16 if grouped else 8, size=(K, N)).astype(np.int8)
            B = R.pack_codes(codes, grouped)
            s2 = rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5
            s3 = (rng.random((K // 128, N), dtype=np.float32) * 15 + 0.5).astype(np.float16) if grouped else None
            bias = (rng.standard_normal(N) * 0.1).astype(np.float16)
            h = GemmHarness(B, s2, s3, dev)
            A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
            s1 = rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001
            eD, eacc = C.qqq_gemm(A, B, s1, s2, s3, return_acc=True)
            eDb = (torch.from_numpy(eD.copy()) + torch.from_numpy(bias)).numpy()
            for shape in (dict(), dict(mt=8), dict(bm=128)):
                tune = dict(kernel=5, glds=2, **shape)
                pl = _lib.plan(M, N, K, 128 if grouped else -1, 16, tune=tune)
                assert pl["kernel"] == 5 and pl["glds"] == 2 and pl["ksplit"] == 1, pl
                D, acc = h.run(A, s1, tune)
                assert np.array_equal(acc, eacc), (grouped, K, tune)
                assert ulp_distance(D, eD) == 0, (grouped, K, tune)
                Db, _ = h.run(A, s1, tune, want_acc=False, bias=bias)   # the production flush: no test hook in the way
                assert ulp_distance(Db, eDb) == 0, (grouped, K, tune)


def test_m_split_of_ragged_token_counts(dev):
    """A token count one past a whole number of tiles / rounds of the wide kernel runs as TWO launches (rows [0, split_m) by the wide kernel, the
    remainder as a call of its own -- column, stream or panel kernel --; include/qqq_amd.h `split_m`): int32 accumulators and fp16 outputs (with a bias)
    against the oracle over all rows, bit-identical to the unsplit launch of the same call, workspace zero afterwards; the plan must really be split."""
    from oracle import c_oracle as C
    from oracle import qqq_ref as R
    from qqq_amd import _lib

    rng = np.random.default_rng(11)
    for (M, N, K, grouped) in ((4099, 4096, 1024, False), (4200, 4096, 2048, True), (1030, 8192, 2048, False), (2051, 8192, 1024, True)):
        gs = 128 if grouped else -1
        pl = _lib.plan(M, N, K, gs, 16)
        assert pl["kernel"] == 5 and 0 < pl["split_m"] < M and pl["split_m"] % 256 == 0, pl
        assert _lib.plan(M, N, K, gs, 16, tune=dict(split_m=-1))["split_m"] == 0
        codes = rng.integers(0 if grouped else -8, 16 if grouped else 8, size=(K, N)).astype(np.int8)
        B = R.pack_codes(codes, grouped)
        s2 = rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5
        s3 = (rng.random((K // 128, N), dtype=np.float32) * 15 + 0.5).astype(np.float16) if grouped else None
        bias = (rng.standard_normal(N) * 0.1).astype(np.float16)
        h = GemmHarness(B, s2, s3, dev)
        A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
        s1 = rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001
        eD, eacc = C.qqq_gemm(A, B, s1, s2, s3, return_acc=True)
        eDb = (torch.from_numpy(eD.copy()) + torch.from_numpy(bias)).numpy()
        D, acc = h.run(A, s1, None)                        # automatic dispatch: the split
        assert np.array_equal(acc, eacc), (M, N, K, grouped)
        assert ulp_distance(D, eD) == 0, (M, N, K, grouped)
        Db, _ = h.run(A, s1, None, want_acc=False, bias=bias)
        assert ulp_distance(Db, eDb) == 0, (M, N, K, grouped)
        D1, acc1 = h.run(A, s1, dict(split_m=-1))          # the same call in one launch
        assert np.array_equal(acc1, eacc) and np.array_equal(D1.view(np.uint16), D.view(np.uint16)), (M, N, K, grouped)


def test_every_variant_under_load(dev):
    """Every tuning variant, repeatedly, while a second stream keeps the chip busy with other GEMMs of mixed weight (so
    that workgroups of different kernels share CUs and the waves of a workgroup drift apart): results must equal the
    quiet run's bit for bit.  Idle-chip parity alone had missed two races."""
    from oracle import qqq_ref as R

    rng = np.random.default_rng(4242)
    N, K = 2048, 2048
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for grouped in (False, True):
        codes = rng.integers(0 if grouped else -8, 16 if grouped else 8, size=(K, N)).astype(np.int8)
        B = torch.from_numpy(R.pack_codes(codes, grouped)).to(dev)
        s2 = torch.from_numpy(rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5).to(dev)
        s3 = torch.from_numpy((rng.random((K // 128, N), dtype=np.float32) * 15 + 0.5).astype(np.float16)).to(dev) if grouped else None
        h = GemmHarness(B, s2, s3, dev)       # the layer under test
        bg = GemmHarness(B, s2, s3, dev)      # the background layer: own scratch
        toks = {}
        for M in (40, 64, 200, 520):
            A = torch.from_numpy(rng.integers(-128, 128, size=(M, K), dtype=np.int8)).to(dev)
            s1 = torch.from_numpy(rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001).to(dev)
            toks[M] = (A, s1)
        bg_tunes = [dict(kernel=4, waves=4, ksplit=2, pf=2), dict(kernel=1), dict(kernel=2, bm=64, ksplit=2), dict(kernel=4, mt=4, ksplit=3), dict(kernel=3)]
        for M in (40, 200, 520):
            A, s1 = toks[M]
            want = torch.empty((M, N), dtype=torch.float16, device=dev)
            ops.qqq_gemm_ex(A, h.B, h.C, want, s1, h.s2, h.s3, h.ws, -1, -1, -1, 16, tune=dict(kernel=1, ksplit=1))
            torch.cuda.synchronize()
            for tune in variants(M, K, N):
                outs = []
                for it in range(6):
                    with torch.cuda.stream(streams[1]):
                        for j in range(3):
                            Mb = (64, 40, 200)[(it + j) % 3]
                            Ab, s1b = toks[Mb]
                            Db = torch.empty((Mb, N), dtype=torch.float16, device=dev)
                            ops.qqq_gemm_ex(Ab, bg.B, bg.C, Db, s1b, bg.s2, bg.s3, bg.ws, -1, -1, -1, 16, tune=bg_tunes[(it + j) % len(bg_tunes)])
                    D = torch.empty((M, N), dtype=torch.float16, device=dev)
                    with torch.cuda.stream(streams[0]):
                        ops.qqq_gemm_ex(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, -1, -1, -1, 16, tune=tune or None)
                    outs.append(D)
                torch.cuda.synchronize()
                for it, D in enumerate(outs):
                    assert torch.equal(D.view(torch.int16), want.view(torch.int16)), (grouped, M, tune, it)
            assert int(h.ws.abs().max()) == 0 and int(bg.ws.abs().max()) == 0


@HANDOFFS
def test_splitk_deposits_with_three_workgroups_per_cu(dev, hand):
    """The reproducer of the missing-s_nop bug (inline-asm 16-byte deposit stores whose data registers hipcc re-used before
    the store had read them): light 4-wave panel workgroups, three to a CU, the same split-K launch hammered from two
    streams.  Failed 25-40 % of the calls at m = 33 / 64 before the fix, on every box; idle-chip tests never did."""
    from qqq_amd import pack as P

    g = torch.Generator(device="cpu").manual_seed(123)
    N, K = 2048, 4096
    layers = []
    for i in range(2):
        codes = torch.randint(-7, 8, (K, N), generator=g, dtype=torch.int8)
        layers.append(GemmHarness(P.pack_codes(codes.to(dev), False), (torch.rand((1, N), generator=g) * 2e-4 + 1e-5).to(torch.float32), None, dev))
    Ms = [33, 64, 100, 128, 200, 256, 300]
    toks, want = {}, {}
    for M in Ms:
        A = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
        s1 = (torch.rand((M, 1), generator=g) * 0.05 + 0.001).to(torch.float32).to(dev)
        toks[M] = (A, s1)
        for li, h in enumerate(layers):
            D = torch.empty((M, N), dtype=torch.float16, device=dev)
            ops.qqq_gemm_ex(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, -1, -1, -1, 16, tune=dict(kernel=1, ksplit=1))
            want[(li, M)] = D
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for tune in (dict(kernel=4, waves=4, ksplit=4, pf=2), dict(kernel=4, waves=4, ksplit=4, pf=2, mt=2), dict(kernel=2, bm=64, ksplit=4), dict(kernel=4, mt=4, ksplit=4)):
        for trial in range(3):
            outs = []
            for it in range(100):
                for li, h in enumerate(layers):
                    M = Ms[(it + 2 * li) % len(Ms)]
                    A, s1 = toks[M]
                    D = torch.empty((M, N), dtype=torch.float16, device=dev)
                    with torch.cuda.stream(streams[li]):
                        ops.qqq_gemm_ex(A, h.B, h.C, D, s1, h.s2, h.s3, h.ws, -1, -1, -1, 16, tune=(_with_handoff(tune, hand) if tune else tune))
                    outs.append((li, M, D))
            torch.cuda.synchronize()
            bad = [(li, M) for li, M, D in outs if not torch.equal(D.view(torch.int16), want[(li, M)].view(torch.int16))]
            assert not bad, (tune, trial, len(bad), bad[:4])


def test_k_tail_large_m_through_auto_dispatch(dev):
    """K % 128 == 64 with many tokens (round 1 sent these to the stream kernel, one weight pass per 64 tokens): the
    automatic dispatch must take the panel kernel and match the oracle on sampled rows."""
    from oracle import c_oracle as C
    from oracle import qqq_ref as R
    from qqq_amd import _lib

    rng = np.random.default_rng(8)
    for (M, N, K, grouped) in ((1000, 1024, 1344, False), (3000, 768, 2112 + 64, False), (700, 512, 1280, True)):
        codes = rng.integers(0, 16, size=(K, N), dtype=np.int8) if grouped else rng.integers(-8, 8, size=(K, N), dtype=np.int8)
        s3 = (rng.random((K // 128, N), dtype=np.float32) * 15 + 0.5).astype(np.float16) if grouped else np.zeros((0,), np.float16)
        B = R.pack_codes(codes, grouped)
        A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
        s1 = (rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001)
        s2 = (rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5)
        if K % 128:
            assert _lib.plan(M, N, K, 128 if grouped else -1, 16)["kernel"] == 4
        h = GemmHarness(B, s2, s3, dev)
        D, acc = h.run(A, s1, None)
        rows = np.unique(np.r_[0, M - 1, rng.integers(0, M, 60)])
        eD, eacc = C.qqq_gemm(A[rows], B, s1[rows], s2, s3 if grouped else None, return_acc=True)
        assert np.array_equal(acc[rows], eacc), (M, N, K)
        assert ulp_distance(D[rows], eD) == 0


def test_split_k_scratch_stays_inside_the_reduce_buffer(dev):
    """Every split-K family keeps its partial sums inside the caller's `C` (max_par*64 x n int32) and its tickets inside
    `workspace` (n/128*max_par ints) -- also when the column strips overhang n (n % 256 != 0) and max_par is small.
    Both buffers are followed by sentinel words that must survive."""
    from oracle import c_oracle as C
    from oracle import qqq_ref as R

    rng = np.random.default_rng(31)
    for (M, N, K) in ((128, 320, 2048), (256, 192, 1536), (100, 1088, 1024)):
        codes = rng.integers(-8, 8, size=(K, N), dtype=np.int8)
        B = torch.from_numpy(R.pack_codes(codes, False)).to(dev)
        A = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
        s1 = (rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001)
        s2 = (rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5)
        eD = C.qqq_gemm(A, R.pack_codes(codes, False), s1, s2, None)
        tA, ts1, ts2 = torch.from_numpy(A).to(dev), torch.from_numpy(s1).to(dev), torch.from_numpy(s2).to(dev)
        s3 = torch.empty(0, dtype=torch.float16, device=dev)
        for max_par in (1, 2, 3, 16):
            nC, nW = max_par * 64 * N, max(N // 128, 1) * max_par
            Cbig = torch.full((nC + 4096,), 0x5A5A5A5A, dtype=torch.int32, device=dev)
            Wbig = torch.zeros(nW + 256, dtype=torch.int32, device=dev)
            Wbig[nW:] = 0x5A5A5A5A
            Cbuf, ws = Cbig[:nC].view(max_par * 64, N), Wbig[:nW]
            for tune in (None, dict(kernel=4, ksplit=2), dict(kernel=4, ksplit=4, bm=256), dict(kernel=4, ksplit=3, mt=4),
                         dict(kernel=2, bm=64, ksplit=3), dict(kernel=2, bm=131, ksplit=2), dict(kernel=1, ksplit=2),
                         dict(kernel=3, mt=2, ksplit=2)):
                D = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
                ops.qqq_gemm_ex(tA, B, Cbuf, D, ts1, ts2, s3, ws, -1, -1, -1, max_par, tune=tune)
                torch.cuda.synchronize()
                assert ulp_distance(D.cpu().numpy(), eD) == 0, (M, N, K, max_par, tune)
                assert bool((Cbig[nC:] == 0x5A5A5A5A).all()) and bool((Wbig[nW:] == 0x5A5A5A5A).all()), (M, N, K, max_par, tune)
                assert int(ws.abs().sum().item()) == 0


def test_sms_caps_the_compute_units_of_a_call(dev):
    """The reference's `sms` argument (csrc/qqq_gemm.cu:998: the number of persistent threadblocks) is a CU cap here: with 0 < sms < the
    device's CUs the call's kernels run on a CU-masked stream forked from / joined into the caller's stream.  Results are bit-identical
    whatever the cap, the caller's stream order holds (a dependent kernel on the caller's stream sees the output), and a quarter of
    the chip is measurably slower than all of it at a size that fills the chip."""
    import ctypes

    from qqq_amd import _lib, ops

    rng = np.random.default_rng(77)
    N, K = 4096, 4096
    codes = torch.from_numpy(rng.integers(-7, 8, size=(K, N), dtype=np.int8)).to(dev)
    from qqq_amd import pack as P

    B = P.pack_codes(codes, False)
    s2 = torch.from_numpy((rng.random((1, N), dtype=np.float32) * 2e-4 + 1e-5)).to(dev)
    s3 = torch.empty(0, dtype=torch.float16, device=dev)
    C = torch.zeros((16 * 64, N), dtype=torch.int32, device=dev)
    ws = torch.zeros(N // 128 * 16, dtype=torch.int32, device=dev)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    for M in (16, 128, 2048):
        A = torch.from_numpy(rng.integers(-128, 128, size=(M, K), dtype=np.int8)).to(dev)
        s1 = torch.from_numpy((rng.random((M, 1), dtype=np.float32) * 0.05 + 0.001)).to(dev)
        outs = {}
        for sms in (-1, cus, cus // 4, 8):
            D = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
            ops.qqq_gemm(A, B, C, D, s1, s2, s3, ws, -1, -1, sms, 16)
            chk = D.float().sum()  # a dependent kernel on the caller's stream: ordered behind the join
            torch.cuda.synchronize()
            assert bool(torch.isfinite(chk)), (M, sms)
            outs[sms] = D
            assert int(ws.abs().sum().item()) == 0
        for sms in (cus, cus // 4, 8):
            assert torch.equal(outs[sms], outs[-1]), (M, sms)
    # timing at 2048 tokens (128 tiles of the wide kernel or more): 8 CUs must be far slower than the whole chip
    def timed(sms, reps=5):
        ops.qqq_gemm(A, B, C, D, s1, s2, s3, ws, -1, -1, sms, 16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.qqq_gemm(A, B, C, D, s1, s2, s3, ws, -1, -1, sms, 16)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_all, t_8 = timed(-1), timed(8)
    assert t_8 > 3.0 * t_all, (t_all, t_8)
    # the capped call under stream capture: the fork / join are captured as cross-stream dependencies and the graph replays correctly
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.qqq_gemm(A, B, C, D, s1, s2, s3, ws, -1, -1, cus // 4, 16)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    D.fill_(float("nan"))
    with torch.cuda.graph(g):
        ops.qqq_gemm(A, B, C, D, s1, s2, s3, ws, -1, -1, cus // 4, 16)
    D.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(D, outs[-1])


def test_local_deposit_kill_switch_env(dev):
    """QQQ_AMD_NO_LOCAL_DEPOSITS=1 (read once per process) forces every split-K deposit of the wide kernel to be written through -- the escape hatch
    should a driver / partition-mode change ever break the XCD-local hand-off.  Same results, in a process of its own."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, QQQ_AMD_NO_LOCAL_DEPOSITS="1")
    cmd = [sys.executable, os.path.join(root, "tools", "check_variant.py"), "--nk", "4096,4096", "--ms", "700,1024",
           "--tunes", "[dict(kernel=5, ksplit=2), dict(kernel=5, bm=128, ksplit=2, skew=3), dict(kernel=5, mt=8, ksplit=3)]", "--ref", "dict(kernel=4, ksplit=1)"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0 and "MISMATCH" not in out and out.count("bit-exact") == 12, out[-2000:]
