"""The oracle's closed-form layout against the way the REFERENCE KERNEL ITSELF reads the packed tensors (CPU only).

tests/marlin_model.py follows `Marlin<>` (/root/reference/csrc/qqq_gemm.cu:240-820) thread by thread -- its own index
expressions for the global -> shared copies, the ldmatrix / packed-word / scale reads, the mma.m16n8k16 fragment layouts, the
shared-memory reduce and the write-out.  Here that model runs on operands PACKED BY THE REFERENCE's pack() (the committed
goldens: `ref_B`, `ref_s_channel`, `ref_s_group`, activations from the reference's dynamic_quant) and must reproduce the
oracle's int32 accumulators and fp16 outputs bit for bit, for both tile configurations of the reference's large-m / small-m
tables (thread_k x thread_n = 128 x 128 and 64 x 256 with 256 threads, 128 x 64 and 64 x 128 with 128; csrc/qqq_gemm.cu:847-865,
:935-945) and both modes; since round 4 also for a grid of threadblocks (striped partition + global_reduce).
Until round 3 the oracle's layout was pinned through pack()'s OUTPUT only; this closes the other half -- how the kernel
consumes it -- without nvcc."""
import numpy as np
import pytest

from marlin_model import marlin_grid, marlin_one_block


@pytest.mark.parametrize("tag,tkb,tnb,threads", [
    ("g-1_n128_k256", 8, 8, 256), ("g128_n128_k256", 8, 8, 256),      # thread_k = 128, thread_n = 128
    ("g-1_n256_k256", 4, 16, 256), ("g128_n256_k256", 4, 16, 256),    # thread_k = 64,  thread_n = 256
    ("g-1_n256_k512", 8, 8, 256), ("g128_n256_k512", 8, 8, 256),      # two column slices x four k tiles
    ("g128_n128_k1024", 8, 8, 256),                                   # eight groups along k
    # the two 128-thread configurations of the reference's tables (csrc/qqq_gemm.cu:849-864, CALL_IF :937-945)
    ("g-1_n128_k256", 8, 4, 128), ("g128_n128_k256", 8, 4, 128),      # thread_k = 128, thread_n = 64
    ("g-1_n256_k512", 4, 8, 128), ("g128_n256_k512", 4, 8, 128),      # thread_k = 64,  thread_n = 128
    ("g128_n128_k1024", 8, 4, 128),
])
def test_reference_kernel_reads_reproduce_the_oracle(golden, tag, tkb, tnb, threads):
    B, s2, s3 = golden[f"{tag}/ref_B"], golden[f"{tag}/ref_s_channel"], golden[f"{tag}/ref_s_group"]
    done = 0
    for M in golden[f"{tag}/Ms"]:
        M = int(M)
        xq, s1 = golden[f"{tag}/m{M}/ref_xq"], golden[f"{tag}/m{M}/ref_s1"]
        eacc, eD = golden[f"{tag}/m{M}/oracle_acc"], golden[f"{tag}/m{M}/oracle_D"]
        # one threadblock handles up to 16 tokens (thread_m_blocks = 1); larger batches are `parallel` copies of the same
        # program on row blocks (csrc/qqq_gemm.cu:272-276, :296-303) -- run them block by block
        D = np.empty_like(eD)
        acc = np.empty_like(eacc)
        for r0 in range(0, M, 16):
            r1 = min(M, r0 + 16)
            D[r0:r1], acc[r0:r1] = marlin_one_block(xq[r0:r1], B, s1[r0:r1].reshape(-1), s2.reshape(-1),
                                                    s3 if s3.size else None, thread_k_blocks=tkb, thread_n_blocks=tnb, threads=threads)
        assert np.array_equal(acc, eacc), (tag, M)
        assert np.array_equal(D.view(np.uint16), eD.view(np.uint16)), (tag, M)
        done += 1
    assert done


@pytest.mark.parametrize("tag,tkb,tnb,threads", [
    ("g-1_n256_k512", 8, 8, 256), ("g128_n256_k512", 8, 8, 256), ("g128_n128_k1024", 8, 8, 256),
    ("g-1_n256_k512", 4, 16, 256), ("g128_n256_k512", 4, 8, 128), ("g-1_n256_k512", 8, 4, 128),
])
def test_reference_grid_stripes_and_global_reduce_reproduce_the_oracle(golden, tag, tkb, tnb, threads):
    """The rest of `Marlin<>`: SEVERAL threadblocks.  The striped partition (csrc/qqq_gemm.cu:261-338: `iters`, a stripe that starts in
    the middle of a column slice, `slice_count` / `slice_idx` numbered bottom to top) must cover every (column slice, k-tile) exactly
    once -- asserted inside the model -- and `global_reduce` (:606-676) adds the blocks of a column slice through the int32 buffer C in
    lock order (:213-237, :800-803): the first only writes, the last only reads and then runs the write-out.  C starts poisoned.  Grids
    of 2 ... 7 blocks (1 ... 4 blocks per column slice), both modes, all four tile configurations."""
    B, s2, s3 = golden[f"{tag}/ref_B"], golden[f"{tag}/ref_s_channel"], golden[f"{tag}/ref_s_group"]
    M = min(int(m) for m in golden[f"{tag}/Ms"] if int(m) >= 8)
    xq, s1 = golden[f"{tag}/m{M}/ref_xq"][:16], golden[f"{tag}/m{M}/ref_s1"][:16]
    eacc, eD = golden[f"{tag}/m{M}/oracle_acc"][:16], golden[f"{tag}/m{M}/oracle_D"][:16]
    for grid in (2, 3, 5, 7):
        D, acc = marlin_grid(xq, B, s1.reshape(-1), s2.reshape(-1), s3 if s3.size else None, grid=grid, thread_k_blocks=tkb,
                             thread_n_blocks=tnb, threads=threads)
        assert np.array_equal(acc, eacc), (tag, grid)
        assert np.array_equal(D.view(np.uint16), eD.view(np.uint16)), (tag, grid)


def test_model_detects_a_wrong_layout(golden):
    """the model is sensitive to what it is meant to pin: swapping two k-tiles of the packed tensor, or two columns of the
    stored group scales, changes its accumulators"""
    tag = "g128_n128_k256"
    B, s2, s3 = golden[f"{tag}/ref_B"].copy(), golden[f"{tag}/ref_s_channel"], golden[f"{tag}/ref_s_group"].copy()
    xq, s1 = golden[f"{tag}/m16/ref_xq"], golden[f"{tag}/m16/ref_s1"]
    eacc = golden[f"{tag}/m16/oracle_acc"]
    Bs = B.copy()
    Bs[[0, 1]] = Bs[[1, 0]]
    _, a1 = marlin_one_block(xq, Bs, s1.reshape(-1), s2.reshape(-1), s3)
    assert not np.array_equal(a1, eacc)
    s3s = s3.copy()
    s3s[:, [0, 1]] = s3s[:, [1, 0]]
    _, a2 = marlin_one_block(xq, B, s1.reshape(-1), s2.reshape(-1), s3s)
    assert not np.array_equal(a2, eacc)
