"""The oracle's closed-form layout against the way the REFERENCE KERNEL ITSELF reads the packed tensors (CPU only).

tests/marlin_model.py follows `Marlin<>` (/root/reference/csrc/qqq_gemm.cu:240-820) thread by thread -- its own index
expressions for the global -> shared copies, the ldmatrix / packed-word / scale reads, the mma.m16n8k16 fragment layouts, the
shared-memory reduce and the write-out.  Here that model runs on operands PACKED BY THE REFERENCE's pack() (the committed
goldens: `ref_B`, `ref_s_channel`, `ref_s_group`, activations from the reference's dynamic_quant) and must reproduce the
oracle's int32 accumulators and fp16 outputs bit for bit, for both tile configurations of the reference's large-m / small-m
tables that use 256 threads (thread_k x thread_n = 128 x 128 and 64 x 256; csrc/qqq_gemm.cu:847-865) and both modes.
Until round 3 the oracle's layout was pinned through pack()'s OUTPUT only; this closes the other half -- how the kernel
consumes it -- without nvcc."""
import numpy as np
import pytest

from marlin_model import marlin_one_block


@pytest.mark.parametrize("tag,tkb,tnb", [
    ("g-1_n128_k256", 8, 8), ("g128_n128_k256", 8, 8),      # thread_k = 128, thread_n = 128
    ("g-1_n256_k256", 4, 16), ("g128_n256_k256", 4, 16),    # thread_k = 64,  thread_n = 256
    ("g-1_n256_k512", 8, 8), ("g128_n256_k512", 8, 8),      # two column slices x four k tiles
    ("g128_n128_k1024", 8, 8),                              # eight groups along k
])
def test_reference_kernel_reads_reproduce_the_oracle(golden, tag, tkb, tnb):
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
                                                    s3 if s3.size else None, thread_k_blocks=tkb, thread_n_blocks=tnb)
        assert np.array_equal(acc, eacc), (tag, M)
        assert np.array_equal(D.view(np.uint16), eD.view(np.uint16)), (tag, M)
        done += 1
    assert done


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
