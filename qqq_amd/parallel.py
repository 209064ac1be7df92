"""M-sharded W4A8 GEMM across GPUs: one process per GPU, RCCL (torch.distributed "nccl") over xGMI.

Rows of A / D are independent (the per-token scale s1[m] is row-local, the weights are shared), so every rank
computes a set of rows of D against the fully REPLICATED packed weights; the only communication is the all-gather of
the fp16 output rows (BASELINE configs[4]).  The reference has no distributed code at all -- this is new,
specified by north_star.

Row ownership is chunk-cyclic: the M rows are cut into `chunks` super-blocks of world*w rows (w = ceil(M / (world *
chunks))), and inside super-block c rank p owns the w rows starting at (c*world + p)*w.  Every super-block is then ONE
contiguous [world*w, N] slab of the output in which rank p's piece sits at offset p*w -- exactly the layout of an
in-place `all_gather_into_tensor` (no list of views, no staging copies), and super-block c can be gathered on a side
stream while super-block c+1 is still in the GEMM (xGMI is point-to-point: 7 links x ~153 GB/s per GPU, so a shard
takes about as long to move as to compute).  Only a ragged last super-block (M not a multiple of world*w)
goes through a padded scratch slab, copied out on the communication stream -- the host never synchronises.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


# xGMI / GEMM cost model of the chunk choice: a ring all-gather moves (world-1)/world of the
# output through every GPU's links; one link carries ~153 GB/s per direction.  The GEMM of r rows on the BASELINE layer takes
# ~ 25 us + r * 0.125 us from a few hundred rows up (profiles/r02_dispatch_check*.txt, r03_*: 60 us at 256 rows, 148 at 1024,
# 265 at 2048, 500 at 4096), scaled by the layer's N*K.
XGMI_LINK_GBS = 153.0


def gemm_us_model(rows: int, N: int, K: int) -> float:
    scale = (N * K) / (8192.0 * 21760.0)
    return (25.0 + 0.125 * max(rows, 0)) * scale if rows > 0 else 0.0


def allgather_us_model(rows_per_rank: int, N: int, world: int) -> float:
    """ring all-gather of `rows_per_rank` fp16 rows per rank: (world-1) steps, each moving one piece over one link"""
    return 20.0 + (world - 1) * rows_per_rank * N * 2 / (XGMI_LINK_GBS * 1e3)


def pick_chunks(M: int, N: int, world: int, K: int = 21760) -> int:
    """Chunks of the GEMM / all-gather pipeline, from the cost model above: with c chunks the step takes about
    gemm(first chunk) + (c - 1) * max(gemm(chunk), gather(chunk)) + gather(last chunk); every extra chunk costs the GEMM's
    fixed part once more.  The chunk count with the smallest estimate wins; a chunk never drops below 256 rows per rank
    (a full panel-kernel m-block pair -- below that the GEMM stops being a large-m GEMM)."""
    if world <= 1 or M <= 0:
        return 1
    rows = -(-M // world)
    best, best_us = 1, None
    for c in (1, 2, 3, 4):
        w = -(-rows // c)
        if c > 1 and w < 256:
            break
        g, a = gemm_us_model(w, N, K), allgather_us_model(w, N, world)
        us = g + (c - 1) * max(g, a) + a
        if best_us is None or us < best_us * 0.98:  # a further chunk must pay for itself
            best, best_us = c, us
    return best


def row_spans(M: int, world: int, rank: int, chunks: int) -> List[Tuple[int, int]]:
    """global row spans [start, end) owned by `rank`, one per super-block (possibly empty at the ragged end)"""
    chunks = max(1, chunks)
    w = -(-M // (world * chunks)) if M > 0 else 0
    out = []
    for c in range(chunks):
        s = min((c * world + rank) * w, M)
        e = min(s + w, M)
        out.append((s, e))
    return out


def take_rows(t: torch.Tensor, spans: List[Tuple[int, int]]) -> torch.Tensor:
    """the local rows of a replicated [M, ...] tensor, in span order, as one contiguous tensor"""
    parts = [t[s:e] for (s, e) in spans if e > s]
    if not parts:
        return t[:0].contiguous()
    return torch.cat(parts, dim=0) if len(parts) > 1 else parts[0].contiguous()


class ShardedGemm:
    """D_full[M,N] = all_gather_rows( gemm(A_local) ) with the chunk-cyclic row ownership of `row_spans`.

    gemm_fn(a_rows, s1_rows, d_rows_out) computes one contiguous block of local rows in place (the product passes a
    closure over qqq_amd.qqq_gemm; the CPU/gloo tests pass the oracle).  The inputs are either the REPLICATED [M_total, ..]
    tensors (sliced here -- the form to prefer: the ownership is not a contiguous shard, and a caller that cut its own
    contiguous shard would get rows of D_full in the wrong places) or this rank's rows in span order, made with
    `take_rows(t, sg.spans(M_total, N))` and passed with `local=True`.
    """

    def __init__(self, gemm_fn: Callable, group: Optional[dist.ProcessGroup] = None, chunks: Optional[int] = None,
                 comm_stream: Optional["torch.cuda.Stream"] = None, K: int = 21760):
        self.gemm_fn = gemm_fn
        self.group = group
        self.chunks = chunks  # None: pick_chunks(M, N, world, K) per call
        self.K = K            # the layer's reduction depth: the GEMM side of the chunk cost model scales with N * K (a 4096-deep
                              # layer's GEMM is 5x shorter than the BASELINE layer's, i.e. communication-bound sooner); spans() and
                              # __call__ must agree on the chunk count, hence stored here rather than passed per call
        self.comm_stream = comm_stream
        self._tail = None  # scratch slab of a ragged last super-block

    def spans(self, M_total: int, N: int) -> List[Tuple[int, int]]:
        world = dist.get_world_size(self.group)
        chunks = self.chunks if self.chunks else pick_chunks(M_total, N, world, self.K)
        return row_spans(M_total, world, dist.get_rank(self.group), chunks)

    def __call__(self, A: torch.Tensor, s1: torch.Tensor, M_total: int, N: int,
                 D_full: Optional[torch.Tensor] = None, local: bool = False, do_gemm: bool = True,
                 do_gather: bool = True) -> torch.Tensor:
        """do_gemm / do_gather = False leave out one half of the pipeline (bench.py times GEMM-only and all-gather-only next
        to the overlapped total); the result is only meaningful with both."""
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        chunks = self.chunks if self.chunks else pick_chunks(M_total, N, world, self.K)
        spans = row_spans(M_total, world, rank, chunks)
        if local:
            A_local, s1_local = A, s1
            if A_local.shape[0] != sum(e - s for s, e in spans):
                raise ValueError(f"ShardedGemm: {A_local.shape[0]} local rows, but this rank owns the spans {spans} (take_rows)")
        else:
            if A.shape[0] != M_total or s1.shape[0] != M_total:
                raise ValueError(f"ShardedGemm: expected the replicated [{M_total}, ...] inputs (got {A.shape[0]} rows); rows cut by "
                                 "the caller must follow `spans()` and be passed with local=True")
            A_local, s1_local = take_rows(A, spans), take_rows(s1, spans)
        dev = A_local.device
        w = -(-M_total // (world * chunks)) if M_total > 0 else 0
        if D_full is None:
            D_full = torch.empty((M_total, N), dtype=torch.float16, device=dev)
        assert D_full.is_contiguous() and D_full.shape == (M_total, N)
        if M_total == 0:
            return D_full
        use_streams = dev.type == "cuda"
        comm = None
        if use_streams:
            comm = self.comm_stream or torch.cuda.Stream(device=dev)
            self.comm_stream = comm
        off = 0
        for c, (s, e) in enumerate(spans):
            blk0 = c * world * w  # first row of super-block c
            if blk0 >= M_total:
                break  # nobody owns rows here
            full = blk0 + world * w <= M_total
            if full:
                slab = D_full[blk0 : blk0 + world * w]
            else:
                if self._tail is None or self._tail.shape != (world * w, N) or self._tail.device != dev:
                    self._tail = torch.empty((world * w, N), dtype=torch.float16, device=dev)
                slab = self._tail
            piece = slab[rank * w : (rank + 1) * w]  # in-place all-gather: this rank's piece inside the output slab
            if e > s:
                if do_gemm:
                    self.gemm_fn(A_local[off : off + (e - s)], s1_local[off : off + (e - s)], piece[: e - s])
                off += e - s
            if not do_gather:
                continue
            if use_streams:
                comm.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(comm):
                    dist.all_gather_into_tensor(slab, piece, group=self.group)
                    if not full:
                        D_full[blk0:M_total].copy_(slab[: M_total - blk0], non_blocking=True)
            else:
                dist.all_gather_into_tensor(slab, piece, group=self.group)
                if not full:
                    D_full[blk0:M_total].copy_(slab[: M_total - blk0])
        if use_streams:
            torch.cuda.current_stream(dev).wait_stream(comm)
        return D_full
