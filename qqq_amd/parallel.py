"""M-sharded W4A8 GEMM across GPUs: one process per GPU, RCCL (torch.distributed "nccl") over xGMI.

Rows of A / D are independent (the per-token scale s1[m] is row-local, the weights are shared), so
rank r owns rows [r*M/P, (r+1)*M/P) and computes its shard of D against the fully REPLICATED packed
weights; the only communication is one all-gather of the fp16 output shards (BASELINE config 5).
The reference has no distributed code at all -- this is new, specified by north_star.

xGMI is point-to-point, so the gather is chunk-pipelined: the local rows are cut into `chunks`
pieces, chunk i is gathered on a side stream while chunk i+1 is still in the GEMM.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_rows(M: int, world: int, rank: int) -> Tuple[int, int]:
    """Balanced contiguous row partition: the first M % world ranks get one extra row."""
    base, extra = divmod(M, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def chunk_bounds(rows: int, chunks: int) -> List[Tuple[int, int]]:
    chunks = max(1, min(chunks, rows)) if rows > 0 else 1
    out = []
    for i in range(chunks):
        a = (rows * i) // chunks
        b = (rows * (i + 1)) // chunks
        out.append((a, b))
    return out


class ShardedGemm:
    """D_full[M,N] = all_gather_rows( gemm(A_local) ).

    gemm_fn(a_rows, s1_rows, d_rows_out) computes one contiguous block of local rows in place
    (the product passes a closure over qqq_amd.qqq_gemm; the CPU/gloo tests pass the oracle).
    """

    def __init__(self, gemm_fn: Callable, group: Optional[dist.ProcessGroup] = None, chunks: int = 2,
                 comm_stream: Optional["torch.cuda.Stream"] = None):
        self.gemm_fn = gemm_fn
        self.group = group
        self.chunks = chunks
        self.comm_stream = comm_stream

    def __call__(self, A_local: torch.Tensor, s1_local: torch.Tensor, M_total: int, N: int,
                 D_full: Optional[torch.Tensor] = None) -> torch.Tensor:
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        r0, r1 = shard_rows(M_total, world, rank)
        rows = r1 - r0
        assert A_local.shape[0] == rows, (A_local.shape, rows)
        dev = A_local.device
        if D_full is None:
            D_full = torch.empty((M_total, N), dtype=torch.float16, device=dev)
        even = (M_total % world == 0)
        use_streams = dev.type == "cuda"
        comm = None
        if use_streams:
            comm = self.comm_stream or torch.cuda.Stream(device=dev)
            self.comm_stream = comm
        max_rows = -(-M_total // world)
        for (a, b) in chunk_bounds(max_rows if not even else rows, self.chunks):
            # local compute of rows [a, b) of this rank's shard (clipped for the short ranks)
            la, lb = min(a, rows), min(b, rows)
            if lb > la:
                self.gemm_fn(A_local[la:lb], s1_local[la:lb], D_full[r0 + la : r0 + lb])
            outs = []
            for p in range(world):
                p0, p1 = shard_rows(M_total, world, p)
                pa, pb = min(a, p1 - p0), min(b, p1 - p0)
                outs.append((p0 + pa, p0 + pb))
            if even:
                views = [D_full[s:e] for (s, e) in outs]
                src = D_full[r0 + la : r0 + lb]
                self._all_gather(views, src, comm, use_streams)
            else:
                # ragged shards: gather fixed-size padded pieces, then copy the valid rows out
                width = b - a
                pad = torch.zeros((width, N), dtype=torch.float16, device=dev)
                if lb > la:
                    pad[: lb - la] = D_full[r0 + la : r0 + lb]
                bufs = [torch.empty_like(pad) for _ in range(world)]
                self._all_gather(bufs, pad, comm, use_streams)
                if use_streams:
                    comm.synchronize()
                for p, (s, e) in enumerate(outs):
                    if e > s and p != rank:
                        D_full[s:e] = bufs[p][: e - s]
        if use_streams:
            torch.cuda.current_stream(dev).wait_stream(comm)
        return D_full

    def _all_gather(self, outs, src, comm, use_streams):
        if use_streams:
            comm.wait_stream(torch.cuda.current_stream(src.device))
            with torch.cuda.stream(comm):
                dist.all_gather(outs, src, group=self.group)
        else:
            dist.all_gather(outs, src, group=self.group)
