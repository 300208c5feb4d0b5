"""Multi-GPU plumbing: the index is replicated, queries are sharded contiguously, results are gathered.

One process per GPU under torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  The data path has no collective: every rank searches its own shard against its own
replica.  The only exchange is the result gather to rank 0 (fixed-size counts / intervals, then the
variable-size hit arrays, padded to the largest shard); there is no reduction, so nothing here is
bound by a ring's per-link bandwidth.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank; shards differ in size by at most one item."""
    lo = n_items * rank // world
    hi = n_items * (rank + 1) // world
    return lo, hi


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def gather_fixed(t: torch.Tensor, dst: int = 0):
    """Equal-sized per-rank tensors -> list of world tensors on dst (None elsewhere)."""
    rank, n = world()
    if n == 1:
        return [t]
    out = [torch.empty_like(t) for _ in range(n)] if rank == dst else None
    dist.gather(t, out, dst=dst)
    return out


def gather_variable(t: torch.Tensor, length: int, dst: int = 0):
    """Per-rank tensors whose first `length` rows are valid -> list of trimmed tensors on dst."""
    rank, n = world()
    if n == 1:
        return [t[:length]]
    lens = torch.tensor([length], dtype=torch.int64, device=t.device)
    all_lens = [torch.zeros_like(lens) for _ in range(n)]
    dist.all_gather(all_lens, lens)
    max_len = max(int(x.item()) for x in all_lens)
    shape = (max_len,) + tuple(t.shape[1:])
    if t.shape[0] >= max_len:
        padded = t[:max_len].contiguous()
    else:
        padded = torch.zeros(shape, dtype=t.dtype, device=t.device)
        padded[:length] = t[:length]
    out = [torch.empty(shape, dtype=t.dtype, device=t.device) for _ in range(n)] if rank == dst else None
    dist.gather(padded, out, dst=dst)
    if rank != dst:
        return None
    return [o[: int(l.item())] for o, l in zip(out, all_lens)]


def max_over_ranks(value: float, device) -> float:
    rank, n = world()
    if n == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
