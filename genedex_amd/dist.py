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


class PipelinedGather:
    """Result gather to `dst` that overlaps with the next batch's kernels.

    Two slots of equal-shaped result tensors (double buffering): `submit(slot)` enqueues an asynchronous
    gather of the slot's tensors (RCCL runs it on its own stream after the kernels that produced them);
    `acquire(slot)` must be called before the slot's tensors are overwritten and waits for the gather that
    last read them; `drain()` waits for everything.  Per-rank payloads are padded to the same shape up front
    (`max_rows`), so no size exchange happens inside the timed loop.
    """

    def __init__(self, slots, dst: int = 0):
        # slots: list (one per slot) of lists of tensors; same shapes/dtypes on every rank
        self.slots = slots
        self.dst = dst
        self.rank, self.world = world()
        self.pending = [[] for _ in slots]
        self.recv = None
        if self.world > 1 and self.rank == dst:
            self.recv = [[[torch.empty_like(t) for _ in range(self.world)] for t in slot] for slot in slots]

    def acquire(self, slot: int) -> None:
        for w in self.pending[slot]:
            w.wait()
        self.pending[slot] = []

    def submit(self, slot: int) -> None:
        if self.world == 1:
            return
        for k, t in enumerate(self.slots[slot]):
            out = self.recv[slot][k] if self.rank == self.dst else None
            self.pending[slot].append(dist.gather(t, out, dst=self.dst, async_op=True))

    def drain(self) -> None:
        for slot in range(len(self.slots)):
            self.acquire(slot)

    def gathered(self, slot: int):
        """On dst: per tensor of the slot the list of world tensors (after acquire/drain)."""
        return self.recv[slot] if self.recv is not None else [[t] for t in self.slots[slot]]


def max_int_over_ranks(value: int, device) -> int:
    rank, n = world()
    if n == 1:
        return value
    t = torch.tensor([value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def gather_ints(value: int, device):
    """one integer per rank -> list of world integers on every rank"""
    rank, n = world()
    if n == 1:
        return [int(value)]
    t = torch.tensor([value], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(n)]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]
