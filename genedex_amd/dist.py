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


def shard_range(n_items: int, rank: int, world: int, root_weight: float = 1.0):
    """Contiguous shard [lo, hi) of rank; shards differ in size by at most one item.
    root_weight != 1: rank 0's shard is that fraction of every other rank's -- the root of a gather also splits what it
    receives (root_weight_for), so a smaller shard of its own evens out what the GPUs have to do."""
    if root_weight == 1.0 or world == 1:
        lo = n_items * rank // world
        hi = n_items * (rank + 1) // world
        return lo, hi
    total = root_weight + (world - 1)
    cut = lambda r: 0 if r == 0 else min(n_items, int(n_items * (root_weight + (r - 1)) / total))  # noqa: E731
    return cut(rank), (n_items if rank == world - 1 else cut(rank + 1))


# One-GPU measurements behind the shard sizes of a gathered batch (profiles/r05/shard_step.json; an MI355X, the headline
# index): a count + locate step costs STEP_FIXED_MS + STEP_PS_PER_READ per read, the root's split of a received read
# (gdx_wire_split_dev) SPLIT_PS_PER_READ, a read travels as WIRE_BYTES_PER_READ bytes, and a link into the root carries one
# direction of an xGMI link.
STEP_FIXED_MS, STEP_PS_PER_READ, SPLIT_PS_PER_READ, WIRE_BYTES_PER_READ, XGMI_ONE_DIRECTION_GBPS = 0.085, 30.6, 2.55, 3.725, 76.8


def root_weight_for(world: int, n_reads: int = 100_000_000, link_GBps: float = XGMI_ONE_DIRECTION_GBPS,
                    bytes_per_read: float = WIRE_BYTES_PER_READ) -> float:
    """rank 0's shard relative to every other rank's such that the root's work on a step -- its own shard's step and the
    split of the world - 1 shards it receives -- takes as long as a link needs for one of those shards (the links bound a
    gathered step: DESIGN.md section 6).  With s reads per other rank, s = n / (world - 1 + w):
        STEP_FIXED + s (w STEP + (world - 1) SPLIT) = s L,   L = bytes per read / link rate
    Slow links (or a few fast GPUs) give the root MORE than the others, many ranks less: 1.43 / 1.22 / 0.79 at 2 / 4 / 8
    ranks with the numbers above.  Clamped to [0.25, 2]."""
    if world <= 1:
        return 1.0
    fixed_ps = STEP_FIXED_MS * 1e9 / max(n_reads, 1)  # the fixed part of a step, spread over the batch's reads
    link_ps = bytes_per_read / max(link_GBps, 1e-3) * 1e3
    w = (link_ps - (world - 1) * (SPLIT_PS_PER_READ + fixed_ps)) / (STEP_PS_PER_READ + fixed_ps)
    return float(min(2.0, max(0.25, w)))


def gather_rate_probe(device, nbytes: int = 64 << 20, reps: int = 3) -> float:
    """GB/s one rank's buffer reaches rank 0 at while all ranks send at once (what a gathered step sees of a link); the
    same number on every rank.  A few gathers of `nbytes` per rank, timed on the host around a synchronisation."""
    import time

    rank, n = world()
    if n == 1:
        return XGMI_ONE_DIRECTION_GBPS
    t = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    out = [torch.empty_like(t) for _ in range(n)] if rank == 0 else None
    best = None
    for _ in range(reps + 1):  # (the first one also sets the connections up)
        dist.barrier()
        if t.is_cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.gather(t, out, dst=0)
        if t.is_cuda:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    rate = torch.tensor([nbytes / best / 1e9], dtype=torch.float64, device=device)
    dist.broadcast(rate, src=0)  # (rank 0 waits for everybody's bytes: its time is the gather's)
    return float(rate.item())


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

    def __init__(self, slots, dst: int = 0, on_gathered=None):
        # slots: list (one per slot) of lists of tensors; same shapes/dtypes on every rank
        # on_gathered(slot): called on dst, from acquire(), once the slot's gather has arrived and before its receive
        # buffers can be written again (what dst does with a received batch, e.g. GPU kernels enqueued behind the gather)
        self.slots = slots
        self.dst = dst
        self.on_gathered = on_gathered
        self.rank, self.world = world()
        self.pending = [[] for _ in slots]
        self.recv = None
        if self.world > 1 and self.rank == dst:
            self.recv = [[[torch.empty_like(t) for _ in range(self.world)] for t in slot] for slot in slots]

    def acquire(self, slot: int) -> None:
        arrived = bool(self.pending[slot])
        for w in self.pending[slot]:
            w.wait()
        self.pending[slot] = []
        if arrived and self.on_gathered is not None and self.rank == self.dst:
            self.on_gathered(slot)

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


# ---- compact results on the wire --------------------------------------------------------------------------------------
# A count + locate shard travels to the root as the search's compact results -- one 32-bit word per query: the text position
# of its only hit, COMPACT_NONE, or COMPACT_SEE (gdx.h, gdx_locate_many_search_compact_dev) -- plus, for the queries that
# say COMPACT_SEE, their counts and hits ("exceptions", in query order).  4 bytes per query on a text without repeats
# instead of a count byte and 5 bytes per hit; the root turns the words into text id + position with one kernel
# (gdx_compact_split_hits_dev).  The functions below are the tensor plumbing around that; they run on any device.
COMPACT_NONE = -1  # (the int32 views of 0xffffffff / 0xfffffffe)
COMPACT_SEE = -2


def exception_sizes(compact: torch.Tensor, hit_offsets: torch.Tensor, nq: int):
    """(number of queries that say COMPACT_SEE, number of their hits); synchronises -- for sizing the buffers"""
    if nq == 0:
        return 0, 0
    see = compact[:nq] == COMPACT_SEE
    cnt = hit_offsets[1:nq + 1] - hit_offsets[:nq]
    return int(see.sum().item()), int((cnt * see).sum().item())


def pack_exceptions(compact: torch.Tensor, hit_offsets: torch.Tensor, hits: torch.Tensor, nq: int, exc_cnt: torch.Tensor,
                    exc_ids: torch.Tensor, exc_pos: torch.Tensor, meta: torch.Tensor, listed=None) -> None:
    """Counts (exc_cnt, int32[cap_q]) and hits (exc_ids uint8[cap_h], exc_pos int32[cap_h]) of the queries whose compact
    result says COMPACT_SEE, in query order; meta (int32[2]) = their true numbers (what exceeds a capacity is dropped and
    shows there).  Fixed shapes, no host synchronisation.
    listed = (queries int32[cap_q] in any order, n int64[1]): the list gdx_compact_exceptions_dev made (DeviceEngine.
    compact_exceptions) -- one streaming pass instead of the mask / select passes of the tensor library."""
    cap_q, cap_h = exc_cnt.numel(), exc_ids.numel()
    dev = compact.device
    if listed is not None:
        queries, n_listed = listed
        valid = torch.arange(cap_q, device=dev) < n_listed
        eq = torch.sort(torch.where(valid, queries.to(torch.int64) & 0xffffffff, nq))[0]
        n_see = n_listed[0]
    else:
        see = compact[:nq] == COMPACT_SEE
        eq = torch.nonzero_static(see, size=cap_q, fill_value=nq)[:, 0]  # (padding: query nq, whose count comes out as 0)
        n_see = see.sum()
    lo = hit_offsets[eq.clamp(max=nq)]
    cnt = hit_offsets[(eq + 1).clamp(max=nq)] - lo
    exc_cnt.copy_(cnt)
    fits = torch.cumsum(cnt, 0) <= cap_h
    kept = cnt * fits
    csum = torch.cumsum(kept, 0)
    n_kept = csum[-1:]
    rep = torch.cat([kept, cap_h - n_kept])  # (one more bucket takes the unused tail of the buffers)
    which = torch.repeat_interleave(torch.arange(cap_q + 1, device=dev), rep, output_size=cap_h)
    lo_ext = torch.cat([lo, lo.new_zeros(1)])
    first = torch.cat([csum - kept, n_kept])
    src = lo_ext[which] + (torch.arange(cap_h, device=dev) - first[which])
    src = src.clamp_(min=0, max=hits.shape[0] - 1)
    exc_ids.copy_(hits[src, 0])
    exc_pos.copy_(hits[src, 1])
    meta[0] = n_see
    meta[1] = cnt.sum()


def expand_split_results(ids: torch.Tensor, pos: torch.Tensor, exc_cnt: torch.Tensor, exc_ids: torch.Tensor,
                         exc_pos: torch.Tensor, meta: torch.Tensor, nq: int):
    """One gathered shard -- per-query text id / position (-1 none, -2 exception) and its exceptions -- in the form a
    one-rank run produces: (counts int64[nq], hits int32[total, 2]).  For checks; synchronises."""
    n_exc, n_exc_hits = int(meta[0].item()), int(meta[1].item())
    if n_exc > exc_cnt.numel() or n_exc_hits > exc_ids.numel():
        raise ValueError(f"exceptions ({n_exc} queries, {n_exc_hits} hits) exceed the buffers they travel in "
                         f"({exc_cnt.numel()}, {exc_ids.numel()})")
    dev = pos.device
    p = pos[:nq].to(torch.int64)
    see = p == COMPACT_SEE
    if int(see.sum().item()) != n_exc:
        raise ValueError("the number of exceptions differs from the number of queries that point to them")
    cnt = (p >= 0).to(torch.int64)
    e_cnt = exc_cnt[:n_exc].to(torch.int64)
    cnt[see] = e_cnt
    total = int(cnt.sum().item())
    off = torch.cumsum(cnt, 0) - cnt
    hq = torch.repeat_interleave(torch.arange(nq, device=dev), cnt, output_size=total)  # the query of every hit
    h_id = ids[:nq][hq].to(torch.int32)
    h_pos = pos[:nq][hq].to(torch.int32)
    if n_exc:
        is_exc = see[hq]
        e_rank = torch.cumsum(see.to(torch.int64), 0) - 1
        e_off = torch.cumsum(e_cnt, 0) - e_cnt
        within = torch.arange(total, device=dev) - off[hq]
        src = (e_off[e_rank[hq].clamp(min=0)] + within).clamp(min=0, max=max(n_exc_hits - 1, 0))
        h_id = torch.where(is_exc, exc_ids[src].to(torch.int32), h_id)
        h_pos = torch.where(is_exc, exc_pos[src].to(torch.int32), h_pos)
    return cnt, torch.stack([h_id, h_pos], dim=1)


# ---- the "found bitmap" wire ------------------------------------------------------------------------------------------
# gdx_wire_pack_dev / gdx_wire_split_dev (gdx.h): a bit per read (found: exactly one hit, answered compactly), the found
# reads' text positions back to back, the found reads before every tile of 2048, and the exceptions {read, count} with their
# hits.  3.73 bytes per read where nine in ten are found (the compact words: 4).  Everything a rank sends lies in ONE byte
# buffer -- one gather per step instead of one per array -- laid out by WireLayout, the same on every rank.
WIRE_TILE = 2048


class WireLayout:
    """Where the parts of a shard's wire form lie in its one byte buffer (all offsets 16-byte aligned).
    n_max: reads of the largest shard; cap_found / cap_q / cap_h: capacities for found reads, exceptions, exception hits."""

    FIELDS = (("bitmap", "uint8"), ("tile_found", "int32"), ("found_pos", "int32"), ("exc_q", "int32"), ("exc_cnt", "int32"),
              ("exc_ids", "uint8"), ("exc_pos", "int32"), ("meta", "int32"))

    def __init__(self, n_max: int, cap_found: int, cap_q: int, cap_h: int):
        self.n_max, self.cap_found, self.cap_q, self.cap_h = n_max, max(cap_found, 1), max(cap_q, 1), max(cap_h, 1)
        tiles = (max(n_max, 1) + WIRE_TILE - 1) // WIRE_TILE
        counts = {"bitmap": tiles * (WIRE_TILE // 8), "tile_found": tiles + 2, "found_pos": self.cap_found, "exc_q": self.cap_q,
                  "exc_cnt": self.cap_q, "exc_ids": self.cap_h, "exc_pos": self.cap_h, "meta": 4}
        self.parts, at = {}, 0
        for name, dtype in self.FIELDS:
            nbytes = counts[name] * (1 if dtype == "uint8" else 4)
            self.parts[name] = (at, counts[name], dtype)
            at += (nbytes + 15) // 16 * 16
        self.nbytes = at

    def payload_bytes(self, nq: int, n_found: int, n_exc: int, n_exc_hits: int) -> int:
        """bytes that carry information for a shard of nq reads (the buffer is padded to the capacities)"""
        return (nq + 7) // 8 + 4 * ((nq + WIRE_TILE - 1) // WIRE_TILE + 1) + 4 * n_found + 8 * n_exc + 5 * n_exc_hits + 16

    def views(self, buf: torch.Tensor) -> dict:
        """typed views of the parts of `buf` (uint8, nbytes long, 16-byte aligned storage)"""
        out = {}
        for name, (at, count, dtype) in self.parts.items():
            part = buf[at: at + count * (1 if dtype == "uint8" else 4)]
            out[name] = part if dtype == "uint8" else part.view(torch.int32)
        return out


def wire_pack_reference(compact: torch.Tensor, hit_offsets: torch.Tensor, hits: torch.Tensor, nq: int, v: dict) -> None:
    """What gdx_wire_pack_dev writes, restated with tensor operations on any device (the CPU tests' stand-in for the kernel)."""
    c = compact[:nq].to(torch.int64) & 0xffffffff
    found = c < (COMPACT_SEE & 0xffffffff)
    see = c == (COMPACT_SEE & 0xffffffff)
    tiles = (max(nq, 1) + WIRE_TILE - 1) // WIRE_TILE
    bits = torch.zeros(tiles * WIRE_TILE, dtype=torch.uint8, device=compact.device)
    bits[:nq] = found.to(torch.uint8)
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=compact.device)
    v["bitmap"][: tiles * (WIRE_TILE // 8)] = (bits.view(-1, 8).to(torch.int32) * weights).sum(1).to(torch.uint8)
    per_tile = bits.view(tiles, WIRE_TILE).to(torch.int64).sum(1)
    tf = torch.zeros(tiles + 1, dtype=torch.int64, device=compact.device)
    tf[1:] = torch.cumsum(per_tile, 0)
    v["tile_found"][: tiles + 1] = tf.to(torch.int32)
    fp = c[found]
    n_f = min(int(fp.numel()), v["found_pos"].numel())
    v["found_pos"][:n_f] = fp[:n_f].to(torch.int32)  # (positions beyond 2^31 wrap into the int32 view, bit for bit)
    eq = torch.nonzero(see)[:, 0]
    cnt = (hit_offsets[eq + 1] - hit_offsets[eq]).to(torch.int64)
    n_e = min(int(eq.numel()), v["exc_q"].numel())
    v["exc_q"][:n_e] = eq[:n_e].to(torch.int32)
    v["exc_cnt"][:n_e] = cnt[:n_e].to(torch.int32)
    total_h = int(cnt.sum().item())
    if total_h:
        src = torch.repeat_interleave(hit_offsets[eq].to(torch.int64), cnt) + (torch.arange(total_h, device=compact.device)
                                                                               - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
        n_h = min(total_h, v["exc_ids"].numel())
        v["exc_ids"][:n_h] = hits[src[:n_h], 0].to(torch.uint8)
        v["exc_pos"][:n_h] = hits[src[:n_h], 1].to(torch.int32)
    v["meta"][0], v["meta"][1], v["meta"][2], v["meta"][3] = int(eq.numel()), total_h, int(fp.numel()), 0


def wire_split_reference(v: dict, nq: int, text_starts: torch.Tensor):
    """What gdx_wire_split_dev writes: (uint8 text ids, int32 positions; -1 none, -2 exception) of a received shard.
    text_starts: int64[n_texts + 1], the start of every text in the concatenation (sentinels included)."""
    dev = v["bitmap"].device
    tiles = (max(nq, 1) + WIRE_TILE - 1) // WIRE_TILE
    by = v["bitmap"][: tiles * (WIRE_TILE // 8)].to(torch.int32)
    bits = ((by[:, None] >> torch.arange(8, device=dev, dtype=torch.int32)[None, :]) & 1).reshape(-1)[:nq].to(torch.bool)
    rank = torch.cumsum(bits.to(torch.int64), 0) - 1
    g = (v["found_pos"].to(torch.int64) & 0xffffffff)[rank.clamp(min=0, max=v["found_pos"].numel() - 1)]
    tid = torch.searchsorted(text_starts[1:] - 1, g, right=False).clamp(max=text_starts.numel() - 2)
    ids = torch.where(bits, tid, torch.zeros_like(tid)).to(torch.uint8)
    pos = torch.where(bits, g - text_starts[tid], torch.full_like(g, -1)).to(torch.int32)
    n_exc = min(int(v["meta"][0].item()), v["exc_q"].numel())
    eq = v["exc_q"][:n_exc].to(torch.int64)
    pos[eq] = -2
    ids[eq] = 0
    return ids, pos


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
