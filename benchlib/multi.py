"""benchlib.multi -- N > 1: the gather to rank 0 (three wires), the sharded batch of BASELINE configs[3], its check against the one-rank output (split out of bench.py in round 6; bench.py re-exports everything)."""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .common import *  # noqa: F401,F403
from .pmc import *  # noqa: F401,F403
from .line import *  # noqa: F401,F403
from .baseline import *  # noqa: F401,F403

__all__ = ['report_strong_scaling', 'make_gather', 'make_compact_gather', 'make_bitmap_gather', 'gathered_shards', 'strong_scaling']

def report_strong_scaling(result, wl):
    """N > 1: BASELINE.json configs[3] is ONE batch of nq reads sharded over the ranks, so `value`, `ms_per_step`,
    `scaling` and `config.workload` become those of the strong-scaling measurement; the every-rank-its-own-batch number
    that was timed first moves to `weak_scaling`."""
    st = result["strong_scaling"]
    result["weak_scaling"] = {"value": result["value"], "ms_per_step": result["ms_per_step"], "unit": "queries/s",
                              "queries_per_gpu": result["config"]["queries_per_gpu"],
                              "gathered_bytes_per_rank_and_step": result["config"]["gathered_bytes_per_rank_and_step"]}
    n = result["n_gpus"]
    result["value"], result["ms_per_step"], result["scaling"] = st["value"], st["ms_per_step"], "strong"
    result["steps"] = st["steps"]
    c = result["config"]
    c["workload"] = (f"{wl['short']}: ONE batch of {st['queries_total']} reads sharded over {n} GPUs (BASELINE configs[3]), index "
                     f"{c['index_gb_per_replica']:.0f} GB replicated, results gathered to rank 0 over RCCL")
    c["queries_per_gpu"] = st["queries_this_rank"]
    c["queries_total"] = st["queries_total"]
    c["gathered_bytes_per_rank_and_step"] = st["gathered_bytes_per_rank_and_step"]
    c["gather_wire"] = st.get("gather_wire", c.get("gather_wire"))
    result["parity"]["shards_equal_single_rank_output"] = st.get("shards_equal_single_rank_output")
    result["results_sharded"] = st.get("results_sharded")


def make_gather(torch, gdist, runner, dev, do_locate):
    """Pads the hit buffers to the largest shard, picks the count type, returns (PipelinedGather, count_of, bytes)."""
    nq = runner.nq
    o = runner.outs[0]
    max_count = int(runner.counts(o).max().item()) if nq else 0
    max_count = gdist.max_int_over_ranks(max_count, dev)
    # (torch's RCCL process group maps int8 / uint8 / int32 / int64 and the float types only: no 16-bit integers)
    count_dtype = torch.uint8 if max_count <= 0xff else torch.int32
    max_hits = gdist.max_int_over_ranks(runner.total_hits, dev)
    max_nq = gdist.max_int_over_ranks(nq, dev)
    # On an index with a seed table the search's compact results travel as they are, with the few queries that have more to
    # say beside them (make_compact_gather) -- whenever that is fewer bytes than the arrays below (it is not on a text of repeats)
    # ... or as a bit per read + 4 bytes per FOUND read (make_bitmap_gather: 3.73 bytes per read where nine in ten are found)
    wire = os.environ.get("GDX_BENCH_GATHER", "auto")
    if do_locate and runner.use_compact and int(runner.eng.index.num_texts()) <= 256 and wire in ("auto", "compact", "bitmap"):
        n_exc, n_exc_hits = gdist.exception_sizes(o["compact"], o["hit_offsets"], nq)
        cap_q = max(gdist.max_int_over_ranks(n_exc, dev), 1)
        cap_h = max(gdist.max_int_over_ranks(n_exc_hits, dev), 1)
        n_found = int((o["compact"][:nq] >= 0).sum().item()) + int((o["compact"][:nq] < -2).sum().item()) if nq else 0
        cap_f = max(gdist.max_int_over_ranks(n_found, dev), 1)
        arrays_bytes = max(max_nq, 1) * (1 if max_count <= 0xff else 4) + 5 * max(max_hits, 1)
        compact_bytes = 4 * max(max_nq, 1) + 4 * cap_q + 5 * cap_h + 8
        layout = gdist.WireLayout(max(max_nq, 1), cap_f, cap_q, cap_h)
        exc = {"queries": n_exc, "hits": n_exc_hits, "found": n_found}
        if wire == "bitmap" or (wire == "auto" and layout.nbytes < min(compact_bytes, arrays_bytes)):
            return make_bitmap_gather(torch, gdist, runner, dev, layout, exc)
        if wire == "compact" or (wire == "auto" and compact_bytes < arrays_bytes):
            return make_compact_gather(torch, gdist, runner, dev, max_nq, cap_q, cap_h, exc)
    runner.hits = [torch.zeros((max(max_hits, 1), 2), dtype=torch.int32, device=dev) for _ in range(runner.n_slots)]
    cnts = [torch.zeros(max(max_nq, 1), dtype=count_dtype, device=dev) for _ in range(runner.n_slots)]
    # Hits travel as two arrays -- text ids as bytes when the collection has at most 256 texts, positions as int32 -- instead
    # of (int32, int32) pairs: 5 instead of 8 bytes per hit over the one xGMI link every rank has to rank 0.  With the seed
    # index a rank produces ~21 G results/s; as pairs that would be 172 GB/s per link, more than a link carries (DESIGN.md
    # section 6), and the gather rather than the kernels would bound the step.  Lossless: rank 0 holds the same hits.
    n_texts = int(runner.eng.index.num_texts())
    split = do_locate and n_texts <= 256
    if split:
        ids = [torch.zeros(max(max_hits, 1), dtype=torch.uint8, device=dev) for _ in range(runner.n_slots)]
        pos = [torch.zeros(max(max_hits, 1), dtype=torch.int32, device=dev) for _ in range(runner.n_slots)]
        gather = gdist.PipelinedGather([[c, i, p] for c, i, p in zip(cnts, ids, pos)], dst=0)
    else:
        gather = gdist.PipelinedGather([[c, h] if do_locate else [c] for c, h in zip(cnts, runner.hits)], dst=0)

    counts32 = [torch.empty(max(nq, 1), dtype=torch.int32, device=dev) for _ in range(runner.n_slots)] if runner.use_rec else None

    def count_of(slot):
        # per-query counts out of the step's results in the gather's count type (copy_ narrows); with records one pass of
        # gdx_locate_many_unpack[_compact]_dev instead of torch arithmetic over the strided 16-byte records -- this runs
        # inside every timed step of an N > 1 run, which an N = 1 run does not have
        if runner.use_rec:
            o_ = runner.outs[slot]
            runner.eng.unpack_records(o_["rec"], nq, counts32[slot], None, compact=o_["compact"])
            cnts[slot][:nq].copy_(counts32[slot][:nq])
        else:
            cnts[slot][:nq].copy_(runner.counts(runner.outs[slot]))
        if split:
            th = min(runner.total_hits, ids[slot].numel())
            h_ = runner.hits[slot]
            ids[slot][:th].copy_(h_[:th, 0])  # (text ids < 256: copy_ narrows)
            pos[slot][:th].copy_(h_[:th, 1])

    if split:
        nbytes = cnts[0].numel() * cnts[0].element_size() + 5 * ids[0].numel()
    else:
        nbytes = cnts[0].numel() * cnts[0].element_size() + (runner.hits[0].numel() * 4 if do_locate else 0)
    gather.hits_are_split = split
    return gather, count_of, nbytes


def make_compact_gather(torch, gdist, runner, dev, max_nq, cap_q, cap_h, exceptions):
    """The gather of a count + locate step on an index with a seed table: the search's compact results travel as they are
    -- 4 bytes per query: the text position of its only hit, "none" or "see the exceptions" -- beside the counts and hits
    of the exceptions (dist.pack_exceptions, sized by the sizing pass: the steps repeat the same batch).  A link into rank 0
    carries one direction of one xGMI link's 153.6 GB/s, so at ~21 G results/s per rank the bytes per result decide the
    step: 4.0x instead of 5.5.  Rank 0 turns every arrived shard into text id + position per query
    (gdx_compact_split_hits_dev, one kernel per shard, enqueued when the gather is acquired) -- inside the timed region."""
    nq = runner.nq
    o = runner.outs[0]
    n = max(max_nq, 1)
    slots = [[torch.full((n,), -1, dtype=torch.int32, device=dev), torch.zeros(cap_q, dtype=torch.int32, device=dev),
              torch.zeros(cap_h, dtype=torch.uint8, device=dev), torch.zeros(cap_h, dtype=torch.int32, device=dev),
              torch.zeros(2, dtype=torch.int32, device=dev)] for _ in range(runner.n_slots)]
    rank, world = gdist.world()
    root_ids = root_pos = None
    if rank == 0:
        root_ids = [[torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(runner.n_slots)]
        root_pos = [[torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(world)] for _ in range(runner.n_slots)]

    def on_gathered(slot, own=False):
        # (rank 0's own shard is on its device in every form already: only a check asks for it in this one)
        for r, words in enumerate(gather.gathered(slot)[0]):
            if (r == 0) == own:
                runner.eng.compact_split_hits(words, n, root_ids[slot][r], root_pos[slot][r])

    gather = gdist.PipelinedGather(slots, dst=0, on_gathered=on_gathered)

    # the search writes its compact results straight into the buffer that travels
    for s_, o_ in zip(slots, runner.outs):
        s_[0][:nq].copy_(o_["compact"][:nq])
        o_["compact"] = s_[0]
    listed = [(torch.zeros(cap_q, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
              for _ in range(runner.n_slots)]

    def pack(slot):
        o_ = runner.outs[slot]
        words, exc_cnt, exc_ids, exc_pos, meta = slots[slot]
        runner.eng.compact_exceptions(words, nq, *listed[slot])
        gdist.pack_exceptions(words, o_["hit_offsets"], runner.hits[slot], nq, exc_cnt, exc_ids, exc_pos, meta, listed[slot])

    gather.hits_are_split = True
    gather.compact_wire = True
    gather.root_ids, gather.root_pos = root_ids, root_pos
    gather.split_own = lambda slot: on_gathered(slot, own=True)
    gather.exceptions = exceptions
    return gather, pack, 4 * n + 4 * cap_q + 5 * cap_h + 8


def make_bitmap_gather(torch, gdist, runner, dev, layout, exceptions):
    """The gather of a count + locate step as a bit per read + the text positions of the found reads (gdx_wire_pack_dev, three
    launches on the rank; dist.WireLayout: everything a rank sends lies in ONE byte buffer, one gather per step): 0.125 + 4 x
    the found fraction bytes per read -- 3.73 where nine reads in ten are found -- instead of the 4 of the compact words.  A
    link into rank 0 carries one direction of an xGMI link, and at ~28 G results/s per rank the bytes per result decide the step
    (DESIGN.md section 6).  Rank 0 turns every arrived shard into text id + position per read (gdx_wire_split_dev, one kernel
    per shard, enqueued when the gather is acquired) -- inside the timed region."""
    nq = runner.nq
    n = layout.n_max
    bufs = [torch.zeros(layout.nbytes, dtype=torch.uint8, device=dev) for _ in range(runner.n_slots)]
    views = [layout.views(b) for b in bufs]
    ws = [torch.empty(max(runner.eng.wire_pack_workspace_bytes(nq), 16), dtype=torch.uint8, device=dev) for _ in range(runner.n_slots)]
    rank, world = gdist.world()
    root_ids = root_pos = None
    if rank == 0:
        root_ids = [[torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(runner.n_slots)]
        root_pos = [[torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(world)] for _ in range(runner.n_slots)]

    def on_gathered(slot, own=False):
        for r, buf in enumerate(gather.gathered(slot)[0]):
            if (r == 0) == own:
                runner.eng.wire_split(layout.views(buf), n, root_ids[slot][r], root_pos[slot][r])

    gather = gdist.PipelinedGather([[b] for b in bufs], dst=0, on_gathered=on_gathered)

    def pack(slot):
        o_ = runner.outs[slot]
        runner.eng.wire_pack(o_["compact"], o_["hit_offsets"], runner.hits[slot], nq, views[slot], ws[slot])

    gather.hits_are_split = True
    gather.compact_wire = True
    gather.wire_name = "bitmap"
    gather.root_ids, gather.root_pos = root_ids, root_pos
    gather.split_own = lambda slot: on_gathered(slot, own=True)
    gather.exceptions = exceptions
    # (exception counts, text ids, positions and the true numbers of one received shard, as expand_split_results takes them)
    gather.exception_parts = lambda slot, r: [layout.views(gather.gathered(slot)[0][r])[k] for k in ("exc_cnt", "exc_ids", "exc_pos", "meta")]
    gather.payload_bytes = layout.payload_bytes(nq, exceptions["found"], exceptions["queries"], exceptions["hits"])
    return gather, pack, layout.nbytes


def gathered_shards(torch, gdist, gather, slot, shard_len, sizes, do_locate):
    """rank 0: (counts, hits or None) of the gathered shards of `slot`, concatenated, as a one-rank run would hold them"""
    parts = gather.gathered(slot)
    world = len(shard_len)
    if getattr(gather, "compact_wire", False):
        gather.split_own(slot)
        cnts, hits = [], []
        for r, (a, b) in enumerate(shard_len):
            exc = gather.exception_parts(slot, r) if hasattr(gather, "exception_parts") else [parts[k][r] for k in (1, 2, 3, 4)]
            c, h = gdist.expand_split_results(gather.root_ids[slot][r], gather.root_pos[slot][r], *exc, b - a)
            if h.shape[0] != sizes[r]:
                raise SystemExit(f"PARITY FAILURE: shard {r} arrived with {h.shape[0]} hits, its rank located {sizes[r]}")
            cnts.append(c)
            hits.append(h)
        return torch.cat(cnts), torch.cat(hits)
    cnt_cat = torch.cat([parts[0][r][: b - a] for r, (a, b) in enumerate(shard_len)])
    if do_locate and getattr(gather, "hits_are_split", False):  # (text ids as bytes, positions as int32: back to pairs)
        hit_cat = torch.cat([torch.stack([parts[1][r][: sizes[r]].to(torch.int32), parts[2][r][: sizes[r]]], dim=1)
                             for r in range(world)])
    else:
        hit_cat = torch.cat([parts[1][r][: sizes[r]] for r in range(world)]) if do_locate else None
    return cnt_cat, hit_cat


def strong_scaling(torch, gdist, eng, io_text, lengths, wl, nq_total, do_locate, args, rank, world, dev):
    """BASELINE.json configs[3]: the N = 1 batch (seed 43) split into `world` contiguous shards (dist.shard_range), one
    per rank, results gathered to rank 0; value = nq_total / max-over-ranks step time.  Rank 0 also runs the whole
    batch alone once and requires the concatenated shard results to equal it bit for bit."""
    from genedex_amd.device import DeviceQueries

    full = DeviceQueries.synth(io_text, lengths, nq_total, wl["len_min"], wl["len_max"], wl["sampled_ppm"], seed=43)
    # rank 0's shard relative to the others': from the rate this run's links deliver into rank 0 (a probe gather) and the
    # one-GPU costs of a step and of the root's split (dist.root_weight_for)
    link_rate = gdist.gather_rate_probe(dev)
    seeded = eng.index.seed_info()["k"] != 0 and do_locate
    root_weight = args.root_weight if args.root_weight is not None else \
        gdist.root_weight_for(world, nq_total, link_rate, gdist.WIRE_BYTES_PER_READ if seeded else 5.5)
    lo, hi = gdist.shard_range(nq_total, rank, world, root_weight)
    # (a rank holds its shard as a batch of its own: the form --input names is made from that)
    shard = input_form(full.copy_slice(lo, hi) if args.input != "ascii" else full.slice(lo, hi), eng.index, args, wl)
    runner = StepRunner(torch, eng, shard, hi - lo, do_locate, args.path, hint=not args.no_hint, n_slots=2)
    runner.size()
    gather, count_of, nbytes = make_gather(torch, gdist, runner, dev, do_locate)
    steps = max(args.steps, 1)
    elapsed, last = timed_steps(torch, gdist, runner, steps, args.warmup, dev, gather, count_of)
    ms = elapsed / steps * 1e3
    res = {"scaling": "strong", "value": nq_total / (ms / 1e3), "unit": "queries/s", "ms_per_step": ms,
           "queries_total": nq_total, "queries_this_rank": hi - lo, "steps": steps, "root_weight": root_weight, "gather_probe_GBps_per_link": link_rate,
           "kernel_ms_rank0": {"search": runner.mean_ms(runner.ev_search), "locate": runner.mean_ms(runner.ev_locate)},
           "gathered_bytes_per_rank_and_step": nbytes}
    runner.check_totals()
    # the same sharded step with the results LEFT on their GPUs (no gather): what the kernels and launches of N ranks give; the
    # gather above adds what one direction of the links into rank 0 carries
    e_ng, _ = timed_steps(torch, gdist, runner, steps, args.warmup, dev)
    runner.check_totals()
    res["results_sharded"] = {"value": nq_total / (e_ng / steps), "unit": "queries/s", "ms_per_step": e_ng / steps * 1e3,
                              "what": "the sharded step without the gather: every rank's offsets and hits stay in its own HBM"}
    # bit-exactness: concatenated shards == the one-rank output
    sizes = gdist.gather_ints(runner.total_hits, dev)
    if rank == 0:
        shard_len = [gdist.shard_range(nq_total, r, world, root_weight) for r in range(world)]
        cnt_cat, hit_cat = gathered_shards(torch, gdist, gather, last, shard_len, sizes, do_locate)
        res["gather_wire"] = getattr(gather, "wire_name", "compact" if getattr(gather, "compact_wire", False) else "arrays")
        del gather, runner
        torch.cuda.empty_cache()
        single = StepRunner(torch, eng, input_form(full, eng.index, args, wl), nq_total, do_locate, args.path, hint=not args.no_hint)
        single.size()
        single.step(0, False)
        torch.cuda.synchronize()
        same_counts = bool(torch.equal(cnt_cat.to(torch.int64), single.counts(single.outs[0]).to(torch.int64)))
        same_hits = bool(torch.equal(hit_cat, single.hits[0][: single.total_hits])) if do_locate else None
        res["shards_equal_single_rank_output"] = {"counts": same_counts, "hits": same_hits}
        if not same_counts or same_hits is False:
            raise SystemExit(f"PARITY FAILURE: sharded results differ from the one-rank output: {res}")
    return res
