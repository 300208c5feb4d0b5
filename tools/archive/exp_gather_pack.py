#!/usr/bin/env python3
"""What an N > 1 step of bench.py does beyond an N = 1 step, timed on one GPU at the headline's size: the sending side of the
gather in its two forms (compact words + exceptions; count bytes + text-id bytes + int32 positions) and the receiving side's
split of one arrived shard (gdx_compact_split_hits_dev).  usage: python tools/exp_gather_pack.py [reps]  -> one JSON line"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from genedex_amd import alphabet  # noqa: E402
from genedex_amd import dist as gdist  # noqa: E402
from genedex_amd.device import (DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths,  # noqa: E402
                                synth_text)
from genedex_amd.index import build_options  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
total = int(os.environ.get("GDX_EXP_TOTAL", 3_100_000_000))
nq = int(os.environ.get("GDX_EXP_NQ", 100_000_000))
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32",
                                     options=build_options(**bench.SEED_INDEX))
eng = DeviceEngine(index)
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


res = {"queries": nq}
for wire in ("compact", "arrays"):
    os.environ["GDX_BENCH_GATHER"] = wire
    runner = bench.StepRunner(torch, eng, q, nq, True, "records", n_slots=1)
    runner.size()
    runner.step(0, False)
    torch.cuda.synchronize()
    gather, pack, nbytes = bench.make_gather(torch, gdist, runner, dev, True)
    if wire == "arrays":
        runner.step(0, False)  # (make_gather replaced the hit buffers)
    res[wire] = {"bytes_per_query": nbytes / nq, "sender_ms": timed(lambda: pack(0)), "step_ms": timed(lambda: runner.step(0, False))}
    if wire == "compact":
        res[wire]["exceptions"] = gather.exceptions
        words = gather.slots[0][0]
        ids, pos = gather.root_ids[0][0], gather.root_pos[0][0]  # (world 1: nothing arrives, the split is called by hand)
        res[wire]["receiver_split_ms_per_shard"] = timed(lambda: eng.compact_split_hits(words, nq, ids, pos))
        cnt, hh = gdist.expand_split_results(ids, pos, *gather.slots[0][1:], nq)
        res[wire]["expanded_equals_step_output"] = bool(torch.equal(hh, runner.hits[0][: runner.total_hits])) and \
            bool(torch.equal(cnt.to(torch.int32), runner.counts(runner.outs[0])))
        del cnt, hh
    del gather, pack, runner
    torch.cuda.empty_cache()
print(json.dumps(res))
