#!/usr/bin/env python3
"""Does the search time of the seed index depend on where its structures land in HBM?  One process: build, time the search,
then rebuild the auxiliary structures (fresh allocations, identical contents) and time again, a few times over.
usage: python tools/exp_placement.py [rebuilds]  -> one JSON line"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import (DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths,  # noqa: E402
                                synth_text)
from genedex_amd.index import build_options  # noqa: E402

rebuilds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
total, nq = 3_100_000_000, 100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32",
                                     options=build_options(**bench.SEED_INDEX))
eng = DeviceEngine(index)
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)
rec, cmp_ = eng.alloc_records(nq), eng.alloc_compact(nq)


def search_ms(reps=10):
    eng.locate_search(q, rec, compact=cmp_)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.locate_search(q, rec, compact=cmp_)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return [min(ts), sorted(ts)[len(ts) // 2], max(ts)]


out = {"first_build": search_ms()}
hold = []
for i in range(rebuilds):
    if i % 2 == 1:  # every other time with a different heap state: a few GB held elsewhere
        hold.append(torch.empty(3_000_000_000 + 7_000_000 * i, dtype=torch.uint8, device=dev))
    index.rebuild_aux(**bench.SEED_INDEX)
    out[f"rebuild_{i}"] = search_ms()
out["again"] = search_ms()
print(json.dumps(out))
