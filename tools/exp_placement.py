#!/usr/bin/env python3
"""Does where the seed table lands in HBM move the headline's kernel?  One process, the default index rebuilt several times
(rebuild_aux frees and allocates every auxiliary structure again: the physical pages differ), the same 100 M reads timed after
each rebuild (2-bit uniform reads: the search alone decides the step; GDX_EXP_FORM=ascii: IO symbols + offsets, the headline's form).
usage: python tools/exp_placement.py [rebuilds] [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths, synth_text  # noqa: E402

rebuilds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
total, nq = 3_100_000_000, 100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
q_first = os.environ.get("GDX_EXP_QFIRST") == "1"  # (the reads allocated before the index is built: other physical pages)
if q_first:
    full = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32")
eng = DeviceEngine(index)
if not q_first:
    full = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)
form = os.environ.get("GDX_EXP_FORM", "packed")
q = full.as_packed(index).as_uniform(50) if form == "packed" else full
hog = []
for r in range(rebuilds):
    runner = bench.StepRunner(torch, eng, q, nq, True, "records")
    runner.size()
    for _ in range(3):
        runner.step(0, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.step(0, True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"rebuild": r, "ms_per_step": ms, "search_ms": runner.mean_ms(runner.ev_search)}), flush=True)
    del runner
    torch.cuda.empty_cache()
    if r + 1 < rebuilds:
        # (another few GB held between rebuilds, so that the next table cannot simply take the pages the last one left)
        hog.append(torch.empty((3 + r) << 30, dtype=torch.uint8, device=dev))
        index.rebuild_aux()
