#!/usr/bin/env python3
"""One index, every BASELINE configuration: builds the hg38-scale index with the given build options (none = the library's
defaults, i.e. the default shape) and times on it the count + locate step of 100 M len-50 reads (ASCII + offsets and 2-bit
uniform), exact intervals of the same reads, and workload 5 (50 M reads of 20..150 symbols) fused and through the cursor API.
usage: python tools/exp_default_shape.py [steps] [key=value build options ...]  -> one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import (DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths,  # noqa: E402
                                synth_text)
from genedex_amd.index import build_options  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
opts = {}
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    opts[k] = int(v)
total = int(os.environ.get("GDX_EXP_TOTAL", 3_100_000_000))
nq = int(os.environ.get("GDX_EXP_NQ", 100_000_000))
what = os.environ.get("GDX_EXP_WHAT", "step,exact,mixed").split(",")
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
t0 = time.time()
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32", options=build_options(**opts))
eng = DeviceEngine(index)
res = {"steps": steps, "build_s": time.time() - t0, "index_gb": index.info.device_bytes / 1e9, "aux": eng.aux_info(), "options": opts}
print(json.dumps(res), file=sys.stderr, flush=True)
full = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)
if "step" in what:
    ref = None
    for form in ("ascii", "packed+uniform"):
        q = full if form == "ascii" else full.as_packed(index).as_uniform(50)
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
        runner.check_totals()
        runner.widen_offsets()
        got = (runner.outs[0]["hit_offsets"].clone(), runner.hits[0][: runner.total_hits].clone())
        if ref is None:
            ref = got
        elif not (torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])):
            raise SystemExit("PARITY FAILURE: the two input forms give other offsets or hits")
        res["step_" + form] = {"ms_per_step": ms, "search_ms": runner.mean_ms(runner.ev_search), "locate_ms": runner.mean_ms(runner.ev_locate),
                               "hits": runner.total_hits, "Gq_per_s": nq / ms / 1e6}
        print(json.dumps({form: res["step_" + form]}), file=sys.stderr, flush=True)
        del runner, q
        torch.cuda.empty_cache()
    del ref, got


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


if "exact" in what:
    out = eng.alloc_outputs(nq)
    res["exact_len50_ms"] = timed(lambda: eng.search(full, out))
    print(json.dumps({"exact_len50_ms": res["exact_len50_ms"]}), file=sys.stderr, flush=True)
    del out
del full
torch.cuda.empty_cache()
if "mixed" in what:
    res["mixed"] = bench.mixed_length_secondary(torch, eng, io_text, lengths)
print(json.dumps(res))
