#!/usr/bin/env python3
"""A/B of the fast-path search kernels on the headline workload (and on mixed lengths): time per launch and equality
of the search records.  usage: python tools/exp_fast.py [nq]  -> one JSON line"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import (DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths,  # noqa: E402
                                synth_text)

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
total = 3_100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32")
eng = DeviceEngine(index)
out = {"nq": nq}
for name, lo, hi, n in (("len50", 50, 50, nq), ("len20_150", 20, 150, nq // 2)):
    q = DeviceQueries.synth(io_text, lengths, n, lo, hi, 900_000, seed=43)
    ref = None
    res = {}
    for kind in (1, 0):
        index.set_query_options(search_fast=kind)
        rec = eng.alloc_records(n)
        eng.locate_search(q, rec)
        torch.cuda.synchronize()
        times = []
        for _ in range(6):  # blocks of five launches: the minimum is the figure to compare (clocks wander)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(5):
                eng.locate_search(q, rec)
            ev[1].record()
            torch.cuda.synchronize()
            times.append(ev[0].elapsed_time(ev[1]) / 5)
        res[f"kind{kind}_ms"] = min(times)
        res[f"kind{kind}_ms_max"] = max(times)
        counts = (rec[:n, 1] - rec[:n, 0]).clone()
        if ref is None:
            ref = (counts, rec[:n].clone())
        else:
            res[f"kind{kind}_counts_equal"] = bool(torch.equal(counts, ref[0]))
        del rec
    # exact intervals (cursors_for_many_queries): general kernel against the exact instantiation of the fast path
    ref_iv = None
    for kind in (0, 1):
        index.set_query_options(search_fast=kind)
        o = eng.alloc_outputs(n)
        eng.search(q, o)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            eng.search(q, o)
        ev[1].record()
        torch.cuda.synchronize()
        res[f"intervals_kind{kind}_ms"] = ev[0].elapsed_time(ev[1]) / 5
        if ref_iv is None:
            ref_iv = (o["start"].clone(), o["end"].clone(), o["status"].clone())
        else:
            res["intervals_equal"] = bool(torch.equal(o["start"], ref_iv[0]) and torch.equal(o["end"], ref_iv[1])
                                          and torch.equal(o["status"], ref_iv[2]))
        del o
    out[name] = res
    del q
print(json.dumps(out))
