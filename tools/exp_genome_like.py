#!/usr/bin/env python3
"""The genome-like secondary of bench.py alone (text, index, 100 M reads, search + capped locate), for A/B runs over
environment knobs.  usage: python tools/exp_genome_like.py  -> one JSON line"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from genedex_amd import alphabet  # noqa: E402

args = argparse.Namespace(sa_rate=4, lookup_depth=0, jump_bytes=None, top_depth=None, no_pair_lines=False, lanes=None, load_policy=None)
for kv in sys.argv[1:]:  # e.g. seed_symbols=1 aux_budget_bytes=240000000000 jump_bytes=0 top_depth=0 no_pair_lines=1 full_sa=1 genome_path=records
    k, v = kv.split("=")
    setattr(args, k, int(v) if v.lstrip("-").isdigit() else v)
wl = dict(bench.WORKLOADS["hg38"])
torch.cuda.set_device(0)
res = bench.genome_like_secondary(torch, alphabet.ascii_dna_with_n(), wl, args)
out = {k: res[k] for k in ("value", "ms_per_step", "search_ms", "scan_and_locate_ms", "hits_located", "index_build_seconds",
                         "located_queries_by_hits")}
out["aux"] = res["aux_structures"]
print(json.dumps(out))
