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
wl = dict(bench.WORKLOADS["hg38"])
torch.cuda.set_device(0)
res = bench.genome_like_secondary(torch, alphabet.ascii_dna_with_n(), wl, args)
print(json.dumps({k: res[k] for k in ("value", "ms_per_step", "search_ms", "scan_and_locate_ms", "hits_located")}))
