#!/usr/bin/env python3
"""Seed-table builds at several load factors on a 256 M-symbol text: buckets asked for and built (a table that cannot place an
entry within 30 buckets of its home is rebuilt with 25 % more buckets), displacement, and that the counts do not change.
usage: python tools/exp_seed_load.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from genedex_amd import alphabet
from genedex_amd.device import build_index_from_device_text, hg38_text_lengths, synth_text, DeviceEngine, DeviceQueries
from genedex_amd.index import build_options
dev = torch.device("cuda", 0)
total = 1 << 28
t = synth_text(total, seed=5, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 5)
base = None
for load in (70, 95, 100):
    ix = build_index_from_device_text(t, lengths, alphabet.ascii_dna_with_n(), index_storage="u32",
                                      options=build_options(pair_lines=False, jump_entry_bytes=0, top_table_depth=0, full_suffix_array=True, seed_symbols=True, seed_load_percent=load))
    info = ix.seed_info()
    entries = info["single_entries"] + info["interval_entries"]
    eng = DeviceEngine(ix)
    q = DeviceQueries.synth(t, lengths, 5_000_000, 50, 50, 900_000, seed=9)
    rec = eng.alloc_records(q.nq); eng.locate_search(q, rec); torch.cuda.synchronize()
    counts = (rec[:q.nq,1]-rec[:q.nq,0]).clone()
    if base is None: base = counts
    print(load, info, "asked buckets", -(-entries*100//(8*load)), "same", bool(torch.equal(counts, base)), flush=True)
    del eng, ix
