#!/usr/bin/env python3
"""Speed-vs-HBM ladder on the hg38-scale index: the count + locate step of bench.py on indexes whose acceleration
structures are rebuilt on the same suffix array (gdx_index_rebuild_aux); every rung must reproduce the first rung's
counts.  usage: python tools/exp_ladder.py [rung ...]  -> one JSON line per rung"""
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

text = dict(jump_entry_bytes=0, pair_lines=False, text_units=True)
RUNGS = {
    "default": {},
    "top16_sa_text": dict(top_table_depth=16, full_suffix_array=True, **text),
    "top15_sa_text": dict(top_table_depth=15, full_suffix_array=True, **text),
    "top14_sa_text": dict(top_table_depth=14, full_suffix_array=True, **text),
    "top15_text": dict(top_table_depth=15, **text),
    "top14_text": dict(top_table_depth=14, **text),
    "top13_text": dict(top_table_depth=13, **text),
    "top12_text": dict(top_table_depth=12, **text),
    "top16_sa_text_pairs": dict(top_table_depth=16, full_suffix_array=True, jump_entry_bytes=0, text_units=True),
    "pair_lines_only": dict(top_table_depth=0, jump_entry_bytes=0),
    # seed table in front (one bucket fetch per read; reads whose k-mer occurs once are answered by the entry)
    "seed_sa_text": dict(top_table_depth=0, full_suffix_array=True, seed_symbols=True, **text),
    "seed_text": dict(top_table_depth=0, seed_symbols=True, **text),
    "seed60_sa_text": dict(top_table_depth=0, full_suffix_array=True, seed_symbols=True, seed_load_percent=60, **text),
    "seed80_sa_text": dict(top_table_depth=0, full_suffix_array=True, seed_symbols=True, seed_load_percent=80, **text),
    "seed20_sa_text": dict(top_table_depth=0, full_suffix_array=True, seed_symbols=20, **text),
    "seed_all": dict(seed_symbols=True, aux_budget_bytes=240_000_000_000),
}
names = sys.argv[1:] or list(RUNGS)
total = 3_100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32")
eng = DeviceEngine(index)
nq = 100_000_000
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)


class A:  # the few bench.py arguments time_config reads
    path, no_hint = "records", False


base = None
for name in names:
    t0 = time.time()
    index.rebuild_aux(**RUNGS[name])
    t_aux = time.time() - t0
    ms, s_ms, l_ms, counts = bench.time_config(torch, eng, q, nq, True, A)
    if base is None:
        base = counts.clone()
    print(json.dumps({"name": name, "index_gb": index.info.device_bytes / 1e9, "Gq_per_s": nq / ms / 1e6, "ms_per_step": ms,
                      "search_ms": s_ms, "locate_ms": l_ms, "counts_identical": bool(torch.equal(counts, base)),
                      "aux_rebuild_s": t_aux, "aux": eng.aux_info()}), flush=True)
    del counts
