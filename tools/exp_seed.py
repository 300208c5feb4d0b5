#!/usr/bin/env python3
"""The seed-table search on the hg38-scale index (reference arrays + seed table + text units + full suffix array): times of
the search launch alone and of the whole count + locate step for 100 M len-50 reads, then mixed lengths.  Run it under
rocprofv3 --pmc (tools/pmc_seed.sh) for the request and instruction counters of exactly these launches.
usage: python tools/exp_seed.py [reps] [build option=value ...]  -> one JSON line"""
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

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
opts = dict(pair_lines=False, jump_entry_bytes=0, top_table_depth=0, full_suffix_array=True, seed_symbols=True)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    opts[k] = None if v == "None" else int(v)
total = int(os.environ.get("GDX_EXP_TOTAL", 3_100_000_000))
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32",
                                     options=build_options(**opts))
eng = DeviceEngine(index)
res = {"reps": reps, "index_gb": index.info.device_bytes / 1e9, "aux": eng.aux_info()}


class A:
    path, no_hint = "records", False


nq = int(os.environ.get("GDX_EXP_NQ", 100_000_000))
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)
form = os.environ.get("GDX_EXP_INPUT", "ascii")  # the batch as bench.py --input names it (gdx_query_layout_t)
if "packed" in form:
    q = q.as_packed(index)
if "uniform" in form:
    q = q.as_uniform(50)
res["input"] = form
ms, s_ms, l_ms, counts = bench.time_config(torch, eng, q, nq, True, A, steps=reps)
res["len50"] = {"ms_per_step": ms, "search_ms": s_ms, "locate_ms": l_ms, "Gq_per_s": nq / ms / 1e6,
                "found": int((counts > 0).sum().item())}
del q, counts
if os.environ.get("GDX_EXP_SKIP_MIXED") == "1":  # (PMC passes: one kind of launch only)
    print(json.dumps(res))
    sys.exit(0)
nq2 = nq // 2
q = DeviceQueries.synth(io_text, lengths, nq2, 20, 150, 700_000, seed=47)
ms, s_ms, l_ms, counts = bench.time_config(torch, eng, q, nq2, True, A, steps=reps)
res["len20_150"] = {"ms_per_step": ms, "search_ms": s_ms, "locate_ms": l_ms, "Gq_per_s": nq2 / ms / 1e6,
                    "found": int((counts > 0).sum().item())}
print(json.dumps(res))
