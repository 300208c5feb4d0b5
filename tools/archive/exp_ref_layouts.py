#!/usr/bin/env python3
"""The reference's own occurrence tables as operating points (gdx_build_options_t.reference_table_layout): count + locate of
the headline workload on an index whose table is Condensed64 / Condensed512 / Flat64 / Flat512 exactly as genedex builds it,
queried as it is (one lane per query), next to this library's rank lines without any acceleration structure.
usage: python tools/exp_ref_layouts.py [nq] -> one JSON line per layout"""
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

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
total = 3_100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)


class A:
    path, no_hint = "records", False


base = None
for name, opts in [("rank_lines_only", dict(pair_lines=False, jump_entry_bytes=0, top_table_depth=0)),
                   ("condensed64", dict(reference_table_layout="condensed64")),
                   ("condensed512", dict(reference_table_layout="condensed512")),
                   ("flat64", dict(reference_table_layout="flat64")),
                   ("flat512", dict(reference_table_layout="flat512"))]:
    t0 = time.time()
    index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32",
                                         options=build_options(**opts))
    t_build = time.time() - t0
    eng = DeviceEngine(index)
    ms, s_ms, l_ms, counts = bench.time_config(torch, eng, q, nq, True, A, steps=2)
    if base is None:
        base = counts.clone()
    print(json.dumps({"layout": name, "index_gb": index.info.device_bytes / 1e9, "build_s": t_build, "Mq_per_s": nq / ms / 1e3,
                      "ms_per_step": ms, "search_ms": s_ms, "locate_ms": l_ms, "queries": nq,
                      "counts_identical": bool(torch.equal(counts, base))}), flush=True)
    del eng, index, counts
    torch.cuda.empty_cache()
