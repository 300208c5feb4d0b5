#!/usr/bin/env python3
"""The like-for-like locate walk (sampled_suffix_array.rs:110-138): the reference's arrays and nothing else, the headline's reads,
search -> offsets -> locate_queue_kernel<LineTable>.  Times the locate launch and counts its walk steps; run it under
tools/pmc_exp.sh with the kernel regex locate_queue_kernel for the request counters.  usage: python tools/exp_locate_walk.py [nq] [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths, synth_text  # noqa: E402
from genedex_amd.index import build_options  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
total = 3_100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32",
                                     options=build_options(**bench.REFERENCE_ARRAYS))
eng = DeviceEngine(index)
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)
rec = eng.alloc_records(nq)
eng.locate_search(q, rec)
off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
eng.locate_offsets(rec, nq, off)
torch.cuda.synchronize()
total_hits = int(off[nq].item())
hits = torch.empty((total_hits, 2), dtype=torch.int32, device=dev)
ws = torch.empty(max(eng.locate_workspace_bytes(total_hits), 16), dtype=torch.uint8, device=dev)
eng.locate_hits(rec, nq, off, total_hits, hits, ws)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(reps):
    eng.locate_hits(rec, nq, off, total_hits, hits, ws)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
steps, walked = eng.locate_record_walks(rec, nq, off, total_hits, hits, ws)
res = {"queries": nq, "hits": total_hits, "locate_ms": ms, "walk_steps": steps, "hits_that_walked": walked,
       "walk_steps_per_hit": steps / total_hits, "Ghits_per_s": total_hits / ms / 1e6,
       # 30 B per walk step + 4 B sample + 8 B hit
       "algorithmic_bytes": 30 * steps + 12 * total_hits, "frac_algorithmic": (30 * steps + 12 * total_hits) / (ms / 1e3) / 8e12,
       "index_gb": index.info.device_bytes / 1e9}
print(json.dumps(res))
