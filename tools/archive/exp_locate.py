#!/usr/bin/env python3
"""Where the locate kernel's requests go: walk statistics of the record path at the headline workload.
usage: python tools/exp_locate.py [nq]  -> one JSON line"""
import json
import os
import sys
import time

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
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)
eng = DeviceEngine(index)
rec = eng.alloc_records(nq)
off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
eng.locate_search(q, rec)
eng.locate_offsets(rec, nq, off)
torch.cuda.synchronize()
tot = int(off[nq].item())
hits = torch.empty((tot, 2), dtype=torch.int32, device=dev)
ws = torch.empty(eng.locate_workspace_bytes(tot), dtype=torch.uint8, device=dev)
steps, walked = eng.locate_record_walks(rec, nq, off, tot, hits, ws)
hinted = int((rec[:nq, 2] != -1).sum().item())
single = int(((rec[:nq, 1] - rec[:nq, 0]) == 1).sum().item())
sampled_hint = int(((rec[:nq, 2] != -1) & ((rec[:nq, 2] & 3) == 0)).sum().item())


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


res = {"nq": nq, "hits": tot, "queries_with_one_hit": single, "records_with_hint": hinted,
       "hints_on_sampled_rows": sampled_hint, "hits_that_walk": walked, "walk_steps": steps,
       "walk_steps_per_walking_hit": steps / max(walked, 1),
       "search_ms": timed(lambda: eng.locate_search(q, rec)),
       "offsets_ms": timed(lambda: eng.locate_offsets(rec, nq, off)),
       "locate_ms": timed(lambda: eng.locate_hits(rec, nq, off, tot, hits, ws))}
print(json.dumps(res))
