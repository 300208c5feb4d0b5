#!/usr/bin/env python3
"""Genome-like text: locate walk statistics and kernel times of the record path (debug tool).
usage: python tools/exp_genome_walks.py [nq]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import (DeviceEngine, DeviceQueries, build_index_from_device_text, genome_like_text,  # noqa: E402
                                hg38_text_lengths)

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
total = 3_100_000_000
dev = torch.device("cuda", 0)
text = genome_like_text(total, dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(text, lengths, alphabet.ascii_dna_with_n(), sa_rate=4, index_storage="u32")
eng = DeviceEngine(index)
q = DeviceQueries.synth(text, lengths, nq, 50, 50, 900_000, seed=43)
rec = eng.alloc_records(nq)
off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
out = {}
for name in ("first", "second"):
    eng.locate_search(q, rec)
    eng.locate_offsets(rec, nq, off, 1000)
    torch.cuda.synchronize()
    tot = int(off[nq].item())
    hits = torch.empty((tot, 2), dtype=torch.int32, device=dev)
    ws = torch.empty(eng.locate_workspace_bytes(tot), dtype=torch.uint8, device=dev)
    t0 = time.time()
    eng.locate_hits(rec, nq, off, tot, hits, ws)
    torch.cuda.synchronize()
    t1 = time.time()
    steps, walked = eng.locate_record_walks(rec, nq, off, tot, hits, ws)
    cnt = (rec[:nq, 1] - rec[:nq, 0]).to(torch.int64) & 0xFFFFFFFF
    masked = ((rec[:nq, 3] >> 23) & 1) == 1
    out[name] = {"hits": tot, "locate_s": t1 - t0, "walk_steps": int(steps), "walked_hits": int(walked),
                 "masked_records": int(masked.sum().item()), "hinted": int((rec[:nq, 2] != -1).sum().item()),
                 "max_back": int((rec[:nq, 3] & 0x1fffff)[cnt > 0].max().item())}
print(json.dumps(out))


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.time()
    fn()
    torch.cuda.synchronize()
    return round((time.time() - t0) * 1e3, 3)


res = {"separately_ms": [], "back_to_back_ms": []}
for _ in range(3):
    res["separately_ms"].append([timed(lambda: eng.locate_search(q, rec)), timed(lambda: eng.locate_offsets(rec, nq, off, 1000)),
                                 timed(lambda: eng.locate_hits(rec, nq, off, tot, hits, ws))])
for _ in range(3):
    res["back_to_back_ms"].append(timed(lambda: (eng.locate_search(q, rec), eng.locate_offsets(rec, nq, off, 1000),
                                                 eng.locate_hits(rec, nq, off, tot, hits, ws))))
print(json.dumps(res))
