#!/usr/bin/env python3
"""Workload 5 through the batched cursor API (bench.py mixed_length_secondary), timed call by call.
usage: python tools/exp_cursor.py  -> one JSON line"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import (DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths,  # noqa: E402
                                synth_text)

total, nq, chunk = 3_100_000_000, 50_000_000, 32
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32")
eng = DeviceEngine(index)
q = DeviceQueries.synth(io_text, lengths, nq, 20, 150, 700_000, seed=47)
n = index.total_text_len()
beg, end = q.qoff[:-1], q.qoff[1:]
cur_s = torch.empty(nq, dtype=torch.int32, device=dev)
cur_e = torch.empty(nq, dtype=torch.int32, device=dev)
cur_st = torch.empty(nq, dtype=torch.uint8, device=dev)
act = [torch.empty(nq, dtype=torch.int32, device=dev) for _ in range(2)]
n_act = [torch.empty(1, dtype=torch.int32, device=dev) for _ in range(2)]
edges = [torch.empty(nq, dtype=torch.int64, device=dev) for _ in range(2)]
rounds = -(-150 // chunk)
out = {}
for fast in (1, 0):
    index.set_query_options(search_fast=fast)
    for rep in range(2):
        cur_s.zero_()
        cur_e.fill_(n if n < (1 << 31) else n - (1 << 32))
        cur_st.zero_()
        hi = end
        a, na = None, None
        ev = []
        for r in range(rounds):
            lo = edges[r % 2]
            torch.sub(hi, chunk, out=lo)
            torch.maximum(lo, beg, out=lo)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.cursor_extend_strings(cur_s, cur_e, q.qbuf, lo, hi, nq, cur_st, a, na, act[r % 2], n_act[r % 2])
            e1.record()
            ev.append((e0, e1))
            a, na = act[r % 2], n_act[r % 2]
            hi = lo
        torch.cuda.synchronize()
        out[f"fast{fast}_call_ms"] = [round(x.elapsed_time(y), 3) for x, y in ev]
print(json.dumps(out))
