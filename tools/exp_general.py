#!/usr/bin/env python3
"""The general search kernel on the hg38-scale index: exact intervals of 100 M len-50 reads (cursors_for_many_queries)
and workload 5 through gdx_cursor_extend_front_chunk_dev (5 calls of 32 symbols).  Times per launch; run it under
rocprofv3 --pmc (tools/pmc_general.sh) for the request and instruction counters of exactly these launches.
usage: python tools/exp_general.py [reps]  -> one JSON line"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import (DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths,  # noqa: E402
                                synth_text)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
opts = {}  # build options, e.g. seed_symbols=1 inverse_suffix_array=1 aux_budget_bytes=250000000000
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    opts[k] = int(v)
total = 3_100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
from genedex_amd.index import build_options  # noqa: E402

index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32",
                                     options=build_options(**opts))
eng = DeviceEngine(index)
res = {"reps": reps, "index_gb": index.info.device_bytes / 1e9, "aux": eng.aux_info()}


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


nq = 100_000_000
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)
out = eng.alloc_outputs(nq)
res["exact_len50_ms"] = timed(lambda: eng.search(q, out))
del q, out

nq = 50_000_000
q = DeviceQueries.synth(io_text, lengths, nq, 20, 150, 700_000, seed=47)
out = eng.alloc_outputs(nq)
res["exact_len20_150_ms"] = timed(lambda: eng.search(q, out))
n = index.total_text_len()
cur_s = torch.empty(nq, dtype=torch.int32, device=dev)
cur_e = torch.empty(nq, dtype=torch.int32, device=dev)
cur_st = torch.empty(nq, dtype=torch.uint8, device=dev)
act = [torch.empty(nq, dtype=torch.int32, device=dev) for _ in range(2)]
n_act = [torch.empty(1, dtype=torch.int32, device=dev) for _ in range(2)]
call_ms = []


chunk = int(os.environ.get("GDX_EXP_CHUNK", 32))


def cursor_pass(record=False):
    cur_s.zero_()
    cur_e.fill_(n if n < (1 << 31) else n - (1 << 32))
    cur_st.zero_()
    a, na = None, None
    for r in range((150 + chunk - 1) // chunk):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        eng.cursor_extend_chunk(cur_s, cur_e, q.qbuf, q.qoff, nq, chunk, r, cur_st, a, na, act[r % 2], n_act[r % 2])
        ev[1].record()
        a, na = act[r % 2], n_act[r % 2]
        if record:
            torch.cuda.synchronize()
            call_ms.append(ev[0].elapsed_time(ev[1]))


res["cursor_chunks_ms"] = timed(cursor_pass)
cursor_pass(record=True)
res["cursor_call_ms"] = call_ms
res["cursor_chunk_symbols"] = chunk
res["cursor_equals_fused"] = bool(torch.equal(cur_s, out["start"]) and torch.equal(cur_e, out["end"]))
print(json.dumps(res))
