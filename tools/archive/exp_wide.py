#!/usr/bin/env python3
"""The 64-bit engine (wide.hip) at n = 2^32 + 2^20: build time and host-pointer query rates.  -> one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import DeviceQueries, build_index_from_device_text, synth_text  # noqa: E402

dev = torch.device("cuda", 0)
total = (1 << 32) + (1 << 20) - 3
io_text = synth_text(total, seed=79, n_per_million=10_000, device=dev)
lengths = [total - (1 << 21), (1 << 21) - 5, 5]
t0 = time.time()
g = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="i64")
t_build = time.time() - t0
nq = 4_000_000
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=80)
qbuf, qoff = q.host_slice(0, nq)
g.count_raw(qbuf, qoff)
t0 = time.time()
counts, _ = g.count_raw(qbuf, qoff)
t_count = time.time() - t0
t0 = time.time()
off, t, p, _ = g.locate_alloc_raw(qbuf, qoff)
t_locate = time.time() - t0
print(json.dumps({"n": g.total_text_len(), "index_width": int(g.info.index_width), "index_gb": g.info.device_bytes / 1e9,
                  "build_s": t_build, "queries": nq, "count_qps": nq / t_count, "count_and_locate_qps": nq / t_locate,
                  "hits": int(off[-1])}))
