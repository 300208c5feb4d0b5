#!/usr/bin/env python3
"""The host-pointer calls alone on the headline workload (count, locate with a library-allocated hit buffer), for
A/B runs; GDX_HOST_TIMING=1 prints where the pipeline's threads spend the call.  usage: python tools/exp_e2e.py [nq]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genedex_amd import _lib, alphabet  # noqa: E402
from genedex_amd.device import DeviceQueries, build_index_from_device_text, hg38_text_lengths, synth_text  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
total = 3_100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32")
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)
qbuf = q.qbuf.cpu().numpy()
qoff = q.qoff.cpu().numpy().astype(np.uint64)
lib = _lib.load()
u8p, u64p = _lib.u8p, _lib.u64p
counts = np.empty(nq, dtype=np.uint64)
status = np.empty(nq, dtype=np.uint8)
offs = np.empty(nq + 1, dtype=np.uint64)
out = {"count_s": [], "locate_s": []}
for _ in range(3):
    t0 = time.perf_counter()
    _lib.check(lib.gdx_count_many(index._h, qbuf.ctypes.data_as(u8p), qoff.ctypes.data_as(u64p), nq,
                                  counts.ctypes.data_as(u64p), status.ctypes.data_as(u8p)))
    out["count_s"].append(round(time.perf_counter() - t0, 4))
for _ in range(3):
    ptr = C.POINTER(_lib.HitStruct)()
    tot = C.c_uint64(0)
    t0 = time.perf_counter()
    _lib.check(lib.gdx_locate_many_alloc(index._h, qbuf.ctypes.data_as(u8p), qoff.ctypes.data_as(u64p), nq,
                                         offs.ctypes.data_as(u64p), C.byref(ptr), C.byref(tot), status.ctypes.data_as(u8p)))
    out["locate_s"].append(round(time.perf_counter() - t0, 4))
    lib.gdx_free_hits(ptr)
out["hits"] = tot.value
print(json.dumps(out))
