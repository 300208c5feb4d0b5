#!/usr/bin/env python3
"""Where the three threads of the host-pointer pipeline spend a call (GDX_HOST_TIMING=1): the locate and count calls on
100 M len-50 reads handed over as 2-bit codes without offsets, and as ASCII with offsets.  usage: python tools/host_timing.py"""
import ctypes as C
import os
import sys
import time

os.environ["GDX_HOST_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from genedex_amd import _lib, alphabet  # noqa: E402
from genedex_amd.device import DeviceQueries, build_index_from_device_text, hg38_text_lengths, synth_text  # noqa: E402
from genedex_amd.index import build_options  # noqa: E402

total, nq = 3_100_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32",
                                     options=build_options(**bench.SEED_INDEX))
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)
qbuf = q.qbuf[:q.total_bytes].cpu().numpy()
qoff = q.qoff.cpu().numpy().astype(np.uint64)
lib = _lib.load()
packed = np.zeros(int(lib.gdx_packed_bytes(int(qoff[-1]))), dtype=np.uint8)
n_exc = C.c_uint64(0)
_lib.check(lib.gdx_pack_queries(index._h, qbuf.ctypes.data_as(_lib.u8p), qoff.ctypes.data_as(_lib.u64p), nq,
                                packed.ctypes.data_as(_lib.u8p), None, 0, C.byref(n_exc)))
for name, buf, off, pk, ul in (("packed+uniform", packed, None, True, 50), ("ascii", qbuf, qoff, False, 0)):
    for rep in range(3):
        t0 = time.perf_counter()
        index.count_layout_raw(buf, off, nq, packed=pk, uniform_len=ul)
        t1 = time.perf_counter()
        o, t, p, _ = index.locate_layout_raw(buf, off, nq, packed=pk, uniform_len=ul)
        t2 = time.perf_counter()
        print(f"{name} rep {rep}: count {t1 - t0:.4f}s locate {t2 - t1:.4f}s (incl. the copy of {int(o[-1])} hits into numpy)", flush=True)
