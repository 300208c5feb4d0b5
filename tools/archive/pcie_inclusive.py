#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer ABI (gdx_count_many / gdx_locate_many on host buffers): upload of the
queries, kernels, download of the results.  BASELINE workload 2 scale (256 MB text, 10 M reads).  Never the
bench `value`; recorded in DESIGN.md."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import DeviceQueries, build_index_from_device_text, synth_text  # noqa: E402

total, nq = 1 << 28, 10_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, device=dev)
index = build_index_from_device_text(io_text, [total], alphabet.ascii_dna_with_n(), index_storage="i32")
q = DeviceQueries.synth(io_text, [total], nq, 50, 50, 900_000)
qbuf, qoff = q.host_slice(0, nq)
qbuf, qoff = np.ascontiguousarray(qbuf), np.ascontiguousarray(qoff)
index.count_raw(qbuf[: 50 * 1000], qoff[:1001])
t0 = time.perf_counter()
counts, _ = index.count_raw(qbuf, qoff)
t_count = time.perf_counter() - t0
t0 = time.perf_counter()
off, t, p, _ = index.locate_raw(qbuf, qoff)
t_locate = time.perf_counter() - t0
print(json.dumps({"queries": nq, "count_many_host_s": t_count, "count_many_host_qps": nq / t_count,
                  "locate_many_host_s": t_locate, "locate_many_host_qps": nq / t_locate, "hits": int(off[-1])}))
