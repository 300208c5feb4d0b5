#!/usr/bin/env python3
"""FASTQ file -> hits on the hg38-scale default index (bench.fastq_to_hits alone).  usage: python tools/exp_fastq.py [reads] [batch]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import DeviceQueries, build_index_from_device_text, hg38_text_lengths, synth_text  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
total = int(os.environ.get("GDX_EXP_TOTAL", 3_100_000_000))
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32")
q = DeviceQueries.synth(io_text, lengths, n, 50, 50, 900_000, seed=43)
qbuf = q.qbuf[:q.total_bytes].cpu().numpy()
qoff = q.qoff.cpu().numpy().astype(np.uint64)
off, t, p, _ = index.locate_layout32_raw(qbuf, qoff, n)
for rep in range(2):
    print(json.dumps(bench.fastq_to_hits(np, index, qbuf, qoff, n, off.astype(np.uint64), n_reads=n, batch_reads=batch)), flush=True)
