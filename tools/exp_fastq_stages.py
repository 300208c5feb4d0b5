#!/usr/bin/env python3
"""Where the FASTQ -> hits pipeline of bench.fastq_to_hits spends a batch: reader, packer and GPU call timed one after the other
on the same batches (no overlap), on the hg38-scale default index.  usage: python tools/exp_fastq_stages.py [reads] [batch]"""
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genedex_amd import _lib, alphabet, fastx  # noqa: E402
from genedex_amd.device import DeviceQueries, build_index_from_device_text, hg38_text_lengths, synth_text  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
total = 3_100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32")
q = DeviceQueries.synth(io_text, lengths, n, 50, 50, 900_000, seed=43)
qbuf = q.qbuf[:q.total_bytes].cpu().numpy()
ln = 50
rec = np.empty((n, ln * 2 + 7), dtype=np.uint8)
rec[:, 0], rec[:, 1], rec[:, 2] = ord("@"), ord("r"), 10
rec[:, 3: 3 + ln] = qbuf[: n * ln].reshape(n, ln)
rec[:, 3 + ln], rec[:, 4 + ln], rec[:, 5 + ln] = 10, ord("+"), 10
rec[:, 6 + ln: 6 + 2 * ln] = ord("I")
rec[:, 6 + 2 * ln] = 10
with tempfile.NamedTemporaryFile(prefix="gdx_stages_", suffix=".fq", dir="/tmp", delete=False) as f:
    path = f.name
rec.tofile(path)
del rec
lib = _lib.load()
alpha = alphabet.ascii_dna_with_n()
table = np.ascontiguousarray(alpha.io_to_dense_table, dtype=np.uint8)
lay = _lib.QueryLayout()
lib.gdx_query_layout_init(C.byref(lay))
status = np.empty(batch, dtype=np.uint8)
packed = np.zeros(int(lib.gdx_packed_bytes(batch * ln)), dtype=np.uint8)
exc = np.empty(batch, dtype=np.uint64)
n_exc = C.c_uint64(0)
try:
    for rep in range(2):
        t_read = t_pack = t_gpu = t_gpu_ascii = 0.0
        t0 = time.perf_counter()
        for qb, qo, ul in fastx.read_batches(path, max_records=batch, buffer_bytes=batch * ln, with_uniform_len=True):
            t1 = time.perf_counter()
            t_read += t1 - t0
            nq = qo.size - 1
            _lib.check(lib.gdx_pack_queries_table(table.ctypes.data_as(_lib.u8p), qb.ctypes.data_as(_lib.u8p), qo.ctypes.data_as(_lib.u64p), nq,
                                                  packed.ctypes.data_as(_lib.u8p), exc.ctypes.data_as(_lib.u64p), batch, C.byref(n_exc)))
            t2 = time.perf_counter()
            t_pack += t2 - t1
            lay.packed, lay.uniform_len = 1, ul
            r32 = _lib.Hits32()
            _lib.check(lib.gdx_locate_many_alloc_layout32(index._h, packed.ctypes.data_as(_lib.u8p), None, nq, C.byref(lay), C.byref(r32),
                                                          status.ctypes.data_as(_lib.u8p)))
            lib.gdx_free_hits32(C.byref(r32))
            t3 = time.perf_counter()
            t_gpu += t3 - t2
            # the same batch as ASCII, uniform: the call's own feeder packs it
            lay.packed, lay.uniform_len = 0, ul
            r32 = _lib.Hits32()
            _lib.check(lib.gdx_locate_many_alloc_layout32(index._h, qb.ctypes.data_as(_lib.u8p), None, nq, C.byref(lay), C.byref(r32),
                                                          status.ctypes.data_as(_lib.u8p)))
            lib.gdx_free_hits32(C.byref(r32))
            t0 = time.perf_counter()
            t_gpu_ascii += t0 - t3
        k = n / batch
        print(f"rep {rep}: per {batch} reads: reader {t_read / k * 1e3:.1f} ms, packer {t_pack / k * 1e3:.1f} ms, GPU call on 2-bit codes {t_gpu / k * 1e3:.1f} ms, "
              f"GPU call on the ASCII batch {t_gpu_ascii / k * 1e3:.1f} ms", flush=True)
finally:
    os.remove(path)
