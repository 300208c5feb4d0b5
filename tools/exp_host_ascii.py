#!/usr/bin/env python3
"""The reference's own calls on host pointers -- gdx_count_many / gdx_locate_many_alloc on IO symbols + u64 offsets -- on the
hg38-scale default index, with the pipeline packing the chunks on the host (default) and without (GDX_HOST_PACK=0), and the
thread breakdown of every call (GDX_HOST_TIMING=1).  usage: python tools/exp_host_ascii.py [nq]"""
import ctypes as C
import os
import sys
import time

os.environ["GDX_HOST_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genedex_amd import _lib, alphabet  # noqa: E402
from genedex_amd.device import DeviceQueries, build_index_from_device_text, hg38_text_lengths, synth_text  # noqa: E402

total, nq = 3_100_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32")
q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=43)
qbuf = q.qbuf[:q.total_bytes].cpu().numpy()
qoff = q.qoff.cpu().numpy().astype(np.uint64)
lib = _lib.load()
status = np.empty(nq, dtype=np.uint8)
counts = np.empty(nq, dtype=np.uint64)
offs = np.empty(nq + 1, dtype=np.uint64)
qb_p, qo_p = qbuf.ctypes.data_as(_lib.u8p), qoff.ctypes.data_as(_lib.u64p)
ref = {}
for pack in (("1", "0") if os.environ.get("GDX_EXP_BOTH", "1") == "1" else ("1",)):
    os.environ["GDX_HOST_PACK"] = pack
    for rep in range(3):
        t0 = time.perf_counter()
        _lib.check(lib.gdx_count_many(index._h, qb_p, qo_p, nq, counts.ctypes.data_as(_lib.u64p), status.ctypes.data_as(_lib.u8p)))
        dt = time.perf_counter() - t0
        print(f"GDX_HOST_PACK={pack} count, ASCII + offsets, rep {rep}: {dt:.4f}s = {nq / dt / 1e9:.2f} G reads/s", flush=True)
    ref.setdefault("counts", counts.copy())
    assert np.array_equal(ref["counts"], counts)
    for rep in range(3):
        hp, tot = C.POINTER(_lib.HitStruct)(), C.c_uint64(0)
        t0 = time.perf_counter()
        _lib.check(lib.gdx_locate_many_alloc(index._h, qb_p, qo_p, nq, offs.ctypes.data_as(_lib.u64p), C.byref(hp), C.byref(tot),
                                             status.ctypes.data_as(_lib.u8p)))
        dt = time.perf_counter() - t0
        print(f"GDX_HOST_PACK={pack} locate (wide), ASCII + offsets, rep {rep}: {dt:.4f}s = {nq / dt / 1e9:.2f} G reads/s, {tot.value} hits", flush=True)
        if rep == 2:
            h = np.ctypeslib.as_array(C.cast(hp, _lib.u64p), shape=(2 * tot.value,))
            ref.setdefault("offs", offs.copy())
            ref.setdefault("hits_sum", int(h.sum()))
            assert np.array_equal(ref["offs"], offs) and ref["hits_sum"] == int(h.sum())
        lib.gdx_free_hits(hp)
print("identical with and without host packing")
