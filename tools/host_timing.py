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
lay = _lib.QueryLayout()
lib.gdx_query_layout_init(C.byref(lay))
lay.packed, lay.uniform_len = 1, 50
status = np.empty(nq, dtype=np.uint8)
pinned = torch.from_numpy(packed).pin_memory()


def locate32(ptr, what):
    for rep in range(3):
        res = _lib.Hits32()
        t0 = time.perf_counter()
        _lib.check(lib.gdx_locate_many_alloc_layout32(index._h, ptr, None, nq, C.byref(lay), C.byref(res), status.ctypes.data_as(_lib.u8p)))
        dt = time.perf_counter() - t0
        print(f"locate32 {what} rep {rep}: {dt:.4f}s = {nq / dt / 1e9:.2f} G reads/s, {res.total_hits} hits", flush=True)
        lib.gdx_free_hits32(C.byref(res))


for chunk_q in (0, 4 << 20):  # (the library's default chunking, then larger chunks)
    lib.gdx_debug_set_host_chunking(chunk_q, (128 << 20) if chunk_q else 0)
    print(f"--- chunk queries {chunk_q or 'default'}", flush=True)
    locate32(packed.ctypes.data_as(_lib.u8p), "pageable input")
    locate32(C.cast(C.c_void_p(pinned.data_ptr()), _lib.u8p), "pinned input")
lib.gdx_debug_set_host_chunking(0, 0)
for name, buf, off, pk, ul in (("packed+uniform", packed, None, True, 50),):
    for rep in range(2):
        t0 = time.perf_counter()
        index.count_layout_raw(buf, off, nq, packed=pk, uniform_len=ul)
        t1 = time.perf_counter()
        print(f"{name} rep {rep}: count {t1 - t0:.4f}s", flush=True)
# the wide call (u64 offsets, 16-byte hits: what a binding of the reference's locate_many takes), on the same forms
offs = np.empty(nq + 1, dtype=np.uint64)
for what, ptr in (("pageable input", packed.ctypes.data_as(_lib.u8p)), ("pinned input", C.cast(C.c_void_p(pinned.data_ptr()), _lib.u8p))):
    for rep in range(3):
        hp, total = C.POINTER(_lib.HitStruct)(), C.c_uint64(0)
        t0 = time.perf_counter()
        _lib.check(lib.gdx_locate_many_alloc_layout(index._h, ptr, None, nq, C.byref(lay), offs.ctypes.data_as(_lib.u64p), C.byref(hp),
                                                    C.byref(total), status.ctypes.data_as(_lib.u8p)))
        dt = time.perf_counter() - t0
        print(f"locate (wide) {what} rep {rep}: {dt:.4f}s = {nq / dt / 1e9:.2f} G reads/s, {total.value} hits", flush=True)
        lib.gdx_free_hits(hp)
# the reference's own form: IO symbols + u64 offsets (58 bytes per read on the way in: the link is the bound at 0.98 G reads/s)
counts = np.empty(nq, dtype=np.uint64)
qb_p, qo_p = qbuf.ctypes.data_as(_lib.u8p), qoff.ctypes.data_as(_lib.u64p)
for rep in range(3):
    t0 = time.perf_counter()
    _lib.check(lib.gdx_count_many(index._h, qb_p, qo_p, nq, counts.ctypes.data_as(_lib.u64p), status.ctypes.data_as(_lib.u8p)))
    dt = time.perf_counter() - t0
    print(f"count, ASCII + offsets, rep {rep}: {dt:.4f}s = {nq / dt / 1e9:.2f} G reads/s", flush=True)
for rep in range(3):
    hp, total = C.POINTER(_lib.HitStruct)(), C.c_uint64(0)
    t0 = time.perf_counter()
    _lib.check(lib.gdx_locate_many_alloc(index._h, qb_p, qo_p, nq, offs.ctypes.data_as(_lib.u64p), C.byref(hp), C.byref(total),
                                         status.ctypes.data_as(_lib.u8p)))
    dt = time.perf_counter() - t0
    print(f"locate (wide), ASCII + offsets, rep {rep}: {dt:.4f}s = {nq / dt / 1e9:.2f} G reads/s, {total.value} hits", flush=True)
    lib.gdx_free_hits(hp)
for rep in range(3):
    t0 = time.perf_counter()
    _lib.check(lib.gdx_count_many_layout(index._h, packed.ctypes.data_as(_lib.u8p), None, nq, C.byref(lay), counts.ctypes.data_as(_lib.u64p),
                                         status.ctypes.data_as(_lib.u8p)))
    dt = time.perf_counter() - t0
    print(f"count, 2-bit uniform, rep {rep}: {dt:.4f}s = {nq / dt / 1e9:.2f} G reads/s", flush=True)
