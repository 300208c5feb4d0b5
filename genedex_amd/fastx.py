"""FASTA / FASTQ ingestion into the query layout of the ABI (qbuf + qoff), through libgdx.so's streaming reader
(include/gdx.h gdx_fastx_*).  Host only: works without a GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def read_batches(path: str, max_records: int = 1 << 20, buffer_bytes: int = 1 << 28):
    """Yields (qbuf u8[total], qoff u64[n + 1]) for successive batches of at most max_records sequences whose
    symbols fit into buffer_bytes; the arrays are views of reused buffers, copy them to keep them."""
    lib = _lib.load()
    handle = C.c_void_p()
    _lib.check(lib.gdx_fastx_open(path.encode(), C.byref(handle)))
    qbuf = np.empty(buffer_bytes, dtype=np.uint8)
    qoff = np.empty(max_records + 1, dtype=np.uint64)
    n = C.c_uint64(0)
    try:
        while True:
            _lib.check(lib.gdx_fastx_next_batch(handle, qbuf.ctypes.data_as(C.c_void_p), buffer_bytes,
                                                qoff.ctypes.data_as(C.c_void_p), max_records, C.byref(n)))
            if n.value == 0:
                return
            yield qbuf[: int(qoff[n.value])], qoff[: n.value + 1]
    finally:
        lib.gdx_fastx_close(handle)


def read_packed_batches(path: str, alphabet, max_records: int = 1 << 20, buffer_bytes: int = 1 << 28):
    """Batches of a FASTA / FASTQ file in the form the fastest calls take (gdx_query_layout_t): yields dicts with
    `packed` (u8: 2-bit codes, four symbols per byte), `nq`, `uniform_len` (the reads' common length, or 0 when they differ:
    then `qoff` (u64[nq + 1], counting symbols) goes with the batch), `exceptions` (indices of the reads with a symbol
    outside the alphabet's four searchable ones -- N, IUPAC codes: their packed symbols are meaningless) and `qbuf` / `qoff`
    (the ASCII batch itself, for running the exceptions through the plain calls).  Host only: gdx_fastx_next_batch ->
    gdx_pack_queries_table, no index and no device needed.  The arrays of a batch are views of reused buffers."""
    lib = _lib.load()
    table = np.ascontiguousarray(alphabet.io_to_dense_table, dtype=np.uint8)
    packed = np.zeros(int(lib.gdx_packed_bytes(buffer_bytes)), dtype=np.uint8)
    exc = np.empty(max_records, dtype=np.uint64)
    n_exc = C.c_uint64(0)
    for qbuf, qoff in read_batches(path, max_records, buffer_bytes):
        nq = qoff.size - 1
        _lib.check(lib.gdx_pack_queries_table(table.ctypes.data_as(_lib.u8p), qbuf.ctypes.data_as(_lib.u8p) if qbuf.size else None,
                                              qoff.ctypes.data_as(_lib.u64p), nq, packed.ctypes.data_as(_lib.u8p),
                                              exc.ctypes.data_as(_lib.u64p), max_records, C.byref(n_exc)))
        lens = np.diff(qoff)
        uniform = int(lens[0]) if nq and bool((lens == lens[0]).all()) and int(lens[0]) > 0 else 0
        yield {"packed": packed[: int(lib.gdx_packed_bytes(int(qoff[nq])))], "nq": nq, "uniform_len": uniform,
               "qoff": qoff, "qbuf": qbuf, "exceptions": exc[: n_exc.value].copy()}


def read_sequences(path: str):
    """All sequences of a (small) file as a list of bytes objects, e.g. the texts of an index."""
    out = []
    for qbuf, qoff in read_batches(path, max_records=1 << 16, buffer_bytes=1 << 26):
        raw = qbuf.tobytes()
        out.extend(raw[int(qoff[i]):int(qoff[i + 1])] for i in range(qoff.size - 1))
    return out
