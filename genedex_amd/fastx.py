"""FASTA / FASTQ ingestion into the query layout of the ABI (qbuf + qoff), through libgdx.so's streaming reader
(include/gdx.h gdx_fastx_*).  Host only: works without a GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def make_batch_buffers(max_records: int, buffer_bytes: int, n_buffers: int = 1):
    """buffer sets for read_batches, touched once (np.zeros): a reader that runs for long does not pay the first touch of
    hundreds of megabytes inside every batch"""
    return [(np.zeros(buffer_bytes, dtype=np.uint8), np.zeros(max_records + 1, dtype=np.uint64)) for _ in range(max(1, n_buffers))]


def read_batches(path: str, max_records: int = 1 << 20, buffer_bytes: int = 1 << 28, with_uniform_len: bool = False,
                 n_buffers: int = 1, buffers=None):
    """Yields (qbuf u8[total], qoff u64[n + 1]) for successive batches of at most max_records sequences whose
    symbols fit into buffer_bytes; the arrays are views of reused buffers, copy them to keep them (with n_buffers = k -- or
    `buffers`, sets made by make_batch_buffers -- a batch stays valid until k - 1 more have been taken).  with_uniform_len: a
    third item, the common length of the batch's sequences (0 when they differ), as the reader saw it
    (gdx_fastx_next_batch_ex).  A regular file is parsed by several threads (GDX_FASTX_THREADS; 0 = the streaming reader)."""
    lib = _lib.load()
    handle = C.c_void_p()
    _lib.check(lib.gdx_fastx_open(path.encode(), C.byref(handle)))
    sets = buffers if buffers is not None else [(np.empty(buffer_bytes, dtype=np.uint8), np.empty(max_records + 1, dtype=np.uint64))
                                                for _ in range(max(1, n_buffers))]
    n, ulen = C.c_uint64(0), C.c_uint64(0)
    k = 0
    try:
        while True:
            qbuf, qoff = sets[k % len(sets)]
            k += 1
            _lib.check(lib.gdx_fastx_next_batch_ex(handle, qbuf.ctypes.data_as(C.c_void_p), buffer_bytes,
                                                   qoff.ctypes.data_as(C.c_void_p), max_records, C.byref(n), C.byref(ulen)))
            if n.value == 0:
                return
            if with_uniform_len:
                yield qbuf[: int(qoff[n.value])], qoff[: n.value + 1], int(ulen.value)
            else:
                yield qbuf[: int(qoff[n.value])], qoff[: n.value + 1]
    finally:
        lib.gdx_fastx_close(handle)


def read_packed_batches(path: str, alphabet, max_records: int = 1 << 20, buffer_bytes: int = 1 << 28, n_buffers: int = 1):
    """Batches of a FASTA / FASTQ file in the form the fastest calls take (gdx_query_layout_t): yields dicts with
    `packed` (u8: 2-bit codes, four symbols per byte), `nq`, `uniform_len` (the reads' common length, or 0 when they differ:
    then `qoff` (u64[nq + 1], counting symbols) goes with the batch), `exceptions` (indices of the reads with a symbol
    outside the alphabet's four searchable ones -- N, IUPAC codes: their packed symbols are meaningless) and `qbuf` / `qoff`
    (the ASCII batch itself, for running the exceptions through the plain calls).  Host only: gdx_fastx_next_batch_ex ->
    gdx_pack_queries_table, no index and no device needed.  The arrays of a batch are views of reused buffers: with
    n_buffers = k a batch stays valid until k - 1 more have been taken (a producer thread one batch ahead of its consumer
    needs 3: one being filled, one in the queue, one in use)."""
    lib = _lib.load()
    table = np.ascontiguousarray(alphabet.io_to_dense_table, dtype=np.uint8)
    sets = [dict(qbuf=np.empty(buffer_bytes, dtype=np.uint8), qoff=np.empty(max_records + 1, dtype=np.uint64),
                 packed=np.zeros(int(lib.gdx_packed_bytes(buffer_bytes)), dtype=np.uint8), exc=np.empty(max_records, dtype=np.uint64))
            for _ in range(max(1, n_buffers))]
    handle = C.c_void_p()
    _lib.check(lib.gdx_fastx_open(path.encode(), C.byref(handle)))
    n, ulen, n_exc = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    k = 0
    try:
        while True:
            b = sets[k % len(sets)]
            k += 1
            _lib.check(lib.gdx_fastx_next_batch_ex(handle, b["qbuf"].ctypes.data_as(C.c_void_p), buffer_bytes,
                                                   b["qoff"].ctypes.data_as(C.c_void_p), max_records, C.byref(n), C.byref(ulen)))
            nq = int(n.value)
            if nq == 0:
                return
            qoff = b["qoff"][: nq + 1]
            total = int(qoff[nq])
            _lib.check(lib.gdx_pack_queries_table(table.ctypes.data_as(_lib.u8p), b["qbuf"].ctypes.data_as(_lib.u8p) if total else None,
                                                  qoff.ctypes.data_as(_lib.u64p), nq, b["packed"].ctypes.data_as(_lib.u8p),
                                                  b["exc"].ctypes.data_as(_lib.u64p), max_records, C.byref(n_exc)))
            yield {"packed": b["packed"][: int(lib.gdx_packed_bytes(total))], "nq": nq, "uniform_len": int(ulen.value),
                   "qoff": qoff, "qbuf": b["qbuf"][:total], "exceptions": b["exc"][: min(int(n_exc.value), max_records)].copy()}
    finally:
        lib.gdx_fastx_close(handle)


def read_sequences(path: str, buffer_bytes: int | None = None):
    """All sequences of a file as a list of bytes objects, e.g. the texts of an index from a genome's FASTA file.  The buffer
    is as large as a regular file (at most 4 GB at a time: the symbols cannot be more than the file's bytes, so a genome is one
    batch), 64 MB otherwise; a record larger than the buffer is GDX_ERR_CAPACITY from the reader, which stays at that record:
    the buffer is doubled and the call repeated."""
    import os

    if buffer_bytes is None:
        buffer_bytes = min(max(os.path.getsize(path), 1 << 16), 1 << 32) if os.path.isfile(path) else 1 << 26
    lib = _lib.load()
    handle = C.c_void_p()
    _lib.check(lib.gdx_fastx_open(path.encode(), C.byref(handle)))
    max_records = 1 << 16
    qbuf, qoff = np.empty(buffer_bytes, dtype=np.uint8), np.empty(max_records + 1, dtype=np.uint64)
    n = C.c_uint64(0)
    out = []
    try:
        while True:
            st = lib.gdx_fastx_next_batch_ex(handle, qbuf.ctypes.data_as(C.c_void_p), qbuf.size, qoff.ctypes.data_as(C.c_void_p),
                                             max_records, C.byref(n), None)
            if st == _lib.GDX_ERR_CAPACITY:
                qbuf = np.empty(qbuf.size * 2, dtype=np.uint8)
                continue
            _lib.check(st)
            if n.value == 0:
                return out
            raw = qbuf[: int(qoff[n.value])].tobytes()
            out.extend(raw[int(qoff[i]):int(qoff[i + 1])] for i in range(n.value))
    finally:
        lib.gdx_fastx_close(handle)
