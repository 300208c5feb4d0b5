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


def read_sequences(path: str):
    """All sequences of a (small) file as a list of bytes objects, e.g. the texts of an index."""
    out = []
    for qbuf, qoff in read_batches(path, max_records=1 << 16, buffer_bytes=1 << 26):
        raw = qbuf.tobytes()
        out.extend(raw[int(qoff[i]):int(qoff[i + 1])] for i in range(qoff.size - 1))
    return out
