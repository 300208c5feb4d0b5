"""ctypes front-end of the CPU oracle (oracle/gdx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under genedex_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def build_oracle(out: str | None = None, cflags: str | None = None) -> str:
    """Compile the oracle with gcc (seconds).  Returns the path of the .so."""
    out = out or os.path.join(_HERE, "libgdx_oracle.so")
    cmd = ["make", "-C", _HERE, f"OUT={out}"]
    if cflags:
        cmd.append(f"CFLAGS={cflags}")
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def load(path: str | None = None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    explicit = path is not None
    # GDX_ORACLE_LIB: alternative build of the oracle (e.g. libgdx_oracle_asan.so under LD_PRELOAD=libasan)
    path = path or os.environ.get("GDX_ORACLE_LIB")
    p = path or os.path.join(_HERE, "libgdx_oracle.so")
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("gdx_oracle.c", "gdx_oracle.h"))
    if path is None and (not os.path.exists(p) or os.path.getmtime(p) < src_m):
        build_oracle(p)
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.gdxo_build.restype = vp
    lib.gdxo_build.argtypes = [u8p, u64p, C.c_uint64, u8p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int]
    lib.gdxo_from_bwt.restype = vp
    lib.gdxo_from_bwt.argtypes = [u8p, C.c_uint64, u32p, C.c_uint64, u64p, u64p, u64p, C.c_uint64, u8p,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.gdxo_table_only.restype = vp
    lib.gdxo_table_only.argtypes = [u8p, C.c_uint64, C.c_int]
    lib.gdxo_free.argtypes = [vp]
    lib.gdxo_n.restype = C.c_uint64
    lib.gdxo_n.argtypes = [vp]
    lib.gdxo_num_texts.restype = C.c_uint64
    lib.gdxo_num_texts.argtypes = [vp]
    lib.gdxo_sigma.restype = C.c_int
    lib.gdxo_sigma.argtypes = [vp]
    for name, rt in [("gdxo_count", u64p), ("gdxo_dense_text", u8p), ("gdxo_bwt", u8p), ("gdxo_full_sa", u32p),
                     ("gdxo_border_keys", u64p), ("gdxo_border_vals", u64p), ("gdxo_sentinel_indices", u64p)]:
        f = getattr(lib, name)
        f.restype = rt
        f.argtypes = [vp]
    for name, rt in [("gdxo_blocks", u64p), ("gdxo_block_offsets", u16p), ("gdxo_superblock_offsets", u32p),
                     ("gdxo_sa_samples", u32p)]:
        f = getattr(lib, name)
        f.restype = rt
        f.argtypes = [vp, u64p]
    lib.gdxo_lookup_table.restype = u32p
    lib.gdxo_lookup_table.argtypes = [vp, C.c_int, u64p]
    lib.gdxo_rank.argtypes = [vp, C.c_int, C.c_uint64, u64p]
    lib.gdxo_symbol_at.argtypes = [vp, C.c_uint64, u8p]
    lib.gdxo_replace_many_interval_borders_with_ranks.argtypes = [vp, u64p, u64p, u8p, C.c_uint64]
    lib.gdxo_cursor_for_query.argtypes = [vp, u8p, C.c_uint64, u64p, u64p]
    lib.gdxo_extend_query_front.argtypes = [vp, u64p, u64p, C.c_uint8]
    lib.gdxo_cursors_for_many_queries.argtypes = [vp, u8p, u64p, C.c_uint64, u64p, u64p, C.c_int]
    lib.gdxo_cursors_single_path.restype = None
    lib.gdxo_cursors_single_path.argtypes = [vp, u8p, u64p, C.c_uint64, u64p, u64p, u8p, C.c_int]
    lib.gdxo_locate_interval.restype = None
    lib.gdxo_locate_interval.argtypes = [vp, C.c_uint64, C.c_uint64, u64p, u64p]
    lib.gdxo_locate_intervals.restype = None
    lib.gdxo_locate_intervals.argtypes = [vp, u64p, u64p, C.c_uint64, u64p, u64p, u64p, C.c_int]
    lib.gdxo_lookup_text_id.restype = C.c_uint64
    lib.gdxo_lookup_text_id.argtypes = [vp, C.c_uint64]
    lib.gdxo_recover_range.restype = None
    lib.gdxo_recover_range.argtypes = [vp, C.c_uint64, C.c_uint64, u64p]
    lib.gdxo_table_construct.restype = vp
    lib.gdxo_table_construct.argtypes = [u8p, C.c_uint64, C.c_int, C.c_int, C.c_int]
    lib.gdxo_table_free.argtypes = [vp]
    lib.gdxo_table_rank.argtypes = [vp, C.c_int, C.c_uint64, u64p]
    lib.gdxo_table_symbol_at.argtypes = [vp, C.c_uint64, u8p]
    lib.gdxo_table_blocks.restype = u64p
    lib.gdxo_table_blocks.argtypes = [vp, u64p]
    lib.gdxo_table_block_offsets.restype = u16p
    lib.gdxo_table_block_offsets.argtypes = [vp, u64p]
    lib.gdxo_table_superblock_offsets.restype = u32p
    lib.gdxo_table_superblock_offsets.argtypes = [vp, u64p]
    lib.gdxo_naive_suffix_array.restype = None
    lib.gdxo_naive_suffix_array.argtypes = [u8p, C.c_uint64, u32p]
    if not explicit:
        _LIB = lib
    return lib


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def pack_queries(queries):
    """list of bytes-like -> (qbuf u8, qoff u64[nq+1])"""
    lens = np.fromiter((len(q) for q in queries), dtype=np.uint64, count=len(queries))
    qoff = np.zeros(len(queries) + 1, dtype=np.uint64)
    np.cumsum(lens, out=qoff[1:])
    qbuf = np.frombuffer(b"".join(bytes(q) for q in queries), dtype=np.uint8).copy()
    if qbuf.size == 0:
        qbuf = np.zeros(1, dtype=np.uint8)
    return qbuf, qoff


class OracleIndex:
    def __init__(self, handle, lib, io_to_dense=None):
        if not handle:
            raise ValueError("oracle: construction failed (invalid arguments / invalid text symbol)")
        self._h = C.c_void_p(handle)
        self._lib = lib
        self.io_to_dense = io_to_dense

    def __del__(self):
        try:
            if self._h:
                self._lib.gdxo_free(self._h)
                self._h = None
        except Exception:
            pass

    # ---- constructors -------------------------------------------------------
    @classmethod
    def build(cls, texts, io_to_dense, sigma, n_searchable, sa_rate=4, lookup_depth=0, width=32, lib=None):
        lib = lib or load()
        tbuf, toff = pack_queries(texts)
        tab = np.ascontiguousarray(io_to_dense, dtype=np.uint8)
        h = lib.gdxo_build(_p(tbuf, u8p), _p(toff, u64p), len(texts), _p(tab, u8p), sigma, n_searchable,
                           sa_rate, lookup_depth, width)
        return cls(h, lib, tab)

    @classmethod
    def from_bwt(cls, bwt, sa_samples, sa_rate, border_keys, border_vals, sentinel_indices, io_to_dense, sigma,
                 n_searchable, lookup_depth=0, width=32, n_threads=1, lib=None):
        lib = lib or load()
        bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
        sa_samples = np.ascontiguousarray(sa_samples, dtype=np.uint32)
        bk = np.ascontiguousarray(border_keys, dtype=np.uint64)
        bv = np.ascontiguousarray(border_vals, dtype=np.uint64)
        si = np.ascontiguousarray(sentinel_indices, dtype=np.uint64)
        tab = np.ascontiguousarray(io_to_dense, dtype=np.uint8)
        h = lib.gdxo_from_bwt(_p(bwt, u8p), bwt.size, _p(sa_samples, u32p), sa_rate, _p(bk, u64p), _p(bv, u64p),
                              _p(si, u64p), si.size, _p(tab, u8p), sigma, n_searchable, lookup_depth, width,
                              n_threads)
        return cls(h, lib, tab)

    @classmethod
    def table_only(cls, dense_text, sigma, lib=None):
        lib = lib or load()
        t = np.ascontiguousarray(dense_text, dtype=np.uint8)
        tt = t if t.size else np.zeros(1, dtype=np.uint8)
        return cls(lib.gdxo_table_only(_p(tt, u8p), t.size, sigma), lib)

    # ---- plain data ---------------------------------------------------------
    @property
    def n(self):
        return int(self._lib.gdxo_n(self._h))

    @property
    def num_texts(self):
        return int(self._lib.gdxo_num_texts(self._h))

    @property
    def sigma(self):
        return int(self._lib.gdxo_sigma(self._h))

    def _arr(self, ptr, n, dtype):
        if not ptr or n == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(int(n),)).copy()

    @property
    def count_array(self):
        return self._arr(self._lib.gdxo_count(self._h), self.sigma + 1, np.uint64)

    @property
    def dense_text(self):
        return self._arr(self._lib.gdxo_dense_text(self._h), self.n, np.uint8)

    @property
    def bwt(self):
        return self._arr(self._lib.gdxo_bwt(self._h), self.n, np.uint8)

    @property
    def full_sa(self):
        return self._arr(self._lib.gdxo_full_sa(self._h), self.n, np.uint32)

    def _lenarr(self, fn, dtype):
        ln = C.c_uint64(0)
        ptr = fn(self._h, C.byref(ln))
        return self._arr(ptr, ln.value, dtype)

    @property
    def blocks(self):
        return self._lenarr(self._lib.gdxo_blocks, np.uint64)

    @property
    def block_offsets(self):
        return self._lenarr(self._lib.gdxo_block_offsets, np.uint16)

    @property
    def superblock_offsets(self):
        return self._lenarr(self._lib.gdxo_superblock_offsets, np.uint32)

    @property
    def sa_samples(self):
        return self._lenarr(self._lib.gdxo_sa_samples, np.uint32)

    @property
    def border_keys(self):
        return self._arr(self._lib.gdxo_border_keys(self._h), self.num_texts, np.uint64)

    @property
    def border_vals(self):
        return self._arr(self._lib.gdxo_border_vals(self._h), self.num_texts, np.uint64)

    @property
    def sentinel_indices(self):
        return self._arr(self._lib.gdxo_sentinel_indices(self._h), self.num_texts, np.uint64)

    def lookup_table(self, depth):
        ln = C.c_uint64(0)
        ptr = self._lib.gdxo_lookup_table(self._h, depth, C.byref(ln))
        return self._arr(ptr, ln.value * 2, np.uint32).reshape(-1, 2)

    # ---- operator level -----------------------------------------------------
    def rank(self, symbol, idx):
        out = C.c_uint64(0)
        if self._lib.gdxo_rank(self._h, int(symbol), int(idx), C.byref(out)) != 0:
            raise AssertionError("rank: assert!(is_safe) failed (text_with_rank_support/mod.rs:107-108)")
        return out.value

    def symbol_at(self, idx):
        out = C.c_uint8(0)
        if self._lib.gdxo_symbol_at(self._h, int(idx), C.byref(out)) != 0:
            raise AssertionError("symbol_at: assert!(idx < text_len) failed (condensed.rs:344)")
        return out.value

    def rank_many_scalar(self, symbols, idxs):
        return np.array([self.rank(int(c), int(i)) for c, i in zip(symbols, idxs)], dtype=np.uint64)

    def replace_many(self, starts, ends, symbols):
        s = np.array(starts, dtype=np.uint64)
        e = np.array(ends, dtype=np.uint64)
        sy = np.ascontiguousarray(symbols, dtype=np.uint8)
        rc = self._lib.gdxo_replace_many_interval_borders_with_ranks(self._h, _p(s, u64p), _p(e, u64p),
                                                                     _p(sy, u8p), s.size)
        if rc != 0:
            raise AssertionError("replace_many: assert failed (mod.rs:41-52 / condensed.rs:146)")
        return s, e

    # ---- search -------------------------------------------------------------
    def cursor_for_query(self, q):
        qa = np.frombuffer(bytes(q), dtype=np.uint8).copy() if len(q) else np.zeros(1, dtype=np.uint8)
        s = C.c_uint64(0)
        e = C.c_uint64(0)
        st = self._lib.gdxo_cursor_for_query(self._h, _p(qa, u8p), len(q), C.byref(s), C.byref(e))
        return s.value, e.value, st

    def count(self, q):
        s, e, st = self.cursor_for_query(q)
        if st:
            raise RuntimeError(f"query status {st}")
        return e - s

    def extend_front(self, start, end, io_symbol):
        s = C.c_uint64(start)
        e = C.c_uint64(end)
        st = self._lib.gdxo_extend_query_front(self._h, C.byref(s), C.byref(e), int(io_symbol))
        return s.value, e.value, st

    def cursors_for_many(self, qbuf, qoff, n_threads=1):
        nq = qoff.size - 1
        s = np.zeros(nq, dtype=np.uint64)
        e = np.zeros(nq, dtype=np.uint64)
        rc = self._lib.gdxo_cursors_for_many_queries(self._h, _p(qbuf, u8p), _p(qoff, u64p), nq, _p(s, u64p),
                                                     _p(e, u64p), n_threads)
        if rc != 0:
            raise RuntimeError(f"batched path panicked with status {rc}")
        return s, e

    def cursors_single(self, qbuf, qoff, n_threads=1):
        nq = qoff.size - 1
        s = np.zeros(nq, dtype=np.uint64)
        e = np.zeros(nq, dtype=np.uint64)
        st = np.zeros(nq, dtype=np.uint8)
        self._lib.gdxo_cursors_single_path(self._h, _p(qbuf, u8p), _p(qoff, u64p), nq, _p(s, u64p), _p(e, u64p),
                                           _p(st, u8p), n_threads)
        return s, e, st

    def count_many(self, queries):
        qbuf, qoff = pack_queries(queries)
        s, e = self.cursors_for_many(qbuf, qoff)
        return e - s

    # ---- locate -------------------------------------------------------------
    def locate_interval(self, start, end):
        m = int(end - start)
        t = np.zeros(max(m, 1), dtype=np.uint64)
        p = np.zeros(max(m, 1), dtype=np.uint64)
        self._lib.gdxo_locate_interval(self._h, int(start), int(end), _p(t, u64p), _p(p, u64p))
        return t[:m], p[:m]

    def locate(self, q):
        s, e, st = self.cursor_for_query(q)
        if st:
            raise RuntimeError(f"query status {st}")
        t, p = self.locate_interval(s, e)
        return list(zip(t.tolist(), p.tolist()))

    def locate_intervals(self, starts, ends, n_threads=1):
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        ends = np.ascontiguousarray(ends, dtype=np.uint64)
        off = np.zeros(starts.size + 1, dtype=np.uint64)
        np.cumsum(ends - starts, out=off[1:])
        tot = int(off[-1])
        t = np.zeros(max(tot, 1), dtype=np.uint64)
        p = np.zeros(max(tot, 1), dtype=np.uint64)
        self._lib.gdxo_locate_intervals(self._h, _p(starts, u64p), _p(ends, u64p), starts.size, _p(off, u64p),
                                        _p(t, u64p), _p(p, u64p), n_threads)
        return off, t[:tot], p[:tot]

    def locate_many(self, queries, n_threads=1):
        qbuf, qoff = pack_queries(queries)
        s, e = self.cursors_for_many(qbuf, qoff)
        return self.locate_intervals(s, e, n_threads)

    def recover_range(self, start, end):
        m = int(end - start)
        out = np.zeros(max(m, 1), dtype=np.uint64)
        self._lib.gdxo_recover_range(self._h, int(start), int(end), _p(out, u64p))
        return out[:m]

    def lookup_text_id(self, pos):
        return int(self._lib.gdxo_lookup_text_id(self._h, int(pos)))


def naive_suffix_array(text: np.ndarray) -> np.ndarray:
    lib = load()
    t = np.ascontiguousarray(text, dtype=np.uint8)
    sa = np.zeros(max(t.size, 1), dtype=np.uint32)
    tt = t if t.size else np.zeros(1, dtype=np.uint8)
    lib.gdxo_naive_suffix_array(_p(tt, u8p), t.size, _p(sa, u32p))
    return sa[: t.size]


class OracleTable:
    """One of the reference's four occurrence-table variants (kind 'condensed' | 'flat', block 64 | 512)."""

    KINDS = {"condensed": 0, "flat": 1}

    def __init__(self, dense_text, sigma, kind="condensed", block_bits=64, lib=None):
        self._lib = lib or load()
        t = np.ascontiguousarray(dense_text, dtype=np.uint8)
        tt = t if t.size else np.zeros(1, dtype=np.uint8)
        h = self._lib.gdxo_table_construct(_p(tt, u8p), t.size, sigma, self.KINDS[kind], block_bits)
        if not h:
            raise ValueError("oracle table: invalid arguments")
        self._h = C.c_void_p(h)
        self.n, self.sigma, self.kind, self.block_bits = t.size, sigma, kind, block_bits

    def __del__(self):
        try:
            if self._h:
                self._lib.gdxo_table_free(self._h)
                self._h = None
        except Exception:
            pass

    def rank(self, symbol, idx):
        out = C.c_uint64(0)
        if self._lib.gdxo_table_rank(self._h, int(symbol), int(idx), C.byref(out)) != 0:
            raise AssertionError("rank: assert!(is_safe) failed")
        return out.value

    def symbol_at(self, idx):
        out = C.c_uint8(0)
        if self._lib.gdxo_table_symbol_at(self._h, int(idx), C.byref(out)) != 0:
            raise AssertionError("symbol_at: assert failed")
        return out.value

    def _lenarr(self, fn, dtype):
        ln = C.c_uint64(0)
        ptr = fn(self._h, C.byref(ln))
        if not ptr or ln.value == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(int(ln.value),)).copy()

    @property
    def blocks(self):
        return self._lenarr(self._lib.gdxo_table_blocks, np.uint64)

    @property
    def block_offsets(self):
        return self._lenarr(self._lib.gdxo_table_block_offsets, np.uint16)

    @property
    def superblock_offsets(self):
        return self._lenarr(self._lib.gdxo_table_superblock_offsets, np.uint32)
