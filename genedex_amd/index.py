"""Host-side mirror of the reference's public API for the query path.

Names, argument meaning and error behaviour follow feldroop/genedex v0.2.2:
``FmIndexConfig`` (src/config.rs:17-70), ``FmIndex`` (src/lib.rs:117-327), ``Cursor``
(src/cursor.rs:16-73), ``Hit`` (src/lib.rs:331-335).  Every query runs in the HIP kernels of
libgdx.so; a panic of the reference becomes a Python exception.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import numpy as np

from . import _lib
from ._lib import GdxError, u8p, u16p, u32p, u64p
from .alphabet import Alphabet


class Hit(NamedTuple):
    """src/lib.rs:331-335"""
    text_id: int
    position: int


def _p(a, t):
    return a.ctypes.data_as(t)


def pack_queries(queries):
    """list of bytes-like -> (qbuf u8[...], qoff u64[nq+1])"""
    queries = [bytes(q) for q in queries]
    lens = np.fromiter((len(q) for q in queries), dtype=np.uint64, count=len(queries))
    qoff = np.zeros(len(queries) + 1, dtype=np.uint64)
    np.cumsum(lens, out=qoff[1:])
    joined = b"".join(queries)
    qbuf = np.frombuffer(joined, dtype=np.uint8).copy() if joined else np.zeros(1, dtype=np.uint8)
    return qbuf, qoff


_WIDTHS = {"u32": 32, "i32": -32, "i64": 64}


def build_options(pair_lines=None, jump_entry_bytes=None, top_table_depth=None, aux_budget_bytes=None,
                  full_suffix_array=None, text_units=None, seed_symbols=None, seed_load_percent=None,
                  inverse_suffix_array=None, reference_table_layout=None):
    """gdx_build_options_t (include/gdx.h); None = the library's default for that field."""
    o = _lib.BuildOptions()
    _lib.load().gdx_build_options_init(C.byref(o))
    if pair_lines is not None:
        o.pair_lines = int(bool(pair_lines))
    if jump_entry_bytes is not None:
        o.jump_entry_bytes = int(jump_entry_bytes)
    if top_table_depth is not None:
        o.top_table_depth = int(top_table_depth)
    if aux_budget_bytes is not None:
        o.aux_budget_bytes = int(aux_budget_bytes)
    if full_suffix_array is not None:
        o.full_suffix_array = int(bool(full_suffix_array))
    if text_units is not None:
        o.text_units = int(bool(text_units))
    if seed_symbols is not None:
        o.seed_symbols = 1 if seed_symbols is True else int(seed_symbols)  # True = k chosen from the text length
    if seed_load_percent is not None:
        o.seed_load_percent = int(seed_load_percent)
    if inverse_suffix_array is not None:
        o.inverse_suffix_array = int(bool(inverse_suffix_array))
    if reference_table_layout is not None:
        names = {"condensed64": 1, "condensed512": 2, "flat64": 3, "flat512": 4}
        o.reference_table_layout = names.get(reference_table_layout, reference_table_layout)
    return o


class FmIndexConfig:
    """Builder for the index (src/config.rs:17-82).  `index_storage` is the reference's generic `I`."""

    def __init__(self, index_storage: str = "i32"):
        if index_storage not in _WIDTHS:
            raise ValueError("index_storage must be one of 'i32', 'u32', 'i64'")
        self.index_storage = index_storage
        self._sa_rate = 4    # config.rs:75
        self._depth = 0      # config.rs:76
        self._device = 0
        self._build = {}     # gdx_build_options_t fields (this implementation's own knobs)

    def suffix_array_sampling_rate(self, rate: int) -> "FmIndexConfig":
        assert rate > 0  # config.rs:28
        self._sa_rate = int(rate)
        return self

    def lookup_table_depth(self, depth: int) -> "FmIndexConfig":
        self._depth = int(depth)
        return self

    def construction_performance_priority(self, _priority) -> "FmIndexConfig":
        # config.rs:53-61: selects CPU construction sub-algorithms; the built index is identical, so the
        # GPU builder has nothing to choose.
        return self

    def device(self, device_id: int) -> "FmIndexConfig":
        self._device = int(device_id)
        return self

    def acceleration_structures(self, pair_lines=None, jump_entry_bytes=None, top_table_depth=None,
                                aux_budget_bytes=None, full_suffix_array=None, text_units=None, seed_symbols=None,
                                seed_load_percent=None, inverse_suffix_array=None, reference_table_layout=None) -> "FmIndexConfig":
        """gdx_build_options_t: which derived structures the index carries beside the reference's arrays
        (results are identical with any combination); None keeps the default."""
        self._build = dict(pair_lines=pair_lines, jump_entry_bytes=jump_entry_bytes, top_table_depth=top_table_depth,
                           aux_budget_bytes=aux_budget_bytes, full_suffix_array=full_suffix_array, text_units=text_units,
                           seed_symbols=seed_symbols, seed_load_percent=seed_load_percent,
                           inverse_suffix_array=inverse_suffix_array, reference_table_layout=reference_table_layout)
        return self

    def construct_index(self, texts, alphabet: Alphabet) -> "FmIndex":
        """src/config.rs:63-69"""
        texts = [bytes(t) for t in texts]
        tbuf, toff = pack_queries(texts)
        lib = _lib.load()
        handle = C.c_void_p()
        tab = np.ascontiguousarray(alphabet.io_to_dense_table, dtype=np.uint8)
        opts = build_options(**self._build)
        st = lib.gdx_index_build_ex(_p(tbuf, u8p), _p(toff, u64p), len(texts), _p(tab, u8p),
                                    alphabet.num_dense_symbols(), alphabet.num_searchable_dense_symbols(),
                                    self._sa_rate, self._depth, _WIDTHS[self.index_storage], self._device,
                                    C.byref(opts), C.byref(handle))
        _lib.check(st)
        return FmIndex(handle, alphabet)


class FmIndex:
    """src/lib.rs:89-100.  Owns a handle to the index in HBM."""

    def __init__(self, handle, alphabet: Alphabet):
        self._h = handle
        self._lib = _lib.load()
        self._alphabet = alphabet
        info = _lib.IndexInfo()
        _lib.check(self._lib.gdx_index_info(self._h, C.byref(info)))
        self.info = info

    def __del__(self):
        try:
            if self._h:
                self._lib.gdx_index_free(self._h)
                self._h = None
        except Exception:
            pass

    @classmethod
    def from_parts(cls, count, interleaved_blocks, n, sa_samples, sa_rate, border_keys, border_vals,
                   sentinel_indices, alphabet: Alphabet, lookup_depth=0, index_storage="u32", device=0,
                   table_kind="condensed", block_bits=64, options=None):
        """Import of the reference's logical arrays (include/gdx.h gdx_index_from_parts[_ex]); table_kind
        'condensed' | 'flat' and block_bits 64 | 512 select one of the reference's four table variants."""
        lib = _lib.load()
        count = np.ascontiguousarray(count, dtype=np.uint64)
        blocks = np.ascontiguousarray(interleaved_blocks, dtype=np.uint64)
        sa_samples = np.ascontiguousarray(sa_samples, dtype=np.uint32)
        bk = np.ascontiguousarray(border_keys, dtype=np.uint64)
        bv = np.ascontiguousarray(border_vals, dtype=np.uint64)
        si = np.ascontiguousarray(sentinel_indices, dtype=np.uint64)
        tab = np.ascontiguousarray(alphabet.io_to_dense_table, dtype=np.uint8)
        handle = C.c_void_p()
        opts = options if options is not None else build_options()
        st = lib.gdx_index_from_parts_ex2({"condensed": 0, "flat": 1}[table_kind], int(block_bits),
                                          _p(count, u64p), _p(blocks, u64p), int(n), _p(sa_samples, u32p), int(sa_rate),
                                          _p(bk, u64p), _p(bv, u64p), _p(si, u64p), si.size, _p(tab, u8p),
                                          alphabet.num_dense_symbols(), alphabet.num_searchable_dense_symbols(),
                                          int(lookup_depth), _WIDTHS[index_storage], int(device), C.byref(opts),
                                          C.byref(handle))
        _lib.check(st)
        return cls(handle, alphabet)

    # ---- this implementation's own knobs (include/gdx.h gdx_index_aux / gdx_query_options_t) ------
    def aux(self) -> dict:
        a = _lib.IndexAux()
        _lib.check(self._lib.gdx_index_aux(self._h, C.byref(a)))
        return {f: int(getattr(a, f)) for f, _ in a._fields_}

    def seed_info(self) -> dict:
        """The seed table of the index (gdx_index_seed_info); k == 0: none."""
        out = (C.c_uint64 * 8)()
        _lib.check(self._lib.gdx_index_seed_info(self._h, out))
        names = ("k", "buckets", "single_entries", "interval_entries", "overflowed_buckets", "max_displacement", "bytes",
                 "tag_bits")
        info = {n: int(v) for n, v in zip(names, out)}
        rec = (C.c_uint64 * 4)()  # (repeats of two to four copies with a record of their own: gdx_index_seed_records)
        _lib.check(self._lib.gdx_index_seed_records(self._h, rec))
        info["pair_records"], info["quad_records"] = int(rec[0]), int(rec[1])
        return info

    def set_query_options(self, search_kernel=None, search_lanes=None, load_policy=None, length_schedule=None,
                          locate_kernel=None, locate_jump_walk=None, search_defer_after=None, search_fast=None,
                          search_exact=None, max_hits_per_query=None, search_seed=None) -> None:
        """Kernel variant of the query calls on this handle; None = default.  Results never depend on it."""
        o = _lib.QueryOptions()
        self._lib.gdx_query_options_init(C.byref(o))
        names = {"pair": 2, "quad": 0, "lane": 1}
        if search_kernel is not None:
            o.search_kernel = names.get(search_kernel, search_kernel)
        if search_lanes is not None:
            o.search_lanes = int(search_lanes)
        if load_policy is not None:
            o.load_policy = int(load_policy)
        if length_schedule is not None:
            o.length_schedule = int(length_schedule)
        if locate_kernel is not None:
            o.locate_kernel = {"queue": 0, "lane": 1, "pair": 2}.get(locate_kernel, locate_kernel)
        if locate_jump_walk is not None:
            o.locate_jump_walk = int(bool(locate_jump_walk))
        if search_defer_after is not None:
            o.search_defer_after = int(search_defer_after)
        if search_fast is not None:
            o.search_fast = int(search_fast)  # False / True / 2 (jumps over up to 16 rows)
        if search_exact is not None:
            o.search_exact = int(bool(search_exact))
        if max_hits_per_query is not None:
            o.max_hits_per_query = int(max_hits_per_query)  # host-pointer locate calls: locate(q).take(k)
        if search_seed is not None:
            o.search_seed = int(bool(search_seed))
        _lib.check(self._lib.gdx_index_set_query_options(self._h, C.byref(o)))

    def rebuild_aux(self, **kw) -> None:
        """bench only (gdx_bench.h): rebuild pair lines / jump / top tables with other build options"""
        o = build_options(**kw)
        _lib.check(self._lib.gdx_index_rebuild_aux(self._h, C.byref(o)))
        _lib.check(self._lib.gdx_index_info(self._h, C.byref(self.info)))

    # ---- lib.rs:296-327 (own file format, see include/gdx.h) -------------------------------------
    def save_to_file(self, path) -> None:
        _lib.check(self._lib.gdx_index_save(self._h, str(path).encode()))

    @classmethod
    def load_from_file(cls, path, alphabet: Alphabet, device=0, options=None) -> "FmIndex":
        lib = _lib.load()
        handle = C.c_void_p()
        opts = options if options is not None else build_options()
        _lib.check(lib.gdx_index_load_ex(str(path).encode(), int(device), C.byref(opts), C.byref(handle)))
        ix = cls(handle, alphabet)
        if ix.info.sigma != alphabet.num_dense_symbols():
            raise ValueError("the file was written for a different alphabet")
        return ix

    # ---- lib.rs:283-294 ----------------------------------------------------------------------
    def alphabet(self) -> Alphabet:
        return self._alphabet

    def num_texts(self) -> int:
        return int(self.info.num_texts)

    def total_text_len(self) -> int:
        return int(self.info.total_text_len)

    # ---- raw (numpy) entry points -------------------------------------------------------------
    def cursors_raw(self, qbuf, qoff, strict=True):
        """-> (start u64[nq], end u64[nq], status u8[nq])"""
        nq = qoff.size - 1
        s = np.zeros(nq, dtype=np.uint64)
        e = np.zeros(nq, dtype=np.uint64)
        status = np.zeros(nq, dtype=np.uint8)
        st = self._lib.gdx_cursors_for_many_queries(self._h, _p(qbuf, u8p), _p(qoff, u64p), nq, _p(s, u64p),
                                                    _p(e, u64p), _p(status, u8p))
        _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
        return s, e, status

    def count_raw(self, qbuf, qoff, strict=True):
        nq = qoff.size - 1
        counts = np.zeros(nq, dtype=np.uint64)
        status = np.zeros(nq, dtype=np.uint8)
        st = self._lib.gdx_count_many(self._h, _p(qbuf, u8p), _p(qoff, u64p), nq, _p(counts, u64p), _p(status, u8p))
        _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
        return counts, status

    def locate_raw(self, qbuf, qoff, strict=True):
        """-> (hit_offsets u64[nq+1], text_ids u64[total], positions u64[total], status)"""
        nq = qoff.size - 1
        off = np.zeros(nq + 1, dtype=np.uint64)
        status = np.zeros(nq, dtype=np.uint8)
        total = C.c_uint64(0)
        allow = (_lib.GDX_ERR_CAPACITY,) + (() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
        st = self._lib.gdx_locate_many(self._h, _p(qbuf, u8p), _p(qoff, u64p), nq, _p(off, u64p), None, 0,
                                       C.byref(total), _p(status, u8p))
        _lib.check(st, allow=allow)
        hits = np.zeros((max(total.value, 1), 2), dtype=np.uint64)
        if total.value:
            st = self._lib.gdx_locate_many(self._h, _p(qbuf, u8p), _p(qoff, u64p), nq, _p(off, u64p),
                                           hits.ctypes.data_as(C.POINTER(_lib.HitStruct)), total.value,
                                           C.byref(total), _p(status, u8p))
            _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
        hits = hits[: total.value]
        return off, hits[:, 0].copy(), hits[:, 1].copy(), status

    def _layout(self, packed, uniform_len):
        lay = _lib.QueryLayout()
        self._lib.gdx_query_layout_init(C.byref(lay))
        lay.packed, lay.uniform_len = (1 if packed else 0), int(uniform_len)
        return lay

    def count_layout_raw(self, qbuf, qoff, nq, packed=False, uniform_len=0, strict=True):
        """gdx_count_many_layout: a packed (2-bit) and / or uniform (no offsets: qoff may be None) host batch"""
        counts = np.zeros(nq, dtype=np.uint64)
        status = np.zeros(nq, dtype=np.uint8)
        lay = self._layout(packed, uniform_len)
        st = self._lib.gdx_count_many_layout(self._h, _p(qbuf, u8p), _p(qoff, u64p) if qoff is not None else None, nq,
                                             C.byref(lay), _p(counts, u64p), _p(status, u8p))
        _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
        return counts, status

    def cursors_layout_raw(self, qbuf, qoff, nq, packed=False, uniform_len=0, strict=True):
        s = np.zeros(nq, dtype=np.uint64)
        e = np.zeros(nq, dtype=np.uint64)
        status = np.zeros(nq, dtype=np.uint8)
        lay = self._layout(packed, uniform_len)
        st = self._lib.gdx_cursors_for_many_queries_layout(self._h, _p(qbuf, u8p), _p(qoff, u64p) if qoff is not None else None,
                                                           nq, C.byref(lay), _p(s, u64p), _p(e, u64p), _p(status, u8p))
        _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
        return s, e, status

    def locate_layout_raw(self, qbuf, qoff, nq, packed=False, uniform_len=0, strict=True):
        """gdx_locate_many_alloc_layout -> (hit_offsets, text_ids, positions, status)"""
        off = np.zeros(nq + 1, dtype=np.uint64)
        status = np.zeros(nq, dtype=np.uint8)
        total = C.c_uint64(0)
        ptr = C.POINTER(_lib.HitStruct)()
        lay = self._layout(packed, uniform_len)
        st = self._lib.gdx_locate_many_alloc_layout(self._h, _p(qbuf, u8p), _p(qoff, u64p) if qoff is not None else None, nq,
                                                    C.byref(lay), _p(off, u64p), C.byref(ptr), C.byref(total), _p(status, u8p))
        try:
            _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
            n = total.value
            hits = np.ctypeslib.as_array(C.cast(ptr, u64p), shape=(max(n, 1) * 2,))[: 2 * n].reshape(n, 2).copy() \
                if n else np.zeros((0, 2), dtype=np.uint64)
        finally:
            if ptr:
                self._lib.gdx_free_hits(ptr)
        return off, hits[:, 0].copy(), hits[:, 1].copy(), status

    def locate_layout32_raw(self, qbuf, qoff, nq, packed=False, uniform_len=0, strict=True, qbuf_ptr=None):
        """gdx_locate_many_alloc_layout32 -> (hit_offsets u32, text_ids u32, positions u32, status): copies of the library's
        pinned arrays (a caller that cares for speed reads them in place and gives them back with gdx_free_hits32).
        qbuf_ptr: the address of the query buffer when it is not a numpy array (pinned memory of another owner)"""
        status = np.zeros(nq, dtype=np.uint8)
        res = _lib.Hits32()
        lay = self._layout(packed, uniform_len)
        qp = C.cast(C.c_void_p(qbuf_ptr), u8p) if qbuf_ptr is not None else _p(qbuf, u8p)
        st = self._lib.gdx_locate_many_alloc_layout32(self._h, qp, _p(qoff, u64p) if qoff is not None else None, nq,
                                                      C.byref(lay), C.byref(res), _p(status, u8p))
        try:
            _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
            n = res.total_hits
            off = np.ctypeslib.as_array(res.hit_offsets, shape=(nq + 1,)).copy()
            hits = np.ctypeslib.as_array(res.hits, shape=(max(n, 1) * 2,))[: 2 * n].reshape(n, 2).copy() \
                if n else np.zeros((0, 2), dtype=np.uint32)
        finally:
            self._lib.gdx_free_hits32(C.byref(res))
        return off, hits[:, 0].copy(), hits[:, 1].copy(), status

    def locate_alloc_raw(self, qbuf, qoff, strict=True):
        """gdx_locate_many_alloc (one pass) -> (hit_offsets, text_ids, positions, status)"""
        nq = qoff.size - 1
        off = np.zeros(nq + 1, dtype=np.uint64)
        status = np.zeros(nq, dtype=np.uint8)
        total = C.c_uint64(0)
        ptr = C.POINTER(_lib.HitStruct)()
        st = self._lib.gdx_locate_many_alloc(self._h, _p(qbuf, u8p), _p(qoff, u64p), nq, _p(off, u64p), C.byref(ptr),
                                             C.byref(total), _p(status, u8p))
        try:
            _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
            n = total.value
            hits = np.ctypeslib.as_array(C.cast(ptr, u64p), shape=(max(n, 1) * 2,))[: 2 * n].reshape(n, 2).copy() \
                if n else np.zeros((0, 2), dtype=np.uint64)
        finally:
            if ptr:
                self._lib.gdx_free_hits(ptr)
        return off, hits[:, 0].copy(), hits[:, 1].copy(), status

    def locate_intervals_raw(self, starts, ends):
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        ends = np.ascontiguousarray(ends, dtype=np.uint64)
        m = starts.size
        off = np.zeros(m + 1, dtype=np.uint64)
        total = C.c_uint64(0)
        st = self._lib.gdx_cursor_locate_many(self._h, _p(starts, u64p), _p(ends, u64p), m, _p(off, u64p), None, 0,
                                              C.byref(total))
        _lib.check(st, allow=(_lib.GDX_ERR_CAPACITY,))
        hits = np.zeros((max(total.value, 1), 2), dtype=np.uint64)
        if total.value:
            st = self._lib.gdx_cursor_locate_many(self._h, _p(starts, u64p), _p(ends, u64p), m, _p(off, u64p),
                                                  hits.ctypes.data_as(C.POINTER(_lib.HitStruct)), total.value,
                                                  C.byref(total))
            _lib.check(st)
        hits = hits[: total.value]
        return off, hits[:, 0].copy(), hits[:, 1].copy()

    def extend_front_raw(self, starts, ends, io_symbols, strict=True):
        s = np.array(starts, dtype=np.uint64)
        e = np.array(ends, dtype=np.uint64)
        sym = np.ascontiguousarray(io_symbols, dtype=np.uint8)
        status = np.zeros(s.size, dtype=np.uint8)
        st = self._lib.gdx_cursor_extend_front_many(self._h, _p(s, u64p), _p(e, u64p), _p(sym, u8p), s.size,
                                                    _p(status, u8p))
        _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
        return s, e, status

    def extend_front_strings_raw(self, starts, ends, qbuf, qoff, status=None, strict=True):
        """gdx_cursor_extend_front_strings: cursor i is extended by the whole string i (right to left)."""
        s = np.array(starts, dtype=np.uint64)
        e = np.array(ends, dtype=np.uint64)
        qbuf = np.ascontiguousarray(qbuf, dtype=np.uint8)
        qoff = np.ascontiguousarray(qoff, dtype=np.uint64)
        st_arr = np.zeros(s.size, dtype=np.uint8) if status is None else np.array(status, dtype=np.uint8)
        st = self._lib.gdx_cursor_extend_front_strings(self._h, _p(s, u64p), _p(e, u64p), _p(qbuf, u8p),
                                                       _p(qoff, u64p), s.size, _p(st_arr, u8p))
        _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
        return s, e, st_arr

    def rank_many(self, symbols, idx):
        """TextWithRankSupport::rank (text_with_rank_support/mod.rs:106-110), batched."""
        sym = np.ascontiguousarray(symbols, dtype=np.uint8)
        ii = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.zeros(sym.size, dtype=np.uint64)
        _lib.check(self._lib.gdx_rank_many(self._h, _p(sym, u8p), _p(ii, u64p), sym.size, _p(out, u64p)))
        return out

    def symbol_at_many(self, idx):
        ii = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.zeros(ii.size, dtype=np.uint8)
        _lib.check(self._lib.gdx_symbol_at_many(self._h, _p(ii, u64p), ii.size, _p(out, u8p)))
        return out

    # ---- lib.rs:147-246 -----------------------------------------------------------------------
    def count(self, query) -> int:
        return int(self.count_many([query])[0])

    def count_many(self, queries):
        qbuf, qoff = pack_queries(queries)
        return self.count_raw(qbuf, qoff)[0]

    def locate(self, query):
        return self.locate_many([query])[0]

    def locate_many(self, queries):
        qbuf, qoff = pack_queries(queries)
        off, t, p, _ = self.locate_raw(qbuf, qoff)
        t, p = t.tolist(), p.tolist()
        return [[Hit(t[h], p[h]) for h in range(int(off[q]), int(off[q + 1]))] for q in range(qoff.size - 1)]

    def cursor_empty(self) -> "Cursor":
        s = C.c_uint64(0)
        e = C.c_uint64(0)
        _lib.check(self._lib.gdx_cursor_empty(self._h, C.byref(s), C.byref(e)))
        return Cursor(self, s.value, e.value)

    def cursor_for_query(self, query) -> "Cursor":
        return self.cursors_for_many_queries([query])[0]

    def cursors_for_many_queries(self, queries):
        qbuf, qoff = pack_queries(queries)
        s, e, _ = self.cursors_raw(qbuf, qoff)
        return [Cursor(self, int(a), int(b)) for a, b in zip(s, e)]

    # ---- exports -------------------------------------------------------------------------------
    def export_count(self):
        out = np.zeros(self.info.sigma + 1, dtype=np.uint64)
        _lib.check(self._lib.gdx_index_export_count(self._h, _p(out, u64p)))
        return out

    def export_bwt(self):
        out = np.zeros(max(self.total_text_len(), 1), dtype=np.uint8)
        _lib.check(self._lib.gdx_index_export_bwt(self._h, _p(out, u8p)))
        return out[: self.total_text_len()]

    def export_sa_samples(self):
        m = -(-self.total_text_len() // int(self.info.sa_rate))
        out = np.zeros(max(m, 1), dtype=np.uint32)
        _lib.check(self._lib.gdx_index_export_sa_samples(self._h, _p(out, u32p)))
        return out[:m]

    def export_borders(self):
        k = np.zeros(self.num_texts(), dtype=np.uint64)
        v = np.zeros(self.num_texts(), dtype=np.uint64)
        _lib.check(self._lib.gdx_index_export_borders(self._h, _p(k, u64p), _p(v, u64p)))
        return k, v

    def export_sentinel_indices(self):
        out = np.zeros(self.num_texts(), dtype=np.uint64)
        _lib.check(self._lib.gdx_index_export_sentinel_indices(self._h, _p(out, u64p)))
        return out

    def export_lookup_table(self, depth):
        entries = int(self.info.n_searchable) ** depth
        out = np.zeros((entries, 2), dtype=np.uint32)
        _lib.check(self._lib.gdx_index_export_lookup_table(self._h, depth, _p(out, u32p)))
        return out

    def export_condensed_table(self):
        n1 = self.total_text_len() + 1
        sigma = int(self.info.sigma)
        nbits = max(1, (sigma - 1).bit_length())
        blocks = np.zeros(-(-n1 // 64) * nbits, dtype=np.uint64)
        bo = np.zeros(-(-n1 // 64) * sigma, dtype=np.uint16)
        sbo = np.zeros(-(-n1 // 65536) * sigma, dtype=np.uint32)
        _lib.check(self._lib.gdx_index_export_condensed_table(self._h, _p(blocks, u64p), _p(bo, u16p), _p(sbo, u32p)))
        return blocks, bo, sbo

    def export_reference_table(self):
        """(interleaved blocks u64, superblock offsets u32) of an index built with reference_table_layout"""
        nw, ns = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self._lib.gdx_index_export_reference_table(self._h, None, 0, C.byref(nw), None, 0, C.byref(ns)))
        blocks = np.zeros(max(nw.value, 1), dtype=np.uint64)
        sbo = np.zeros(max(ns.value, 1), dtype=np.uint32)
        _lib.check(self._lib.gdx_index_export_reference_table(self._h, _p(blocks, u64p), blocks.size, C.byref(nw), _p(sbo, u32p),
                                                              sbo.size, C.byref(ns)))
        return blocks[: nw.value], sbo[: ns.value]

    def build_stats(self):
        st = _lib.BuildStats()
        _lib.check(self._lib.gdx_index_build_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}


class Cursor:
    """src/cursor.rs:16-73: the currently searched query as a half-open suffix-array interval."""

    def __init__(self, index: FmIndex, start: int, end: int):
        self.index = index
        self.start = start
        self.end = end

    def extend_query_front(self, symbol) -> None:
        """cursor.rs:34-38; raises where the reference panics (symbol not in the alphabet)."""
        sym = symbol if isinstance(symbol, int) else bytes(symbol)[0]
        s, e, _ = self.index.extend_front_raw([self.start], [self.end], [sym])
        self.start, self.end = int(s[0]), int(e[0])

    def extend_query_front_by(self, symbols) -> None:
        """extend_query_front for every symbol of `symbols`, last symbol first, in one launch"""
        qbuf, qoff = pack_queries([bytes(symbols)])
        s, e, _ = self.index.extend_front_strings_raw([self.start], [self.end], qbuf, qoff)
        self.start, self.end = int(s[0]), int(e[0])

    def count(self) -> int:
        return self.end - self.start  # cursor.rs:61-63

    def interval(self):
        return self.start, self.end

    def locate(self):
        off, t, p = self.index.locate_intervals_raw([self.start], [self.end])
        return [Hit(int(a), int(b)) for a, b in zip(t, p)]

    def __copy__(self):
        return Cursor(self.index, self.start, self.end)  # cursor.rs:22-28 Cursor is Copy


class PartitionedFmIndex:
    """gdx_parts_* (include/gdx.h): a collection of texts beyond 2^32 - 1 symbols -- the reference's `IndexStorage = i64`
    case (construction/mod.rs:225-252) -- as several 32-bit indexes cut at text borders.  count / locate only: counts
    and hit sets are the reference's; there is no single suffix-array interval of the whole collection, and a query's
    hits come part after part."""

    def __init__(self, handle, alphabet: Alphabet):
        self._h = handle
        self._lib = _lib.load()
        self._alphabet = alphabet
        out = (C.c_uint64 * 4)()
        _lib.check(self._lib.gdx_parts_info(self._h, out))
        self.num_parts, self._total_len, self._num_texts, self.device_bytes = (int(x) for x in out)

    def __del__(self):
        try:
            if self._h:
                self._lib.gdx_parts_free(self._h)
                self._h = None
        except Exception:
            pass

    @classmethod
    def construct(cls, texts, alphabet: Alphabet, sa_rate=4, lookup_depth=0, device=0, max_part_symbols=0, options=None):
        texts = [bytes(t) for t in texts]
        tbuf, toff = pack_queries(texts)
        return cls._build(tbuf.ctypes.data_as(C.c_void_p), 0, toff, len(texts), alphabet, sa_rate, lookup_depth, device,
                          max_part_symbols, options)

    @classmethod
    def _build(cls, texts_ptr, on_device, toff, n_texts, alphabet, sa_rate, lookup_depth, device, max_part_symbols, options):
        lib = _lib.load()
        handle = C.c_void_p()
        tab = np.ascontiguousarray(alphabet.io_to_dense_table, dtype=np.uint8)
        opts = options if options is not None else build_options()
        toff = np.ascontiguousarray(toff, dtype=np.uint64)
        _lib.check(lib.gdx_parts_build(texts_ptr, int(on_device), _p(toff, u64p), int(n_texts), _p(tab, u8p),
                                       alphabet.num_dense_symbols(), alphabet.num_searchable_dense_symbols(), int(sa_rate),
                                       int(lookup_depth), int(device), int(max_part_symbols), C.byref(opts), C.byref(handle)))
        return cls(handle, alphabet)

    def num_texts(self) -> int:
        return self._num_texts

    def total_text_len(self) -> int:
        return self._total_len

    def set_query_options(self, **kw) -> None:
        o = _lib.QueryOptions()
        self._lib.gdx_query_options_init(C.byref(o))
        known = {name for name, _ in _lib.QueryOptions._fields_} - {"struct_size"}
        for k, v in kw.items():
            if k not in known:  # (setattr on a ctypes struct would take any name and change nothing)
                raise TypeError(f"unknown query option {k!r}; gdx_query_options_t has {sorted(known)}")
            setattr(o, k, int(v))
        _lib.check(self._lib.gdx_parts_set_query_options(self._h, C.byref(o)))

    def count_raw(self, qbuf, qoff, strict=True):
        nq = qoff.size - 1
        counts = np.zeros(nq, dtype=np.uint64)
        status = np.zeros(nq, dtype=np.uint8)
        st = self._lib.gdx_parts_count_many(self._h, _p(qbuf, u8p), _p(qoff, u64p), nq, _p(counts, u64p), _p(status, u8p))
        _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
        return counts, status

    def locate_raw(self, qbuf, qoff, strict=True):
        """-> (hit_offsets u64[nq+1], text_ids u64[total], positions u64[total], status)"""
        nq = qoff.size - 1
        off = np.zeros(nq + 1, dtype=np.uint64)
        status = np.zeros(nq, dtype=np.uint8)
        total = C.c_uint64(0)
        ptr = C.POINTER(_lib.HitStruct)()
        st = self._lib.gdx_parts_locate_many_alloc(self._h, _p(qbuf, u8p), _p(qoff, u64p), nq, _p(off, u64p), C.byref(ptr),
                                                   C.byref(total), _p(status, u8p))
        try:
            _lib.check(st, allow=() if strict else (_lib.GDX_ERR_QUERY_STATUS,))
            n = total.value
            hits = np.ctypeslib.as_array(C.cast(ptr, u64p), shape=(max(n, 1) * 2,))[: 2 * n].reshape(n, 2).copy() \
                if n else np.zeros((0, 2), dtype=np.uint64)
        finally:
            if ptr:
                self._lib.gdx_free_hits(ptr)
        return off, hits[:, 0].copy(), hits[:, 1].copy(), status

    def count_many(self, queries):
        return self.count_raw(*pack_queries(queries))[0]

    def count(self, query) -> int:
        return int(self.count_many([query])[0])

    def locate_many(self, queries):
        qbuf, qoff = pack_queries(queries)
        off, t, p, _ = self.locate_raw(qbuf, qoff)
        t, p = t.tolist(), p.tolist()
        return [[Hit(t[h], p[h]) for h in range(int(off[q]), int(off[q + 1]))] for q in range(qoff.size - 1)]

    def locate(self, query):
        return self.locate_many([query])[0]
