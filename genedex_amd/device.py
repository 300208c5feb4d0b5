"""Device-resident use of libgdx.so: indexes built from text that already sits in HBM, query sets
generated in HBM, and the `_dev` entry points of include/gdx.h driven on a torch stream.

torch is used for device memory, streams and events only (plumbing); every computation is a HIP
kernel of libgdx.so.  torch must be imported before libgdx.so is loaded so that both share one HIP
runtime (torch bundles its own libamdhip64.so.7).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .alphabet import Alphabet
from .index import FmIndex, _WIDTHS, _p
from .synth import split_lengths

u8p, u64p = _lib.u8p, _lib.u64p


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def synth_text(total: int, seed: int = 42, n_per_million: int = 10_000, device="cuda") -> torch.Tensor:
    """IO symbols of the synthetic text, generated in HBM (gdx_synth_text_dev)."""
    lib = _lib.load()
    out = torch.empty(max(total, 1), dtype=torch.uint8, device=device)
    _lib.check(lib.gdx_synth_text_dev(_ptr(out), total, seed, n_per_million, _stream()))
    return out


def build_index_from_device_text(io_text: torch.Tensor, text_lengths, alphabet: Alphabet, sa_rate=4, lookup_depth=0,
                                 index_storage="u32", options=None) -> FmIndex:
    lib = _lib.load()
    toff = np.zeros(len(text_lengths) + 1, dtype=np.uint64)
    np.cumsum(np.asarray(text_lengths, dtype=np.uint64), out=toff[1:])
    tab = np.ascontiguousarray(alphabet.io_to_dense_table, dtype=np.uint8)
    handle = C.c_void_p()
    torch.cuda.synchronize()
    from .index import build_options

    opts = options if options is not None else build_options()
    st = lib.gdx_index_build_dev_ex(_ptr(io_text), _p(toff, u64p), len(text_lengths), _p(tab, u8p),
                                    alphabet.num_dense_symbols(), alphabet.num_searchable_dense_symbols(), sa_rate,
                                    lookup_depth, _WIDTHS[index_storage], io_text.device.index or 0, C.byref(opts),
                                    C.byref(handle))
    _lib.check(st)
    return FmIndex(handle, alphabet)


class DeviceQueries:
    """A query set resident in HBM: qbuf (u8, padded to 8 bytes) + qoff (u64[nq+1]).
    packed: qbuf holds 2-bit codes, four symbols per byte (gdx.h "packed queries"; qoff still counts symbols);
    uniform_len != 0: every query has that many symbols and query i starts at symbol i * uniform_len (the search calls then
    pass no offsets: gdx_query_layout_t).  total_bytes = the bytes of qbuf the queries occupy in this form."""

    def __init__(self, qbuf: torch.Tensor, qoff: torch.Tensor, nq: int, total_bytes: int, packed: bool = False,
                 uniform_len: int = 0):
        self.qbuf, self.qoff, self.nq, self.total_bytes = qbuf, qoff, nq, total_bytes
        self.packed, self.uniform_len = packed, uniform_len

    def layout(self):
        """(gdx_query_layout_t or None, offsets pointer) for the *_layout_dev calls"""
        if not self.packed and not self.uniform_len:
            return None, _ptr(self.qoff)
        lay = _lib.QueryLayout()
        _lib.load().gdx_query_layout_init(C.byref(lay))
        lay.packed = 1 if self.packed else 0
        lay.uniform_len = int(self.uniform_len)
        return lay, (C.c_void_p(0) if self.uniform_len else _ptr(self.qoff))

    def total_symbols(self) -> int:
        return self.total_bytes * 4 if self.packed else self.total_bytes  # (packed: rounded up to whole bytes)

    def as_uniform(self, length: int) -> "DeviceQueries":
        """the same batch declared uniform (every query `length` symbols, back to back from symbol 0): checked on the device"""
        n = self.nq
        if n:
            want = torch.arange(0, n + 1, device=self.qoff.device, dtype=torch.int64) * length
            if not torch.equal(self.qoff[: n + 1], want):
                raise ValueError("the batch is not uniform: query i must span [i * length, (i + 1) * length)")
        return DeviceQueries(self.qbuf, self.qoff, n, self.total_bytes, self.packed, length)

    def as_packed(self, index: FmIndex) -> "DeviceQueries":
        """2-bit form of the batch made on the device (gdx_pack_queries_dev); raises if a query has a symbol outside the
        four searchable ones (such queries are the packed form's exceptions and go through the ASCII calls)"""
        lib = _lib.load()
        n_sym = int(self.qoff[self.nq].item()) if self.nq else 0
        nbytes = int(lib.gdx_packed_bytes(n_sym))
        packed = torch.zeros(nbytes, dtype=torch.uint8, device=self.qbuf.device)
        bad = torch.zeros(1, dtype=torch.int64, device=self.qbuf.device)
        _lib.check(lib.gdx_pack_queries_dev(index._h, _ptr(self.qbuf), n_sym, _ptr(packed), C.c_void_p(0), _ptr(bad), _stream()))
        if int(bad.item()):
            raise ValueError(f"{int(bad.item())} symbols outside the four searchable ones: not expressible in 2 bits")
        return DeviceQueries(packed, self.qoff, self.nq, (n_sym + 3) // 4, True, self.uniform_len)

    @classmethod
    def synth(cls, io_text: torch.Tensor, text_lengths, nq: int, len_min: int, len_max: int,
              sampled_per_million: int, seed: int = 43) -> "DeviceQueries":
        lib = _lib.load()
        dev = io_text.device
        toff_h = np.zeros(len(text_lengths) + 1, dtype=np.int64)
        np.cumsum(np.asarray(text_lengths, dtype=np.int64), out=toff_h[1:])
        toff = torch.from_numpy(toff_h).to(dev)
        cap = (nq * len_max + 8 + 7) // 8 * 8
        qbuf = torch.zeros(cap, dtype=torch.uint8, device=dev)
        qoff = torch.empty(nq + 1, dtype=torch.int64, device=dev)
        total = C.c_uint64(0)
        _lib.check(lib.gdx_synth_queries_dev(_ptr(io_text), _ptr(toff), len(text_lengths), nq, len_min, len_max,
                                             sampled_per_million, seed, _ptr(qoff), _ptr(qbuf), cap,
                                             C.byref(total), _stream()))
        return cls(qbuf, qoff, nq, total.value)

    @classmethod
    def from_host(cls, qbuf: np.ndarray, qoff: np.ndarray, device="cuda") -> "DeviceQueries":
        nq = qoff.size - 1
        total = int(qoff[-1])
        pad = np.zeros((total + 8 + 7) // 8 * 8, dtype=np.uint8)
        pad[:total] = qbuf[:total]
        return cls(torch.from_numpy(pad).to(device), torch.from_numpy(qoff.astype(np.int64)).to(device), nq, total)

    def slice(self, lo: int, hi: int) -> "DeviceQueries":
        """queries [lo, hi) as a view: the same byte buffer, a window of the offsets (no copy)"""
        if self.packed or self.uniform_len:
            raise ValueError("slice(): plain batches only (a packed / uniform shard is packed from the sliced plain batch)")
        off = self.qoff[lo:hi + 1]
        nbytes = int((off[-1] - off[0]).item()) if hi > lo else 0
        return DeviceQueries(self.qbuf, off, hi - lo, nbytes)

    def copy_slice(self, lo: int, hi: int) -> "DeviceQueries":
        """queries [lo, hi) as a batch of their own (bytes copied, offsets from 0): what a rank holds of a sharded batch"""
        if self.packed or self.uniform_len:
            raise ValueError("copy_slice(): plain batches only")
        off = self.qoff[lo:hi + 1]
        b0, b1 = (int(off[0].item()), int(off[-1].item())) if hi > lo else (0, 0)
        buf = torch.zeros((b1 - b0 + 8 + 7) // 8 * 8, dtype=torch.uint8, device=self.qbuf.device)
        buf[: b1 - b0] = self.qbuf[b0:b1]
        return DeviceQueries(buf, off - off[0] if hi > lo else off.clone(), hi - lo, b1 - b0)

    def host_slice(self, first: int, count: int):
        """(qbuf, qoff) of queries [first, first+count) as numpy arrays (for the CPU baseline / checks)."""
        off = self.qoff[first:first + count + 1].cpu().numpy().astype(np.uint64)
        b0, b1 = int(off[0]), int(off[-1])
        buf = self.qbuf[b0:b1].cpu().numpy() if b1 > b0 else np.zeros(1, dtype=np.uint8)
        return buf, off - off[0]


class DeviceEngine:
    """count / locate passes over a DeviceQueries set, everything staying in HBM."""

    def __init__(self, index: FmIndex):
        self.index = index
        self.lib = _lib.load()
        self.h = index._h
        self.dev = torch.device("cuda", int(index.info.device_id))

    def alloc_outputs(self, nq: int, hint: bool = False):
        """hint=True adds the opaque locate-hint array: search() then fills it and locate() uses it (the device form
        of the fused locate_many; only valid while start / end stay as search() wrote them)."""
        d = self.dev
        out = {
            "start": torch.empty(nq, dtype=torch.int32, device=d),
            "end": torch.empty(nq, dtype=torch.int32, device=d),
            "status": torch.empty(nq, dtype=torch.uint8, device=d),
            "hit_offsets": torch.empty(nq + 1, dtype=torch.int64, device=d),
        }
        if hint:
            out["hint"] = torch.empty(nq, dtype=torch.int64, device=d)
        return out

    def search(self, q: DeviceQueries, out) -> None:
        """cursors_for_many_queries: intervals + status (the dominant kernel)."""
        lay, qoff = q.layout()
        if lay is not None:
            if "hint" in out:
                raise ValueError("hints go with plain batches")
            _lib.check(self.lib.gdx_cursors_for_many_queries_layout_dev(self.h, _ptr(q.qbuf), qoff, q.nq, C.byref(lay),
                                                                        _ptr(out["start"]), _ptr(out["end"]),
                                                                        _ptr(out["status"]), _stream()))
            return
        if "hint" in out:
            _lib.check(self.lib.gdx_cursors_for_many_queries_hint_dev(self.h, _ptr(q.qbuf), _ptr(q.qoff), q.nq,
                                                                      _ptr(out["start"]), _ptr(out["end"]),
                                                                      _ptr(out["status"]), _ptr(out["hint"]),
                                                                      _stream()))
            return
        _lib.check(self.lib.gdx_cursors_for_many_queries_dev(self.h, _ptr(q.qbuf), _ptr(q.qoff), q.nq,
                                                             _ptr(out["start"]), _ptr(out["end"]),
                                                             _ptr(out["status"]), _stream()))

    def count(self, q: DeviceQueries, counts: torch.Tensor, status: torch.Tensor) -> None:
        lay, qoff = q.layout()
        if lay is not None:
            _lib.check(self.lib.gdx_count_many_layout_dev(self.h, _ptr(q.qbuf), qoff, q.nq, C.byref(lay), _ptr(counts),
                                                          _ptr(status), _stream()))
            return
        _lib.check(self.lib.gdx_count_many_dev(self.h, _ptr(q.qbuf), _ptr(q.qoff), q.nq, _ptr(counts),
                                               _ptr(status), _stream()))

    def hit_offsets(self, out, m: int) -> None:
        _lib.check(self.lib.gdx_hit_offsets_dev(self.h, _ptr(out["start"]), _ptr(out["end"]), m,
                                                _ptr(out["hit_offsets"]), _stream()))

    def locate_workspace_bytes(self, total: int) -> int:
        return int(self.lib.gdx_locate_workspace_bytes(total))

    def locate(self, out, m: int, total: int, hits: torch.Tensor, workspace: torch.Tensor) -> None:
        """hits: int32[total, 2] = (text_id, position) per hit, in suffix-array order per query."""
        if "hint" in out:
            _lib.check(self.lib.gdx_locate_intervals_hint_dev(self.h, _ptr(out["start"]), _ptr(out["end"]), m,
                                                              _ptr(out["hit_offsets"]), total, _ptr(hits),
                                                              _ptr(workspace), _ptr(out["hint"]), _stream()))
            return
        _lib.check(self.lib.gdx_locate_intervals_dev(self.h, _ptr(out["start"]), _ptr(out["end"]), m,
                                                     _ptr(out["hit_offsets"]), total, _ptr(hits), _ptr(workspace),
                                                     _stream()))

    # ---- fused count + locate over 16-byte search records (gdx_locate_many_*_dev) -------------------------
    def alloc_records(self, nq: int) -> torch.Tensor:
        return torch.empty((max(nq, 1), 4), dtype=torch.int32, device=self.dev)

    def alloc_compact(self, nq: int) -> torch.Tensor:
        """compact results beside the records (gdx.h): int32[nq], the position of the only hit / -1 none / -2 see the record"""
        return torch.empty(max(nq, 1), dtype=torch.int32, device=self.dev)

    def locate_search(self, q: DeviceQueries, rec: torch.Tensor, compact: torch.Tensor = None) -> None:
        lay, qoff = q.layout()
        if lay is not None:
            if compact is not None:
                _lib.check(self.lib.gdx_locate_many_search_compact_layout_dev(self.h, _ptr(q.qbuf), qoff, q.nq, C.byref(lay),
                                                                              _ptr(rec), _ptr(compact), _stream()))
            else:
                _lib.check(self.lib.gdx_locate_many_search_layout_dev(self.h, _ptr(q.qbuf), qoff, q.nq, C.byref(lay), _ptr(rec),
                                                                      _stream()))
            return
        if compact is not None:
            _lib.check(self.lib.gdx_locate_many_search_compact_dev(self.h, _ptr(q.qbuf), _ptr(q.qoff), q.nq, _ptr(rec),
                                                                   _ptr(compact), _stream()))
            return
        _lib.check(self.lib.gdx_locate_many_search_dev(self.h, _ptr(q.qbuf), _ptr(q.qoff), q.nq, _ptr(rec), _stream()))

    def locate_search_totals(self, q: DeviceQueries, rec: torch.Tensor, compact: torch.Tensor, scan_ws: torch.Tensor,
                             totals: torch.Tensor, max_hits: int = 0) -> None:
        """gdx_locate_many_search_totals_compact_layout_dev: the compact search and the hit totals in one call"""
        lay, qoff = q.layout()
        _lib.check(self.lib.gdx_locate_many_search_totals_compact_layout_dev(
            self.h, _ptr(q.qbuf), qoff, q.nq, C.byref(lay) if lay is not None else None, max_hits, _ptr(rec), _ptr(compact),
            _ptr(scan_ws), _ptr(totals), _stream()))

    def locate_step(self, q: DeviceQueries, rec: torch.Tensor, compact, scan_ws: torch.Tensor, totals: torch.Tensor,
                    hit_offsets: torch.Tensor, hits: torch.Tensor, workspace: torch.Tensor, max_hits: int = 0,
                    event_after_search=None) -> None:
        """gdx_locate_many_step_compact_layout_dev: search, totals, offsets and hits in one call, no host round trip; hits
        beyond hits.shape[0] are not stored (totals[0], read later, tells); hit_offsets int32 = the narrow form;
        event_after_search: a torch.cuda.Event that has been recorded once (so that its handle exists)"""
        lay, qoff = q.layout()
        _lib.check(self.lib.gdx_locate_many_step_compact_layout_dev(
            self.h, _ptr(q.qbuf), qoff, q.nq, C.byref(lay) if lay is not None else None, max_hits, _ptr(rec),
            _ptr(compact) if compact is not None else None, _ptr(scan_ws), _ptr(totals), _ptr(hit_offsets),
            32 if hit_offsets.dtype == torch.int32 else 64, _ptr(hits), hits.shape[0], _ptr(workspace),
            C.c_void_p(event_after_search.cuda_event) if event_after_search is not None else None, _stream()))

    def locate_offsets(self, rec: torch.Tensor, nq: int, hit_offsets: torch.Tensor, max_hits: int = 0,
                       compact: torch.Tensor = None) -> None:
        """max_hits != 0: queries with more occurrences are counted but get no hit slots"""
        if compact is not None:
            _lib.check(self.lib.gdx_locate_many_offsets_compact_dev(self.h, _ptr(rec), _ptr(compact), nq, max_hits,
                                                                    _ptr(hit_offsets), _stream()))
            return
        _lib.check(self.lib.gdx_locate_many_offsets_capped_dev(self.h, _ptr(rec), nq, max_hits, _ptr(hit_offsets),
                                                               _stream()))

    def locate_hits(self, rec: torch.Tensor, nq: int, hit_offsets: torch.Tensor, total: int, hits: torch.Tensor,
                    workspace: torch.Tensor, compact: torch.Tensor = None) -> None:
        if compact is not None:
            _lib.check(self.lib.gdx_locate_many_hits_compact_dev(self.h, _ptr(rec), _ptr(compact), nq, _ptr(hit_offsets), total,
                                                                 _ptr(hits), _ptr(workspace), _stream()))
            return
        _lib.check(self.lib.gdx_locate_many_hits_dev(self.h, _ptr(rec), nq, _ptr(hit_offsets), total, _ptr(hits),
                                                     _ptr(workspace), _stream()))

    def totals_workspace_bytes(self, nq: int) -> int:
        return int(self.lib.gdx_locate_many_totals_workspace_bytes(nq))

    def locate_totals(self, rec: torch.Tensor, nq: int, scan_ws: torch.Tensor, totals: torch.Tensor, max_hits: int = 0,
                      compact: torch.Tensor = None) -> None:
        """gdx_locate_many_totals_compact_dev: totals = int64[2] (all hit slots, slots behind "see the record")"""
        _lib.check(self.lib.gdx_locate_many_totals_compact_dev(self.h, _ptr(rec), _ptr(compact) if compact is not None else None,
                                                               nq, max_hits, _ptr(scan_ws), _ptr(totals), _stream()))

    def locate_offsets_hits(self, rec: torch.Tensor, nq: int, scan_ws: torch.Tensor, hit_offsets: torch.Tensor, total: int,
                            rest: int, hits: torch.Tensor, workspace: torch.Tensor, max_hits: int = 0,
                            compact: torch.Tensor = None) -> None:
        """gdx_locate_many_offsets_hits_compact_dev: offsets + the hits the compact results answer in one pass, then the rest
        (hit_offsets of dtype int32: the narrow form, gdx_locate_many_offsets32_hits_compact_dev)"""
        call = (self.lib.gdx_locate_many_offsets32_hits_compact_dev if hit_offsets.dtype == torch.int32
                else self.lib.gdx_locate_many_offsets_hits_compact_dev)
        _lib.check(call(
            self.h, _ptr(rec), _ptr(compact) if compact is not None else None, nq, max_hits, _ptr(scan_ws), _ptr(hit_offsets),
            total, rest, _ptr(hits), _ptr(workspace) if workspace is not None else None, _stream()))

    def unpack_records(self, rec: torch.Tensor, nq: int, counts=None, status=None, compact=None) -> None:
        if compact is not None:
            _lib.check(self.lib.gdx_locate_many_unpack_compact_dev(self.h, _ptr(rec), _ptr(compact), nq,
                                                                   _ptr(counts) if counts is not None else None,
                                                                   _ptr(status) if status is not None else None, _stream()))
            return
        _lib.check(self.lib.gdx_locate_many_unpack_dev(self.h, _ptr(rec), nq,
                                                       _ptr(counts) if counts is not None else None,
                                                       _ptr(status) if status is not None else None, _stream()))

    def compact_split_hits(self, compact: torch.Tensor, nq: int, text_ids: torch.Tensor, positions: torch.Tensor) -> None:
        """gdx_compact_split_hits_dev: compact results -> uint8 text ids + int32 positions in the text (-1 none, -2 see the
        record); what the root of a multi-GPU gather turns a received shard into"""
        _lib.check(self.lib.gdx_compact_split_hits_dev(self.h, _ptr(compact), nq, _ptr(text_ids), _ptr(positions), _stream()))

    def wire_pack_workspace_bytes(self, nq: int) -> int:
        return int(self.lib.gdx_wire_pack_workspace_bytes(nq))

    def wire_pack(self, compact: torch.Tensor, hit_offsets: torch.Tensor, hits: torch.Tensor, nq: int, v: dict,
                  workspace: torch.Tensor) -> None:
        """gdx_wire_pack_dev: a located shard into its "found bitmap" wire form; v = dist.WireLayout.views(buffer)"""
        _lib.check(self.lib.gdx_wire_pack_dev(
            self.h, _ptr(compact), _ptr(hit_offsets), 32 if hit_offsets.dtype == torch.int32 else 64, _ptr(hits), nq,
            _ptr(v["bitmap"]), _ptr(v["tile_found"]), _ptr(v["found_pos"]), v["found_pos"].numel(), _ptr(v["exc_q"]),
            _ptr(v["exc_cnt"]), v["exc_q"].numel(), _ptr(v["exc_ids"]), _ptr(v["exc_pos"]), v["exc_ids"].numel(), _ptr(v["meta"]),
            _ptr(workspace), _stream()))

    def wire_split(self, v: dict, nq: int, text_ids: torch.Tensor, positions: torch.Tensor) -> None:
        """gdx_wire_split_dev: a received shard -> uint8 text ids + int32 positions (-1 none, -2 exception)"""
        _lib.check(self.lib.gdx_wire_split_dev(self.h, _ptr(v["bitmap"]), _ptr(v["tile_found"]), _ptr(v["found_pos"]),
                                               v["found_pos"].numel(), nq, _ptr(v["exc_q"]), _ptr(v["meta"]), v["exc_q"].numel(),
                                               _ptr(text_ids), _ptr(positions), _stream()))

    def compact_exceptions(self, compact: torch.Tensor, nq: int, queries: torch.Tensor, n: torch.Tensor) -> None:
        """gdx_compact_exceptions_dev: the queries that say "see the record" into `queries` (int32 / uint32 view, unordered, as
        many as it holds); n (int64[1]) = how many there are"""
        _lib.check(self.lib.gdx_compact_exceptions_dev(self.h, _ptr(compact), nq, _ptr(queries), queries.numel(), _ptr(n),
                                                       _stream()))

    # ---- batched cursor extension by strings (gdx_cursor_extend_front_strings_dev) ------------------------
    def cursor_extend_strings(self, start, end, qbuf, qbeg, qend, m, status=None, active_in=None, n_active_in=None,
                              active_out=None, n_active_out=None) -> None:
        opt = lambda t: _ptr(t) if t is not None else None  # noqa: E731
        _lib.check(self.lib.gdx_cursor_extend_front_strings_dev(self.h, _ptr(start), _ptr(end), _ptr(qbuf), _ptr(qbeg),
                                                                _ptr(qend), m, opt(status), opt(active_in),
                                                                opt(n_active_in), opt(active_out), opt(n_active_out),
                                                                _stream()))

    def cursor_extend_chunk(self, start, end, qbuf, qoff, m, chunk_symbols, chunk_index, status=None, active_in=None,
                            n_active_in=None, active_out=None, n_active_out=None) -> None:
        """gdx_cursor_extend_front_chunk_dev: chunk `chunk_index` (from the right) of every query, no edge arrays."""
        opt = lambda t: _ptr(t) if t is not None else None  # noqa: E731
        _lib.check(self.lib.gdx_cursor_extend_front_chunk_dev(self.h, _ptr(start), _ptr(end), _ptr(qbuf), _ptr(qoff), m,
                                                              int(chunk_symbols), int(chunk_index), opt(status),
                                                              opt(active_in), opt(n_active_in), opt(active_out),
                                                              opt(n_active_out), _stream()))

    def search_step_stats(self, q: DeviceQueries):
        """(LF steps, line fetches of all queries, fetch slots their wavefronts spent)"""
        steps = torch.zeros(3, dtype=torch.int64, device=self.dev)
        _lib.check(self.lib.gdx_search_step_stats_dev(self.h, _ptr(q.qbuf), _ptr(q.qoff), q.nq, _ptr(steps),
                                                      _stream()))
        return [int(x) for x in steps.tolist()]

    def aux_info(self) -> dict:
        """Acceleration structures of the index (gdx_index_aux_info)."""
        import ctypes

        out = (ctypes.c_uint32 * 4)()
        _lib.check(self.lib.gdx_index_aux_info(self.h, out))
        a = self.index.aux()
        return {"pair_lines": bool(out[0]), "jump_entry_bytes": int(out[1]), "top_table_depth": int(out[2]),
                "full_suffix_array": bool(out[3] & 1), "text_units": bool(out[3] & 2), "inverse_suffix_array": bool(out[3] & 4),
                "default_shape": bool(out[3] & 8),  # the library chose it: every option was left at its default
                "aux_bytes": a["aux_bytes"], "aux_budget_bytes": a["aux_budget_bytes"], "wide_permille": a["wide_permille"],
                "shrunk_by_budget": (a["wanted_jump_entry_bytes"], a["wanted_top_table_depth"])
                != (a["jump_entry_bytes"], a["top_table_depth"]),
                "seed": self.index.seed_info()}

    def search_lf_steps(self, q: DeviceQueries) -> int:
        return self.search_step_stats(q)[0]

    def locate_record_walks(self, rec, nq: int, hit_offsets, total: int, hits, workspace):
        """(walk steps executed with the records' hints, hits that walked)"""
        steps = torch.zeros(2, dtype=torch.int64, device=self.dev)
        _lib.check(self.lib.gdx_locate_many_hits_stats_dev(self.h, _ptr(rec), nq, _ptr(hit_offsets), total, _ptr(hits),
                                                           _ptr(workspace), _ptr(steps), _stream()))
        return [int(x) for x in steps.tolist()]

    def locate_walk_steps(self, out, m: int, total: int, hits: torch.Tensor, workspace: torch.Tensor) -> int:
        steps = torch.zeros(2, dtype=torch.int64, device=self.dev)
        _lib.check(self.lib.gdx_locate_step_stats_dev(self.h, _ptr(out["start"]), _ptr(out["end"]), m,
                                                      _ptr(out["hit_offsets"]), total, _ptr(hits), _ptr(workspace),
                                                      _ptr(steps), _stream()))
        return int(steps[0].item())


def measure_bandwidth(device="cuda", gib: float = 4.0, reps: int = 3):
    """Roofline denominators measured on this GPU: streaming copy and random line gathers (GB/s)."""
    lib = _lib.load()
    nbytes = int(gib * (1 << 30)) // 4096 * 4096
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    src.random_(0, 255)
    dst = torch.empty_like(src)
    sink = torch.zeros(1, dtype=torch.int32, device=device)
    res = {}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None or ms < best else best
        return best / 1e3

    t = timed(lambda: _lib.check(lib.gdx_bench_stream_copy(_ptr(dst), _ptr(src), nbytes, _stream())))
    res["stream_copy_GBps"] = 2 * nbytes / t / 1e9
    t = timed(lambda: _lib.check(lib.gdx_bench_stream_read(_ptr(src), nbytes, _ptr(sink), _stream())))
    res["stream_read_GBps"] = nbytes / t / 1e9
    n_acc = 1 << 27
    names = {0: "lane", 1: "group", 2: "lane_dependent"}
    for line in (64, 128):
        for mode in (0, 1, 2):
            t = timed(lambda: _lib.check(lib.gdx_bench_random_gather(_ptr(src), nbytes // line, line, n_acc, 7, mode,
                                                                     _ptr(sink), _stream())))
            res[f"gather{line}_{names[mode]}_GBps"] = n_acc * line / t / 1e9
            res[f"gather{line}_{names[mode]}_Glines_per_s"] = n_acc / t / 1e9
    return res


def genome_like_text(total: int, dev, seed: int = 7) -> torch.Tensor:
    """A text with the repeat structure of a genome instead of i.i.d. symbols: random base sequence, then copies --
    30 % of the text is made of duplicated segments (1 k .. 2 M symbols, 0.5 % substitutions), 3 % tandem repeats
    (unit 2..60), 1 % poly-A, 2 % runs of N (assembly gaps, up to total / 100)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    text = synth_text(total, seed=seed, n_per_million=100, device=dev)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)

    def place(n_symbols, make):
        done = 0
        while done < n_symbols:
            done += make()

    def duplication():
        ln = int(min(rng.integers(1_000, 2_000_000), total // 8))
        src, dst = int(rng.integers(0, total - ln)), int(rng.integers(0, total - ln))
        seg = text[src:src + ln].clone()
        n_mut = max(1, ln // 200)
        # distinct positions: a scatter with duplicate indices has no defined winner, and the text must be the same
        # text in every run
        at = torch.unique(torch.randint(0, ln, (n_mut,), device=dev, generator=g))
        seg[at] = acgt[torch.randint(0, 4, (at.numel(),), device=dev, generator=g)]
        text[dst:dst + ln] = seg
        return ln

    def tandem():
        unit = int(rng.integers(2, 61))
        ln = int(rng.integers(unit * 5, unit * 2000))
        dst = int(rng.integers(0, total - ln))
        u = acgt[torch.randint(0, 4, (unit,), device=dev, generator=g)]
        text[dst:dst + ln] = u.repeat(ln // unit + 1)[:ln]
        return ln

    def poly_a():
        ln = int(rng.integers(20, 5000))
        dst = int(rng.integers(0, total - ln))
        text[dst:dst + ln] = ord("A")
        return ln

    def gap():
        ln = int(rng.integers(1000, max(2000, total // 100)))
        dst = int(rng.integers(0, total - ln))
        text[dst:dst + ln] = ord("N")
        return ln

    place(int(0.30 * total), duplication)
    place(int(0.03 * total), tandem)
    place(int(0.01 * total), poly_a)
    place(int(0.02 * total), gap)
    return text


def hg38_text_lengths(total: int, n_texts: int = 24):
    return split_lengths(total, n_texts)


def build_parts_from_device_text(io_text: torch.Tensor, text_lengths, alphabet: Alphabet, sa_rate=4, lookup_depth=0,
                                 max_part_symbols=0, options=None):
    """gdx_parts_build on a text that already sits in HBM: a collection beyond 2^32 - 1 symbols as several 32-bit
    indexes cut at text borders (index.PartitionedFmIndex)."""
    from .index import PartitionedFmIndex

    toff = np.zeros(len(text_lengths) + 1, dtype=np.uint64)
    np.cumsum(np.asarray(text_lengths, dtype=np.uint64), out=toff[1:])
    torch.cuda.synchronize()
    return PartitionedFmIndex._build(_ptr(io_text), 1, toff, len(text_lengths), alphabet, sa_rate, lookup_depth,
                                     io_text.device.index or 0, max_part_symbols, options)
