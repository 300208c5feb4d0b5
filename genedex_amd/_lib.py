"""ctypes binding of libgdx.so (include/gdx.h, include/gdx_bench.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded the import of the
query API fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GDX_LIB_PATH: another build of the library, for A/B measurements of two versions on one box)
LIB_PATH = os.environ.get("GDX_LIB_PATH") or os.path.join(_HERE, "libgdx.so")

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p

GDX_OK = 0
GDX_ERR_INVALID_ARGUMENT = 1
GDX_ERR_INVALID_TEXT_SYMBOL = 2
GDX_ERR_TEXT_TOO_LONG = 3
GDX_ERR_DEVICE = 4
GDX_ERR_CAPACITY = 5
GDX_ERR_QUERY_STATUS = 6
GDX_ERR_UNSUPPORTED = 7

GDX_Q_OK = 0
GDX_Q_INVALID_SYMBOL = 1
GDX_Q_UNSEARCHABLE_IN_LOOKUP = 2


class GdxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"gdx status {status}: {message}")
        self.status = status


class HitStruct(C.Structure):
    _fields_ = [("text_id", C.c_uint64), ("position", C.c_uint64)]


class Hits32(C.Structure):
    """gdx_hits32_t: narrow results in pinned memory the library owns (gdx_locate_many_alloc_layout32)"""
    _fields_ = [("hit_offsets", C.POINTER(C.c_uint32)), ("hits", C.POINTER(C.c_uint32)), ("total_hits", C.c_uint64),
                ("nq", C.c_uint64), ("reserved", C.c_uint64 * 2)]


class IndexInfo(C.Structure):
    _fields_ = [("total_text_len", C.c_uint64), ("num_texts", C.c_uint64), ("sigma", C.c_int32),
                ("n_searchable", C.c_int32), ("lookup_depth", C.c_int32), ("index_width", C.c_int32),
                ("sa_rate", C.c_uint64), ("device_bytes", C.c_uint64), ("device_id", C.c_int32),
                ("table_layout", C.c_int32)]


class BuildOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("pair_lines", C.c_int32), ("jump_entry_bytes", C.c_int32),
                ("top_table_depth", C.c_int32), ("aux_budget_bytes", C.c_uint64), ("full_suffix_array", C.c_int32),
                ("text_units", C.c_int32), ("seed_symbols", C.c_int32), ("seed_load_percent", C.c_int32),
                ("inverse_suffix_array", C.c_int32), ("reference_table_layout", C.c_int32)]


class QueryOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("search_kernel", C.c_int32), ("search_lanes", C.c_int32),
                ("load_policy", C.c_int32), ("length_schedule", C.c_int32), ("locate_kernel", C.c_int32),
                ("locate_jump_walk", C.c_int32), ("search_defer_after", C.c_int32), ("search_fast", C.c_int32),
                ("search_exact", C.c_int32), ("max_hits_per_query", C.c_uint32), ("search_seed", C.c_int32)]


class QueryLayout(C.Structure):
    """gdx_query_layout_t"""
    _fields_ = [("struct_size", C.c_uint32), ("packed", C.c_int32), ("uniform_len", C.c_uint64)]


class DeviceShard(C.Structure):
    _fields_ = [("d_qbuf", C.c_void_p), ("d_qoff", C.c_void_p), ("nq", C.c_uint64)]


class Gathered(C.Structure):
    _fields_ = [("d_counts", C.c_void_p), ("d_hit_offsets", C.c_void_p), ("d_hits", C.c_void_p), ("d_status", C.c_void_p),
                ("nq", C.c_uint64), ("total_hits", C.c_uint64), ("device_id", C.c_int32), ("used_rccl", C.c_int32)]


class IndexAux(C.Structure):
    _fields_ = [("pair_lines", C.c_int32), ("jump_entry_bytes", C.c_int32), ("top_table_depth", C.c_int32),
                ("wanted_jump_entry_bytes", C.c_int32), ("wanted_top_table_depth", C.c_int32),
                ("wide_permille", C.c_int32), ("aux_bytes", C.c_uint64), ("aux_budget_bytes", C.c_uint64)]


class BuildStats(C.Structure):
    _fields_ = [("sa_initial_order", C.c_uint64), ("sa_pending_after_sort", C.c_uint64), ("sa_rounds", C.c_uint64),
                ("seconds_encode", C.c_double), ("seconds_sa", C.c_double), ("seconds_bwt", C.c_double),
                ("seconds_table", C.c_double), ("seconds_lookup", C.c_double), ("seconds_pairs", C.c_double)]


# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "gdx_last_error": [],
    "gdx_device_count": [],
    "gdx_index_build": [u8p, u64p, C.c_uint64, u8p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                        C.POINTER(vp)],
    "gdx_index_build_dev": [vp, u64p, C.c_uint64, u8p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                            C.POINTER(vp)],
    "gdx_index_from_parts": [u64p, u64p, C.c_uint64, u32p, C.c_uint64, u64p, u64p, u64p, C.c_uint64, u8p, C.c_int,
                             C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
    "gdx_index_from_parts_ex": [C.c_int, C.c_int, u64p, u64p, C.c_uint64, u32p, C.c_uint64, u64p, u64p, u64p,
                                C.c_uint64, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
    "gdx_build_options_init": [C.POINTER(BuildOptions)],
    "gdx_query_options_init": [C.POINTER(QueryOptions)],
    "gdx_index_build_ex": [u8p, u64p, C.c_uint64, u8p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                           C.POINTER(BuildOptions), C.POINTER(vp)],
    "gdx_index_build_dev_ex": [vp, u64p, C.c_uint64, u8p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                               C.POINTER(BuildOptions), C.POINTER(vp)],
    "gdx_index_from_parts_ex2": [C.c_int, C.c_int, u64p, u64p, C.c_uint64, u32p, C.c_uint64, u64p, u64p, u64p,
                                 C.c_uint64, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(BuildOptions), C.POINTER(vp)],
    "gdx_index_load_ex": [C.c_char_p, C.c_int, C.POINTER(BuildOptions), C.POINTER(vp)],
    "gdx_index_aux": [vp, C.POINTER(IndexAux)],
    "gdx_index_seed_info": [vp, C.POINTER(C.c_uint64)],
    "gdx_index_seed_records": [vp, C.POINTER(C.c_uint64)],
    "gdx_index_set_query_options": [vp, C.POINTER(QueryOptions)],
    "gdx_index_get_query_options": [vp, C.POINTER(QueryOptions)],
    "gdx_index_rebuild_aux": [vp, C.POINTER(BuildOptions)],
    "gdx_index_save": [vp, C.c_char_p],
    "gdx_index_load": [C.c_char_p, C.c_int, C.POINTER(vp)],
    "gdx_index_free": [vp],
    "gdx_index_info": [vp, C.POINTER(IndexInfo)],
    "gdx_index_export_count": [vp, u64p],
    "gdx_index_export_bwt": [vp, u8p],
    "gdx_index_export_sa_samples": [vp, u32p],
    "gdx_index_export_borders": [vp, u64p, u64p],
    "gdx_index_export_sentinel_indices": [vp, u64p],
    "gdx_index_export_lookup_table": [vp, C.c_int, u32p],
    "gdx_index_export_condensed_table": [vp, u64p, u16p, u32p],
    "gdx_index_export_reference_table": [vp, u64p, C.c_uint64, u64p, u32p, C.c_uint64, u64p],
    "gdx_rank_many": [vp, u8p, u64p, C.c_uint64, u64p],
    "gdx_symbol_at_many": [vp, u64p, C.c_uint64, u8p],
    "gdx_count_many": [vp, u8p, u64p, C.c_uint64, u64p, u8p],
    "gdx_cursors_for_many_queries": [vp, u8p, u64p, C.c_uint64, u64p, u64p, u8p],
    "gdx_locate_many": [vp, u8p, u64p, C.c_uint64, u64p, C.POINTER(HitStruct), C.c_uint64, u64p, u8p],
    "gdx_locate_many_alloc": [vp, u8p, u64p, C.c_uint64, u64p, C.POINTER(C.POINTER(HitStruct)), u64p, u8p],
    "gdx_free_hits": [C.POINTER(HitStruct)],
    "gdx_locate_many_alloc_layout32": [vp, u8p, u64p, C.c_uint64, C.POINTER(QueryLayout), C.POINTER(Hits32), u8p],
    "gdx_free_hits32": [C.POINTER(Hits32)],
    "gdx_release_cached_hits": [],
    "gdx_multi_build": [u8p, u64p, C.c_uint64, u8p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                        C.POINTER(C.c_int), C.c_int, C.POINTER(BuildOptions), C.POINTER(vp)],
    "gdx_multi_from_indexes": [C.POINTER(vp), C.c_int, C.POINTER(vp)],
    "gdx_multi_free": [vp],
    "gdx_debug_force_wide": [C.c_int],
    "gdx_parts_build": [vp, C.c_int, u64p, C.c_uint64, u8p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_uint64,
                        C.POINTER(BuildOptions), C.POINTER(vp)],
    "gdx_parts_free": [vp],
    "gdx_parts_info": [vp, u64p],
    "gdx_parts_set_query_options": [vp, C.POINTER(QueryOptions)],
    "gdx_parts_count_many": [vp, u8p, u64p, C.c_uint64, u64p, u8p],
    "gdx_parts_locate_many_alloc": [vp, u8p, u64p, C.c_uint64, u64p, C.POINTER(C.POINTER(HitStruct)), u64p, u8p],
    "gdx_multi_locate_many_gather_dev": [vp, vp, C.c_int, C.c_int, vp],
    "gdx_multi_set_query_options": [vp, C.POINTER(QueryOptions)],
    "gdx_multi_replicas": [vp],
    "gdx_multi_count_many": [vp, u8p, u64p, C.c_uint64, u64p, u8p],
    "gdx_multi_cursors_for_many_queries": [vp, u8p, u64p, C.c_uint64, u64p, u64p, u8p],
    "gdx_multi_locate_many_alloc": [vp, u8p, u64p, C.c_uint64, u64p, C.POINTER(C.POINTER(HitStruct)), u64p, u8p],
    "gdx_cursor_empty": [vp, u64p, u64p],
    "gdx_cursor_extend_front_many": [vp, u64p, u64p, u8p, C.c_uint64, u8p],
    "gdx_cursor_locate_many": [vp, u64p, u64p, C.c_uint64, u64p, C.POINTER(HitStruct), C.c_uint64, u64p],
    "gdx_cursors_for_many_queries_dev": [vp, vp, vp, C.c_uint64, vp, vp, vp, vp],
    "gdx_count_many_dev": [vp, vp, vp, C.c_uint64, vp, vp, vp],
    "gdx_cursor_extend_front_many_dev": [vp, vp, vp, vp, C.c_uint64, vp, vp],
    "gdx_hit_offsets_dev": [vp, vp, vp, C.c_uint64, vp, vp],
    "gdx_locate_workspace_bytes": [C.c_uint64],
    "gdx_locate_intervals_dev": [vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp],
    "gdx_locate_intervals_hint_dev": [vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, vp],
    "gdx_cursors_for_many_queries_hint_dev": [vp, vp, vp, C.c_uint64, vp, vp, vp, vp, vp],
    "gdx_rank_many_dev": [vp, vp, vp, C.c_uint64, vp, vp, vp],
    "gdx_locate_many_search_dev": [vp, vp, vp, C.c_uint64, vp, vp],
    "gdx_locate_many_offsets_dev": [vp, vp, C.c_uint64, vp, vp],
    "gdx_locate_many_offsets_capped_dev": [vp, vp, C.c_uint64, C.c_uint32, vp, vp],
    "gdx_locate_many_hits_dev": [vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp],
    "gdx_locate_many_unpack_dev": [vp, vp, C.c_uint64, vp, vp, vp],
    "gdx_locate_many_search_compact_dev": [vp, vp, vp, C.c_uint64, vp, vp, vp],
    "gdx_locate_many_offsets_compact_dev": [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp],
    "gdx_locate_many_hits_compact_dev": [vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp],
    "gdx_locate_many_unpack_compact_dev": [vp, vp, vp, C.c_uint64, vp, vp, vp],
    "gdx_compact_split_hits_dev": [vp, vp, C.c_uint64, vp, vp, vp],
    "gdx_compact_exceptions_dev": [vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp],
    "gdx_wire_bitmap_bytes": [C.c_uint64],
    "gdx_wire_pack_workspace_bytes": [C.c_uint64],
    "gdx_wire_pack_dev": [vp, vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp, vp],
    "gdx_wire_split_dev": [vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp, vp, C.c_uint64, vp, vp, vp],
    "gdx_locate_many_totals_workspace_bytes": [C.c_uint64],
    "gdx_locate_many_totals_compact_dev": [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp],
    "gdx_locate_many_offsets_hits_compact_dev": [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, C.c_uint64, C.c_uint64, vp, vp, vp],
    "gdx_locate_many_offsets32_hits_compact_dev": [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, C.c_uint64, C.c_uint64, vp, vp, vp],
    "gdx_packed_bytes": [C.c_uint64],
    "gdx_pack_queries": [vp, u8p, u64p, C.c_uint64, u8p, u64p, C.c_uint64, u64p],
    "gdx_pack_queries_dev": [vp, vp, C.c_uint64, vp, vp, vp, vp],
    "gdx_pack_queries_table": [u8p, u8p, u64p, C.c_uint64, u8p, u64p, C.c_uint64, u64p],
    "gdx_count_many_packed": [vp, u8p, u64p, C.c_uint64, u64p, u8p],
    "gdx_cursors_for_many_queries_packed": [vp, u8p, u64p, C.c_uint64, u64p, u64p, u8p],
    "gdx_cursors_for_many_queries_packed_dev": [vp, vp, vp, C.c_uint64, vp, vp, vp, vp],
    "gdx_count_many_packed_dev": [vp, vp, vp, C.c_uint64, vp, vp, vp],
    "gdx_locate_many_search_packed_dev": [vp, vp, vp, C.c_uint64, vp, vp],
    "gdx_query_layout_init": [C.POINTER(QueryLayout)],
    "gdx_count_many_layout": [vp, u8p, u64p, C.c_uint64, C.POINTER(QueryLayout), u64p, u8p],
    "gdx_cursors_for_many_queries_layout": [vp, u8p, u64p, C.c_uint64, C.POINTER(QueryLayout), u64p, u64p, u8p],
    "gdx_locate_many_alloc_layout": [vp, u8p, u64p, C.c_uint64, C.POINTER(QueryLayout), u64p, C.POINTER(C.POINTER(HitStruct)),
                                     u64p, u8p],
    "gdx_locate_many_search_compact_layout_dev": [vp, vp, vp, C.c_uint64, C.POINTER(QueryLayout), vp, vp, vp],
    "gdx_locate_many_search_layout_dev": [vp, vp, vp, C.c_uint64, C.POINTER(QueryLayout), vp, vp],
    "gdx_locate_many_search_totals_compact_layout_dev": [vp, vp, vp, C.c_uint64, C.POINTER(QueryLayout), C.c_uint32, vp, vp, vp, vp, vp],
    "gdx_locate_many_step_compact_layout_dev": [vp, vp, vp, C.c_uint64, C.POINTER(QueryLayout), C.c_uint32, vp, vp, vp, vp, vp,
                                                C.c_uint32, vp, C.c_uint64, vp, vp, vp],
    "gdx_count_many_layout_dev": [vp, vp, vp, C.c_uint64, C.POINTER(QueryLayout), vp, vp, vp],
    "gdx_cursors_for_many_queries_layout_dev": [vp, vp, vp, C.c_uint64, C.POINTER(QueryLayout), vp, vp, vp, vp],
    "gdx_cursor_extend_front_strings_dev": [vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp, vp, vp, vp],
    "gdx_cursor_extend_front_chunk_dev": [vp, vp, vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp],
    "gdx_cursor_extend_front_strings": [vp, u64p, u64p, u8p, u64p, C.c_uint64, u8p],
    # gdx_bench.h
    "gdx_index_build_stats": [vp, C.POINTER(BuildStats)],
    "gdx_synth_text_dev": [vp, C.c_uint64, C.c_uint64, C.c_uint32, vp],
    "gdx_synth_queries_dev": [vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp,
                              C.c_uint64, u64p, vp],
    "gdx_debug_set_search_variant": [C.c_int],
    "gdx_debug_set_host_chunking": [C.c_uint64, C.c_uint64],
    "gdx_bench_stream_copy": [vp, vp, C.c_uint64, vp],
    "gdx_bench_stream_read": [vp, C.c_uint64, vp, vp],
    "gdx_bench_random_gather": [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp],
    "gdx_search_step_stats_dev": [vp, vp, vp, C.c_uint64, vp, vp],
    "gdx_index_aux_info": [vp, C.POINTER(C.c_uint32)],
    "gdx_bench_lf_walk_dev": [vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp],
    "gdx_fastx_open": [C.c_char_p, C.POINTER(vp)],
    "gdx_fastx_next_batch": [vp, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64)],
    "gdx_fastx_next_batch_ex": [vp, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "gdx_fastx_close": [vp],
    "gdx_locate_step_stats_dev": [vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, vp],
    "gdx_locate_many_hits_stats_dev": [vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, vp],
}
_RESTYPES = {"gdx_locate_many_totals_workspace_bytes": C.c_uint64, "gdx_wire_bitmap_bytes": C.c_uint64,
             "gdx_wire_pack_workspace_bytes": C.c_uint64, "gdx_last_error": C.c_char_p, "gdx_index_free": None, "gdx_fastx_close": None,
             "gdx_build_options_init": None, "gdx_query_layout_init": None, "gdx_query_options_init": None, "gdx_free_hits": None, "gdx_free_hits32": None, "gdx_release_cached_hits": None, "gdx_multi_free": None, "gdx_parts_free": None,
             "gdx_locate_workspace_bytes": C.c_uint64, "gdx_packed_bytes": C.c_uint64}

_lib = None


def _share_hip_runtime_with_torch():
    """A process must hold ONE HIP runtime.  torch wheels bundle their own libamdhip64.so.7; whichever copy is
    loaded first serves every later DT_NEEDED of that soname.  If libgdx.so pulled in /opt/rocm's copy first, a
    later `import torch` would find "No HIP GPUs".  So, when torch is installed (not necessarily imported),
    its copy is loaded first, and libgdx.so and torch share it whatever the import order."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Loads libgdx.so; raises if it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"or `make -C genedex_amd/csrc`. genedex_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export what gdx.h declares
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def last_error() -> str:
    return load().gdx_last_error().decode(errors="replace")


def check(status: int, allow=()):
    if status != GDX_OK and status not in allow:
        raise GdxError(status, last_error())
    return status
