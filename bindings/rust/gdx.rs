//! Rust binding of libgdx.so (include/gdx.h) with safe wrappers named like genedex's API.
//!
//! NOT compiled in this repository's image (no rustc/cargo); kept as the reference-side stub a
//! genedex maintainer would add, e.g. as `src/gpu.rs` behind a `gpu` cargo feature, linking with
//! `cargo:rustc-link-lib=dylib=gdx`.  It is a parallel type (`GpuFmIndex`), not `FmIndex<I, R>` itself:
//! the reference's `FmIndex` is generic over a sealed operator trait whose methods are called one
//! rank at a time (text_with_rank_support/mod.rs:88-133); a GPU cannot be driven at that granularity,
//! so the drop-in point is the batched public API (`count_many`, `locate_many`,
//! `cursors_for_many_queries`, lib.rs:155-246), whose signatures are mirrored here: any
//! `IntoIterator<Item: AsRef<[u8]>>` in, lazy iterators out (the results are computed by one call and
//! then handed out lazily).
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct gdx_fastx_t {
    _private: [u8; 0],
}
#[repr(C)]
pub struct gdx_index_t {
    _private: [u8; 0],
}
#[repr(C)]
pub struct gdx_multi_t {
    _private: [u8; 0],
}
#[repr(C)]
pub struct gdx_parts_t {
    _private: [u8; 0],
}

/// lib.rs:331-335
#[repr(C)]
#[derive(Debug, Clone, Copy, PartialEq, Eq, PartialOrd, Ord, Hash)]
pub struct Hit {
    pub text_id: u64,
    pub position: u64,
}

/// gdx_hit32_t: a hit as two u32 (what the device writes; Hit widened on demand)
#[repr(C)]
#[derive(Debug, Clone, Copy, PartialEq, Eq, PartialOrd, Ord, Hash)]
pub struct Hit32 {
    pub text_id: u32,
    pub position: u32,
}

/// gdx_hits32_t: narrow results in pinned memory of the library's (gdx_locate_many_alloc_layout32)
#[repr(C)]
pub struct Hits32Raw {
    pub hit_offsets: *mut u32,
    pub hits: *mut Hit32,
    pub total_hits: u64,
    pub nq: u64,
    pub reserved: [u64; 2],
}

/// gdx_build_options_t: which derived acceleration structures an index carries (include/gdx.h).
/// `BuildOptions::default()` -- every field at its default -- builds the library's DEFAULT SHAPE (round 6): seed table + text
/// units + full and inverse suffix array + pair lines + top table, no jump table (104 GB at hg38 scale), the one index that
/// serves count / locate, exact intervals, cursors and reads from repeats at their best measured speeds; setting any of
/// jump_entry_bytes / full_suffix_array / text_units / seed_symbols / inverse_suffix_array makes the fields mean what they say.
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct BuildOptions {
    pub struct_size: u32,
    pub pair_lines: i32,       // -1 default, 0 off, 1 on
    pub jump_entry_bytes: i32, // -1 default (none in the default shape, else 32), 0, 8, 16, 32
    pub top_table_depth: i32,  // -1 default, 0 none, 1..=16
    pub aux_budget_bytes: u64, // 0 = default
    pub full_suffix_array: i32, // -1 / 0 off, 1: SA[row] of every row as its own array
    pub text_units: i32,        // -1 / 0 off, 1: the text at 4 bits per symbol (count / locate compare with it)
    pub seed_symbols: i32,      // -1 / 0 off, 1: seed table with k from the text length, 8..=24: that k
    pub seed_load_percent: i32, // 0 = default (60 in the default shape, else 70)
    pub inverse_suffix_array: i32, // -1 / 0 off, 1: ISA as its own array (exact intervals through the seed table)
    pub reference_table_layout: i32, // -1 / 0 this library's table, 1..=4: Condensed64 / Condensed512 / Flat64 / Flat512 as genedex builds them
}
impl Default for BuildOptions {
    fn default() -> Self {
        let mut o = std::mem::MaybeUninit::<BuildOptions>::uninit();
        unsafe {
            gdx_build_options_init(o.as_mut_ptr());
            o.assume_init()
        }
    }
}

/// gdx_query_options_t: kernel variants of the query calls on a handle (results never depend on them).
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct QueryOptions {
    pub struct_size: u32,
    pub search_kernel: i32,
    pub search_lanes: i32,
    pub load_policy: i32,
    pub length_schedule: i32,
    pub locate_kernel: i32,
    pub locate_jump_walk: i32,
    pub search_defer_after: i32,
    pub search_fast: i32,
    pub search_exact: i32,
    /// host-pointer locate calls return at most this many hits per query, the first ones in suffix-array order:
    /// `locate(q).take(k)` on the reference's lazy iterator (lib.rs:187-197); 0 = all
    pub max_hits_per_query: u32,
    pub search_seed: i32, // -1 default (on when the index has a seed table), 0 off
}

/// gdx_device_shard_t / gdx_gathered_t: gdx_multi_locate_many_gather_dev (device-resident shards, results gathered on the
/// root replica's GPU with ncclSend / ncclRecv)
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct DeviceShard {
    pub d_qbuf: *const c_void,
    pub d_qoff: *const c_void,
    pub nq: u64,
}
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct Gathered {
    pub d_counts: *mut c_void,
    pub d_hit_offsets: *mut c_void,
    pub d_hits: *mut c_void,
    pub d_status: *mut c_void,
    pub nq: u64,
    pub total_hits: u64,
    pub device_id: i32,
    pub used_rccl: i32,
}

pub const GDX_OK: c_int = 0;
pub const GDX_ERR_CAPACITY: c_int = 5;
pub const GDX_ERR_QUERY_STATUS: c_int = 6;

/// gdx_query_layout_t
#[repr(C)]
pub struct QueryLayout {
    pub struct_size: u32,
    pub packed: i32,      // 0: IO symbols, 1: 2-bit codes (A C G T of the DNA alphabets)
    pub uniform_len: u64, // 0: offsets array; L: every read has L symbols, no offsets
}

extern "C" {
    pub fn gdx_last_error() -> *const c_char;
    pub fn gdx_build_options_init(opts: *mut BuildOptions);
    pub fn gdx_query_options_init(opts: *mut QueryOptions);
    pub fn gdx_index_build_ex(
        texts_buf: *const u8, text_offsets: *const u64, n_texts: u64, io_to_dense: *const u8,
        sigma: c_int, n_searchable: c_int, sa_rate: u64, lookup_depth: c_int, index_width: c_int,
        device_id: c_int, opts: *const BuildOptions, out: *mut *mut gdx_index_t,
    ) -> c_int;
    /// table_kind 0 = condensed, 1 = flat; block_bits 64 | 512 (FmIndexCondensed64/512, FmIndexFlat64/512)
    pub fn gdx_index_from_parts_ex2(
        table_kind: c_int, block_bits: c_int, count: *const u64, interleaved_blocks: *const u64, n: u64,
        sa_samples: *const u32, sa_rate: u64, border_keys: *const u64, border_vals: *const u64,
        sentinel_indices: *const u64, n_texts: u64, io_to_dense: *const u8, sigma: c_int, n_searchable: c_int,
        lookup_depth: c_int, index_width: c_int, device_id: c_int, opts: *const BuildOptions,
        out: *mut *mut gdx_index_t,
    ) -> c_int;
    pub fn gdx_index_save(ix: *const gdx_index_t, path: *const c_char) -> c_int;
    pub fn gdx_index_load_ex(
        path: *const c_char, device_id: c_int, opts: *const BuildOptions, out: *mut *mut gdx_index_t,
    ) -> c_int;
    pub fn gdx_index_free(ix: *mut gdx_index_t);
    pub fn gdx_index_set_query_options(ix: *mut gdx_index_t, opts: *const QueryOptions) -> c_int;
    pub fn gdx_count_many(
        ix: *const gdx_index_t, qbuf: *const u8, qoff: *const u64, nq: u64, out_counts: *mut u64,
        out_status: *mut u8,
    ) -> c_int;
    pub fn gdx_cursors_for_many_queries(
        ix: *const gdx_index_t, qbuf: *const u8, qoff: *const u64, nq: u64, out_start: *mut u64,
        out_end: *mut u64, out_status: *mut u8,
    ) -> c_int;
    /// one pass, library-allocated hit array (release with gdx_free_hits)
    pub fn gdx_locate_many_alloc(
        ix: *const gdx_index_t, qbuf: *const u8, qoff: *const u64, nq: u64, out_hit_offsets: *mut u64,
        out_hits: *mut *mut Hit, out_total: *mut u64, out_status: *mut u8,
    ) -> c_int;
    pub fn gdx_free_hits(hits: *mut Hit);
    /// query layouts (include/gdx.h): packed = 2-bit codes, uniform_len = L: read i is symbols [i L, (i + 1) L), qoff may be null
    pub fn gdx_query_layout_init(layout: *mut QueryLayout);
    pub fn gdx_packed_bytes(n_symbols: u64) -> u64;
    pub fn gdx_pack_queries(
        ix: *const gdx_index_t, qbuf: *const u8, qoff: *const u64, nq: u64, out_packed: *mut u8,
        out_exceptions: *mut u64, exceptions_capacity: u64, out_n_exceptions: *mut u64,
    ) -> c_int;
    pub fn gdx_count_many_layout(
        ix: *const gdx_index_t, qbuf: *const u8, qoff: *const u64, nq: u64, layout: *const QueryLayout,
        out_counts: *mut u64, out_status: *mut u8,
    ) -> c_int;
    pub fn gdx_locate_many_alloc_layout(
        ix: *const gdx_index_t, qbuf: *const u8, qoff: *const u64, nq: u64, layout: *const QueryLayout,
        out_hit_offsets: *mut u64, out_hits: *mut *mut Hit, out_total: *mut u64, out_status: *mut u8,
    ) -> c_int;
    /// narrow results (u32 offsets, 8-byte hits) in pinned memory the library owns; release with gdx_free_hits32
    pub fn gdx_locate_many_alloc_layout32(
        ix: *const gdx_index_t, qbuf: *const u8, qoff: *const u64, nq: u64, layout: *const QueryLayout,
        out_results: *mut Hits32Raw, out_status: *mut u8,
    ) -> c_int;
    pub fn gdx_free_hits32(results: *mut Hits32Raw);
    pub fn gdx_release_cached_hits();
    pub fn gdx_cursor_empty(ix: *const gdx_index_t, start: *mut u64, end: *mut u64) -> c_int;
    pub fn gdx_cursor_extend_front_many(
        ix: *const gdx_index_t, start: *mut u64, end: *mut u64, io_symbols: *const u8, m: u64,
        out_status: *mut u8,
    ) -> c_int;
    /// every cursor extended by a whole string (right to left) in one launch
    pub fn gdx_cursor_extend_front_strings(
        ix: *const gdx_index_t, start: *mut u64, end: *mut u64, qbuf: *const u8, qoff: *const u64, m: u64,
        status: *mut u8,
    ) -> c_int;
    pub fn gdx_cursor_locate_many(
        ix: *const gdx_index_t, start: *const u64, end: *const u64, m: u64, out_hit_offsets: *mut u64,
        hits: *mut Hit, hits_capacity: u64, out_total: *mut u64,
    ) -> c_int;
    pub fn gdx_rank_many(
        ix: *const gdx_index_t, symbols: *const u8, idx: *const u64, m: u64, out: *mut u64,
    ) -> c_int;
    // device-resident variants take *const c_void device pointers and a hipStream_t.  The step calls that follow are declared in
    // include/gdx_experimental.h since round 6 (the core call is gdx_locate_many_step_compact_layout_dev: the whole count + locate
    // step in one call); they stay exported.  The fused count + locate in steps:
    pub fn gdx_locate_many_search_dev(
        ix: *const gdx_index_t, d_qbuf: *const c_void, d_qoff: *const c_void, nq: u64, d_records: *mut c_void,
        stream: *mut c_void,
    ) -> c_int;
    pub fn gdx_locate_many_offsets_dev(
        ix: *const gdx_index_t, d_records: *const c_void, nq: u64, d_hit_offsets: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    pub fn gdx_locate_workspace_bytes(total_hits: u64) -> u64;
    pub fn gdx_locate_many_hits_dev(
        ix: *const gdx_index_t, d_records: *const c_void, nq: u64, d_hit_offsets: *const c_void, total_hits: u64,
        d_hits: *mut c_void, d_workspace: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    // the same with compact results beside the records (u32 per query: position of the only hit, !0 = none, !1 = see the
    // record): on an index with a seed table scan and locate stream 4 bytes per query instead of 16
    pub fn gdx_locate_many_search_compact_dev(
        ix: *const gdx_index_t, d_qbuf: *const c_void, d_qoff: *const c_void, nq: u64, d_records: *mut c_void,
        d_compact: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    pub fn gdx_locate_many_offsets_compact_dev(
        ix: *const gdx_index_t, d_records: *const c_void, d_compact: *const c_void, nq: u64, max_hits: u32,
        d_hit_offsets: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    pub fn gdx_locate_many_hits_compact_dev(
        ix: *const gdx_index_t, d_records: *const c_void, d_compact: *const c_void, nq: u64, d_hit_offsets: *const c_void,
        total_hits: u64, d_hits: *mut c_void, d_workspace: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    // offsets and hits around the one host round trip: totals (u64[2] on the device: all hit slots, slots behind "see the
    // record") -> read back, size d_hits -> offsets + the compactly answered hits in one pass, then the rest
    pub fn gdx_locate_many_totals_workspace_bytes(nq: u64) -> u64;
    pub fn gdx_locate_many_totals_compact_dev(
        ix: *const gdx_index_t, d_records: *const c_void, d_compact: *const c_void, nq: u64, max_hits: u32,
        d_scan_workspace: *mut c_void, d_totals: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    pub fn gdx_locate_many_offsets_hits_compact_dev(
        ix: *const gdx_index_t, d_records: *const c_void, d_compact: *const c_void, nq: u64, max_hits: u32,
        d_scan_workspace: *const c_void, d_hit_offsets: *mut c_void, total_hits: u64, rest_hits: u64, d_hits: *mut c_void,
        d_workspace: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    /// the whole count + locate step of a device-resident batch in one call, no host round trip (gdx.h)
    pub fn gdx_locate_many_step_compact_layout_dev(
        ix: *const gdx_index_t, d_qbuf: *const c_void, d_qoff: *const c_void, nq: u64, layout: *const QueryLayout,
        max_hits: u32, d_records: *mut c_void, d_compact: *mut c_void, d_scan_workspace: *mut c_void, d_totals: *mut c_void,
        d_hit_offsets: *mut c_void, offsets_width: u32, d_hits: *mut c_void, hits_capacity: u64, d_workspace: *mut c_void,
        event_after_search: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    /// a located shard on its way to another device: a bit per read + the found reads' positions + the exceptions (gdx.h)
    pub fn gdx_wire_bitmap_bytes(nq: u64) -> u64;
    pub fn gdx_wire_pack_workspace_bytes(nq: u64) -> u64;
    pub fn gdx_wire_pack_dev(
        ix: *const gdx_index_t, d_compact: *const c_void, d_hit_offsets: *const c_void, offsets_width: u32,
        d_hits: *const c_void, nq: u64, d_bitmap: *mut c_void, d_tile_found: *mut c_void, d_found_pos: *mut c_void,
        found_capacity: u64, d_exc_queries: *mut c_void, d_exc_counts: *mut c_void, exc_capacity: u64,
        d_exc_text_ids: *mut c_void, d_exc_positions: *mut c_void, exc_hits_capacity: u64, d_meta: *mut c_void,
        d_workspace: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    pub fn gdx_wire_split_dev(
        ix: *const gdx_index_t, d_bitmap: *const c_void, d_tile_found: *const c_void, d_found_pos: *const c_void,
        found_capacity: u64, nq: u64, d_exc_queries: *const c_void, d_meta: *const c_void, exc_capacity: u64,
        d_out_text_ids: *mut c_void, d_out_positions: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    pub fn gdx_index_seed_info(ix: *const gdx_index_t, out: *mut u64) -> c_int;
    /// out[4]: records of two-copy repeats, of three- and four-copy repeats, their bytes, 0
    pub fn gdx_index_seed_records(ix: *const gdx_index_t, out: *mut u64) -> c_int;
    /// an index built with `reference_table_layout`: genedex's own interleaved blocks and superblock offsets as they sit in HBM
    pub fn gdx_index_export_reference_table(
        ix: *const gdx_index_t, interleaved_blocks: *mut u64, capacity_words: u64, out_n_words: *mut u64,
        interleaved_superblock_offsets: *mut u32, capacity_offsets: u64, out_n_offsets: *mut u64,
    ) -> c_int;
    pub fn gdx_locate_many_unpack_compact_dev(
        ix: *const gdx_index_t, d_records: *const c_void, d_compact: *const c_void, nq: u64, d_out_counts: *mut c_void,
        d_out_status: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    /// compact results as text id bytes + positions in the text (-1 none, -2 see the record): the multi-GPU gather's form
    pub fn gdx_compact_split_hits_dev(
        ix: *const gdx_index_t, d_compact: *const c_void, nq: u64, d_out_text_ids: *mut c_void, d_out_positions: *mut c_void,
        stream: *mut c_void,
    ) -> c_int;
    /// the queries whose compact result says "see the record" (unordered, up to `capacity`); *d_out_n (u64) = all of them
    pub fn gdx_compact_exceptions_dev(
        ix: *const gdx_index_t, d_compact: *const c_void, nq: u64, d_out_queries: *mut c_void, capacity: u64,
        d_out_n: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    // several GPUs of one node behind one handle
    pub fn gdx_multi_build(
        texts_buf: *const u8, text_offsets: *const u64, n_texts: u64, io_to_dense: *const u8, sigma: c_int,
        n_searchable: c_int, sa_rate: u64, lookup_depth: c_int, index_width: c_int, device_ids: *const c_int,
        n_devices: c_int, opts: *const BuildOptions, out: *mut *mut gdx_multi_t,
    ) -> c_int;
    pub fn gdx_multi_free(m: *mut gdx_multi_t);
    // collections beyond 2^32 - 1 symbols as several 32-bit indexes cut at text borders (count / locate); a single text
    // beyond that, or cursors over the whole collection: build with index_width 64 (the 64-bit engine) instead
    pub fn gdx_parts_build(
        texts_buf: *const c_void, texts_on_device: c_int, text_offsets: *const u64, n_texts: u64, io_to_dense: *const u8,
        sigma: c_int, n_searchable: c_int, sa_rate: u64, lookup_depth: c_int, device_id: c_int, max_part_symbols: u64,
        opts: *const BuildOptions, out: *mut *mut gdx_parts_t,
    ) -> c_int;
    pub fn gdx_parts_free(p: *mut gdx_parts_t);
    pub fn gdx_parts_info(p: *const gdx_parts_t, out: *mut u64) -> c_int;
    pub fn gdx_parts_set_query_options(p: *mut gdx_parts_t, opts: *const QueryOptions) -> c_int;
    pub fn gdx_parts_count_many(
        p: *const gdx_parts_t, qbuf: *const u8, qoff: *const u64, nq: u64, out_counts: *mut u64, out_status: *mut u8,
    ) -> c_int;
    pub fn gdx_parts_locate_many_alloc(
        p: *const gdx_parts_t, qbuf: *const u8, qoff: *const u64, nq: u64, out_hit_offsets: *mut u64,
        out_hits: *mut *mut Hit, out_total: *mut u64, out_status: *mut u8,
    ) -> c_int;
    pub fn gdx_multi_set_query_options(m: *mut gdx_multi_t, opts: *const QueryOptions) -> c_int;
    pub fn gdx_multi_locate_many_gather_dev(
        m: *mut gdx_multi_t, shards: *const DeviceShard, n_shards: c_int, root: c_int, out: *mut Gathered,
    ) -> c_int;
    pub fn gdx_multi_count_many(
        m: *const gdx_multi_t, qbuf: *const u8, qoff: *const u64, nq: u64, out_counts: *mut u64, out_status: *mut u8,
    ) -> c_int;
    pub fn gdx_multi_locate_many_alloc(
        m: *const gdx_multi_t, qbuf: *const u8, qoff: *const u64, nq: u64, out_hit_offsets: *mut u64,
        out_hits: *mut *mut Hit, out_total: *mut u64, out_status: *mut u8,
    ) -> c_int;
    // FASTA / FASTQ ingestion into (qbuf, qoff) batches (host only)
    pub fn gdx_fastx_open(path: *const c_char, out: *mut *mut gdx_fastx_t) -> c_int;
    pub fn gdx_fastx_next_batch(
        reader: *mut gdx_fastx_t, qbuf: *mut u8, qbuf_capacity: u64, qoff: *mut u64, max_records: u64, n_out: *mut u64,
    ) -> c_int;
    /// the same; *out_uniform_len = the batch's common read length (0 when they differ): gdx_query_layout_t.uniform_len
    pub fn gdx_fastx_next_batch_ex(
        reader: *mut gdx_fastx_t, qbuf: *mut u8, qbuf_capacity: u64, qoff: *mut u64, max_records: u64, n_out: *mut u64,
        out_uniform_len: *mut u64,
    ) -> c_int;
    pub fn gdx_fastx_close(reader: *mut gdx_fastx_t);
}

fn check(rc: c_int) {
    if rc != GDX_OK {
        let msg = unsafe { CStr::from_ptr(gdx_last_error()) }.to_string_lossy().into_owned();
        panic!("gdx: {msg}"); // the reference panics in the same situations
    }
}

fn pack<Q: AsRef<[u8]>>(queries: impl IntoIterator<Item = Q>) -> (Vec<u8>, Vec<u64>) {
    let (mut buf, mut off) = (Vec::new(), vec![0u64]);
    for q in queries {
        buf.extend_from_slice(q.as_ref());
        off.push(buf.len() as u64);
    }
    (buf, off)
}

/// Hits of one call, owned by the library's allocation; handed out per query without copying.
pub struct Hits {
    ptr: *mut Hit,
    total: usize,
    offsets: Vec<u64>,
}
impl Drop for Hits {
    fn drop(&mut self) {
        unsafe { gdx_free_hits(self.ptr) }
    }
}
impl Hits {
    /// hits of query i, in suffix-array order (lib.rs:187-197)
    pub fn of(&self, i: usize) -> &[Hit] {
        let (a, b) = (self.offsets[i] as usize, self.offsets[i + 1] as usize);
        if self.total == 0 { &[] } else { unsafe { std::slice::from_raw_parts(self.ptr.add(a), b - a) } }
    }
    /// the shape of `FmIndex::locate_many`: an iterator over queries of iterators over hits
    pub fn iter(&self) -> impl Iterator<Item = impl Iterator<Item = Hit> + '_> + '_ {
        (0..self.offsets.len() - 1).map(move |i| self.of(i).iter().copied())
    }
}

/// The result of `GpuFmIndex::locate_many_packed`: u32 offsets and 8-byte hits where the device wrote them (pinned memory of
/// the library's; given back on drop).  Same shape of access as `Hits`.
pub struct Hits32 {
    raw: Hits32Raw,
}
impl Drop for Hits32 {
    fn drop(&mut self) {
        unsafe { gdx_free_hits32(&mut self.raw) }
    }
}
impl Hits32 {
    pub fn of(&self, i: usize) -> &[Hit32] {
        let off = unsafe { std::slice::from_raw_parts(self.raw.hit_offsets, self.raw.nq as usize + 1) };
        let (a, b) = (off[i] as usize, off[i + 1] as usize);
        if self.raw.total_hits == 0 { &[] } else { unsafe { std::slice::from_raw_parts(self.raw.hits.add(a), b - a) } }
    }
    pub fn iter(&self) -> impl Iterator<Item = impl Iterator<Item = Hit> + '_> + '_ {
        (0..self.raw.nq as usize).map(move |i| {
            self.of(i).iter().map(|h| Hit { text_id: h.text_id as u64, position: h.position as u64 })
        })
    }
}

/// Owns an index replica in HBM.  Send + Sync like `FmIndex` (handles are immutable).
pub struct GpuFmIndex {
    raw: *mut gdx_index_t,
}
unsafe impl Send for GpuFmIndex {}
unsafe impl Sync for GpuFmIndex {}

impl Drop for GpuFmIndex {
    fn drop(&mut self) {
        unsafe { gdx_index_free(self.raw) }
    }
}

#[derive(Clone, Copy)]
pub struct GpuCursor<'a> {
    index: &'a GpuFmIndex,
    start: u64,
    end: u64,
}

impl GpuFmIndex {
    /// `FmIndexConfig::<I>::construct_index` (config.rs:63-69); `io_to_dense` is
    /// `Alphabet::io_to_dense_representation_table` (alphabet.rs:25).
    pub fn construct<T: AsRef<[u8]>>(
        texts: impl IntoIterator<Item = T>, io_to_dense: &[u8; 256], sigma: usize, n_searchable: usize,
        sa_rate: usize, lookup_depth: usize, index_width: i32, device: i32, options: &BuildOptions,
    ) -> Self {
        let (buf, off) = pack(texts);
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            gdx_index_build_ex(buf.as_ptr(), off.as_ptr(), off.len() as u64 - 1, io_to_dense.as_ptr(), sigma as c_int,
                               n_searchable as c_int, sa_rate as u64, lookup_depth as c_int, index_width, device,
                               options, &mut raw)
        });
        Self { raw }
    }

    /// lib.rs:155-161
    pub fn count_many<Q: AsRef<[u8]>>(&self, queries: impl IntoIterator<Item = Q>) -> impl Iterator<Item = usize> {
        let (buf, off) = pack(queries);
        let nq = off.len() - 1;
        let mut counts = vec![0u64; nq];
        check(unsafe { gdx_count_many(self.raw, buf.as_ptr(), off.as_ptr(), nq as u64, counts.as_mut_ptr(), std::ptr::null_mut()) });
        counts.into_iter().map(|c| c as usize)
    }

    /// lib.rs:147-149
    pub fn count(&self, query: &[u8]) -> usize {
        self.count_many([query]).next().unwrap()
    }

    /// count_many for a batch of sequencer reads of one length, as a FASTQ reader holds them: `reads` = the reads back to
    /// back (ASCII).  They cross PCIe as 2-bit codes without offsets (gdx_query_layout_t { packed, uniform_len }); reads
    /// with a symbol outside A C G T (the packed form's exceptions) are counted through the plain call.  Same counts as
    /// count_many (lib.rs:155-161).
    pub fn count_reads(&self, reads: &[u8], read_len: usize) -> Vec<usize> {
        assert!(read_len > 0 && reads.len() % read_len == 0);
        let nq = reads.len() / read_len;
        let off: Vec<u64> = (0..=nq).map(|i| (i * read_len) as u64).collect();
        let mut packed = vec![0u8; unsafe { gdx_packed_bytes(reads.len() as u64) } as usize];
        let mut exceptions = vec![0u64; nq.max(1)];
        let mut n_exc = 0u64;
        check(unsafe {
            gdx_pack_queries(self.raw, reads.as_ptr(), off.as_ptr(), nq as u64, packed.as_mut_ptr(), exceptions.as_mut_ptr(),
                             nq as u64, &mut n_exc)
        });
        let mut layout = QueryLayout { struct_size: 0, packed: 0, uniform_len: 0 };
        unsafe { gdx_query_layout_init(&mut layout) };
        layout.packed = 1;
        layout.uniform_len = read_len as u64;
        let mut counts = vec![0u64; nq];
        check(unsafe {
            gdx_count_many_layout(self.raw, packed.as_ptr(), std::ptr::null(), nq as u64, &layout, counts.as_mut_ptr(),
                                  std::ptr::null_mut())
        });
        let mut out: Vec<usize> = counts.into_iter().map(|c| c as usize).collect();
        for &q in &exceptions[..n_exc as usize] {
            let q = q as usize;
            out[q] = self.count(&reads[q * read_len..(q + 1) * read_len]);
        }
        out
    }

    /// lib.rs:179-185, one pass (search, scan, locate pipelined over chunks of the batch)
    pub fn locate_many<Q: AsRef<[u8]>>(&self, queries: impl IntoIterator<Item = Q>) -> Hits {
        let (buf, off) = pack(queries);
        let nq = off.len() - 1;
        let mut offsets = vec![0u64; nq + 1];
        let (mut total, mut ptr) = (0u64, std::ptr::null_mut());
        check(unsafe {
            gdx_locate_many_alloc(self.raw, buf.as_ptr(), off.as_ptr(), nq as u64, offsets.as_mut_ptr(), &mut ptr,
                                  &mut total, std::ptr::null_mut())
        });
        Hits { ptr, total: total as usize, offsets }
    }

    /// lib.rs:179-185 for reads of one length handed over as 2-bit codes (gdx_pack_queries): 12.5 bytes per len-50 read go in,
    /// 12 bytes per result come out, and no host thread copies a result
    pub fn locate_many_packed(&self, packed: &[u8], n_reads: usize, read_len: usize) -> Hits32 {
        let mut lay = QueryLayout { struct_size: std::mem::size_of::<QueryLayout>() as u32, packed: 1, uniform_len: read_len as u64 };
        let mut raw = Hits32Raw { hit_offsets: std::ptr::null_mut(), hits: std::ptr::null_mut(), total_hits: 0, nq: 0, reserved: [0; 2] };
        check(unsafe {
            gdx_locate_many_alloc_layout32(self.raw, packed.as_ptr(), std::ptr::null(), n_reads as u64, &mut lay, &mut raw,
                                           std::ptr::null_mut())
        });
        Hits32 { raw }
    }

    /// lib.rs:169-177
    pub fn locate(&self, query: &[u8]) -> impl Iterator<Item = Hit> {
        self.locate_many([query]).of(0).to_vec().into_iter()
    }

    /// lib.rs:241-246
    pub fn cursors_for_many_queries<'a, Q: AsRef<[u8]>>(
        &'a self, queries: impl IntoIterator<Item = Q>,
    ) -> impl Iterator<Item = GpuCursor<'a>> {
        let (buf, off) = pack(queries);
        let nq = off.len() - 1;
        let (mut s, mut e) = (vec![0u64; nq], vec![0u64; nq]);
        check(unsafe {
            gdx_cursors_for_many_queries(self.raw, buf.as_ptr(), off.as_ptr(), nq as u64, s.as_mut_ptr(), e.as_mut_ptr(),
                                         std::ptr::null_mut())
        });
        s.into_iter().zip(e).map(move |(start, end)| GpuCursor { index: self, start, end })
    }

    /// lib.rs:217-235
    pub fn cursor_for_query(&self, query: &[u8]) -> GpuCursor<'_> {
        self.cursors_for_many_queries([query]).next().unwrap()
    }

    /// lib.rs:202-210
    pub fn cursor_empty(&self) -> GpuCursor<'_> {
        let (mut start, mut end) = (0u64, 0u64);
        check(unsafe { gdx_cursor_empty(self.raw, &mut start, &mut end) });
        GpuCursor { index: self, start, end }
    }

    /// Batched form of the cursor API (ROADMAP.md:33): every cursor extended by its own string, last symbol first,
    /// in ONE launch (up to 32 LF steps per memory fetch), instead of one launch per symbol.
    pub fn extend_cursors_front<'a, Q: AsRef<[u8]>>(
        &'a self, cursors: &mut [GpuCursor<'a>], strings: impl IntoIterator<Item = Q>,
    ) {
        let (buf, off) = pack(strings);
        assert_eq!(off.len() - 1, cursors.len());
        let mut s: Vec<u64> = cursors.iter().map(|c| c.start).collect();
        let mut e: Vec<u64> = cursors.iter().map(|c| c.end).collect();
        check(unsafe {
            gdx_cursor_extend_front_strings(self.raw, s.as_mut_ptr(), e.as_mut_ptr(), buf.as_ptr(), off.as_ptr(),
                                            cursors.len() as u64, std::ptr::null_mut())
        });
        for (c, (start, end)) in cursors.iter_mut().zip(s.into_iter().zip(e)) {
            c.start = start;
            c.end = end;
        }
    }
}

impl<'a> GpuCursor<'a> {
    /// cursor.rs:34-38
    pub fn extend_query_front(&mut self, symbol: u8) {
        check(unsafe {
            gdx_cursor_extend_front_many(self.index.raw, &mut self.start, &mut self.end, &symbol, 1, std::ptr::null_mut())
        });
    }
    /// cursor.rs:61-63
    pub fn count(&self) -> usize {
        (self.end - self.start) as usize
    }
    /// cursor.rs:71-73
    pub fn locate(&self) -> impl Iterator<Item = Hit> {
        let mut offsets = [0u64; 2];
        let mut total = 0u64;
        let mut hits = vec![Hit { text_id: 0, position: 0 }; self.count()];
        check(unsafe {
            gdx_cursor_locate_many(self.index.raw, &self.start, &self.end, 1, offsets.as_mut_ptr(), hits.as_mut_ptr(),
                                   hits.len() as u64, &mut total)
        });
        hits.into_iter()
    }
}

/// Index replicas on several GPUs of one node behind one handle: the batch is cut into contiguous shards, one
/// pipeline thread per device; results equal the one-GPU results bit for bit.
pub struct MultiGpuFmIndex {
    raw: *mut gdx_multi_t,
}
unsafe impl Send for MultiGpuFmIndex {}
unsafe impl Sync for MultiGpuFmIndex {}
impl Drop for MultiGpuFmIndex {
    fn drop(&mut self) {
        unsafe { gdx_multi_free(self.raw) }
    }
}
impl MultiGpuFmIndex {
    pub fn construct<T: AsRef<[u8]>>(
        texts: impl IntoIterator<Item = T>, io_to_dense: &[u8; 256], sigma: usize, n_searchable: usize,
        sa_rate: usize, lookup_depth: usize, index_width: i32, devices: &[i32], options: &BuildOptions,
    ) -> Self {
        let (buf, off) = pack(texts);
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            gdx_multi_build(buf.as_ptr(), off.as_ptr(), off.len() as u64 - 1, io_to_dense.as_ptr(), sigma as c_int,
                            n_searchable as c_int, sa_rate as u64, lookup_depth as c_int, index_width,
                            devices.as_ptr(), devices.len() as c_int, options, &mut raw)
        });
        Self { raw }
    }
    pub fn count_many<Q: AsRef<[u8]>>(&self, queries: impl IntoIterator<Item = Q>) -> impl Iterator<Item = usize> {
        let (buf, off) = pack(queries);
        let nq = off.len() - 1;
        let mut counts = vec![0u64; nq];
        check(unsafe { gdx_multi_count_many(self.raw, buf.as_ptr(), off.as_ptr(), nq as u64, counts.as_mut_ptr(), std::ptr::null_mut()) });
        counts.into_iter().map(|c| c as usize)
    }
    pub fn locate_many<Q: AsRef<[u8]>>(&self, queries: impl IntoIterator<Item = Q>) -> Hits {
        let (buf, off) = pack(queries);
        let nq = off.len() - 1;
        let mut offsets = vec![0u64; nq + 1];
        let (mut total, mut ptr) = (0u64, std::ptr::null_mut());
        check(unsafe {
            gdx_multi_locate_many_alloc(self.raw, buf.as_ptr(), off.as_ptr(), nq as u64, offsets.as_mut_ptr(), &mut ptr,
                                        &mut total, std::ptr::null_mut())
        });
        Hits { ptr, total: total as usize, offsets }
    }
}
