//! Rust binding of libgdx.so (include/gdx.h) with safe wrappers named like genedex's API.
//!
//! NOT compiled in this repository's image (no rustc/cargo); kept as the reference-side stub a
//! genedex maintainer would add, e.g. as `src/gpu.rs` behind a `gpu` cargo feature, linking with
//! `cargo:rustc-link-lib=dylib=gdx`.
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

/// lib.rs:331-335
#[repr(C)]
#[derive(Debug, Clone, Copy, PartialEq, Eq, PartialOrd, Ord, Hash)]
pub struct Hit {
    pub text_id: u64,
    pub position: u64,
}

pub const GDX_OK: c_int = 0;
pub const GDX_ERR_CAPACITY: c_int = 5;
pub const GDX_ERR_QUERY_STATUS: c_int = 6;

extern "C" {
    pub fn gdx_last_error() -> *const c_char;
    pub fn gdx_index_build(
        texts_buf: *const u8, text_offsets: *const u64, n_texts: u64, io_to_dense: *const u8,
        sigma: c_int, n_searchable: c_int, sa_rate: u64, lookup_depth: c_int, index_width: c_int,
        device_id: c_int, out: *mut *mut gdx_index_t,
    ) -> c_int;
    pub fn gdx_index_from_parts(
        count: *const u64, interleaved_blocks: *const u64, n: u64, sa_samples: *const u32, sa_rate: u64,
        border_keys: *const u64, border_vals: *const u64, sentinel_indices: *const u64, n_texts: u64,
        io_to_dense: *const u8, sigma: c_int, n_searchable: c_int, lookup_depth: c_int,
        index_width: c_int, device_id: c_int, out: *mut *mut gdx_index_t,
    ) -> c_int;
    /// table_kind 0 = condensed, 1 = flat; block_bits 64 | 512 (FmIndexCondensed64/512, FmIndexFlat64/512)
    pub fn gdx_index_from_parts_ex(
        table_kind: c_int, block_bits: c_int, count: *const u64, interleaved_blocks: *const u64, n: u64,
        sa_samples: *const u32, sa_rate: u64, border_keys: *const u64, border_vals: *const u64,
        sentinel_indices: *const u64, n_texts: u64, io_to_dense: *const u8, sigma: c_int, n_searchable: c_int,
        lookup_depth: c_int, index_width: c_int, device_id: c_int, out: *mut *mut gdx_index_t,
    ) -> c_int;
    pub fn gdx_index_save(ix: *const gdx_index_t, path: *const c_char) -> c_int;
    pub fn gdx_index_load(path: *const c_char, device_id: c_int, out: *mut *mut gdx_index_t) -> c_int;
    pub fn gdx_index_free(ix: *mut gdx_index_t);
    pub fn gdx_count_many(
        ix: *const gdx_index_t, qbuf: *const u8, qoff: *const u64, nq: u64, out_counts: *mut u64,
        out_status: *mut u8,
    ) -> c_int;
    pub fn gdx_cursors_for_many_queries(
        ix: *const gdx_index_t, qbuf: *const u8, qoff: *const u64, nq: u64, out_start: *mut u64,
        out_end: *mut u64, out_status: *mut u8,
    ) -> c_int;
    pub fn gdx_locate_many(
        ix: *const gdx_index_t, qbuf: *const u8, qoff: *const u64, nq: u64, out_hit_offsets: *mut u64,
        hits: *mut Hit, hits_capacity: u64, out_total: *mut u64, out_status: *mut u8,
    ) -> c_int;
    pub fn gdx_cursor_empty(ix: *const gdx_index_t, start: *mut u64, end: *mut u64) -> c_int;
    pub fn gdx_cursor_extend_front_many(
        ix: *const gdx_index_t, start: *mut u64, end: *mut u64, io_symbols: *const u8, m: u64,
        out_status: *mut u8,
    ) -> c_int;
    pub fn gdx_cursor_locate_many(
        ix: *const gdx_index_t, start: *const u64, end: *const u64, m: u64, out_hit_offsets: *mut u64,
        hits: *mut Hit, hits_capacity: u64, out_total: *mut u64,
    ) -> c_int;
    pub fn gdx_rank_many(
        ix: *const gdx_index_t, symbols: *const u8, idx: *const u64, m: u64, out: *mut u64,
    ) -> c_int;
    // device-resident variants take *const c_void device pointers and a hipStream_t
    pub fn gdx_count_many_dev(
        ix: *const gdx_index_t, d_qbuf: *const c_void, d_qoff: *const c_void, nq: u64,
        d_out_counts: *mut c_void, d_out_status: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    // search + locate on device buffers: the search leaves an opaque 8-byte hint per query for the locate of the
    // same intervals (a sampled suffix-array row the query passed through), which then needs no walk
    pub fn gdx_cursors_for_many_queries_hint_dev(
        ix: *const gdx_index_t, d_qbuf: *const c_void, d_qoff: *const c_void, nq: u64, d_out_start: *mut c_void,
        d_out_end: *mut c_void, d_out_status: *mut c_void, d_hint: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    pub fn gdx_hit_offsets_dev(
        ix: *const gdx_index_t, d_start: *const c_void, d_end: *const c_void, m: u64, d_hit_offsets: *mut c_void,
        stream: *mut c_void,
    ) -> c_int;
    pub fn gdx_locate_workspace_bytes(total_hits: u64) -> u64;
    // FASTA / FASTQ ingestion into (qbuf, qoff) batches (host only)
    pub fn gdx_fastx_open(path: *const c_char, out: *mut *mut gdx_fastx_t) -> c_int;
    pub fn gdx_fastx_next_batch(
        reader: *mut gdx_fastx_t, qbuf: *mut u8, qbuf_capacity: u64, qoff: *mut u64, max_records: u64, n_out: *mut u64,
    ) -> c_int;
    pub fn gdx_fastx_close(reader: *mut gdx_fastx_t);
    pub fn gdx_locate_intervals_hint_dev(
        ix: *const gdx_index_t, d_start: *const c_void, d_end: *const c_void, m: u64, d_hit_offsets: *const c_void,
        total_hits: u64, d_hits: *mut c_void, d_workspace: *mut c_void, d_hint: *const c_void, stream: *mut c_void,
    ) -> c_int;
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
        sa_rate: usize, lookup_depth: usize, index_width: i32, device: i32,
    ) -> Self {
        let (buf, off) = pack(texts);
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            gdx_index_build(buf.as_ptr(), off.as_ptr(), off.len() as u64 - 1, io_to_dense.as_ptr(), sigma as c_int,
                            n_searchable as c_int, sa_rate as u64, lookup_depth as c_int, index_width, device, &mut raw)
        });
        Self { raw }
    }

    /// lib.rs:155-161
    pub fn count_many<Q: AsRef<[u8]>>(&self, queries: impl IntoIterator<Item = Q>) -> Vec<usize> {
        let (buf, off) = pack(queries);
        let nq = off.len() - 1;
        let mut counts = vec![0u64; nq];
        check(unsafe { gdx_count_many(self.raw, buf.as_ptr(), off.as_ptr(), nq as u64, counts.as_mut_ptr(), std::ptr::null_mut()) });
        counts.into_iter().map(|c| c as usize).collect()
    }

    /// lib.rs:147-149
    pub fn count(&self, query: &[u8]) -> usize {
        self.count_many([query])[0]
    }

    /// lib.rs:179-185; hits of query i are `hits[offsets[i]..offsets[i+1]]`
    pub fn locate_many<Q: AsRef<[u8]>>(&self, queries: impl IntoIterator<Item = Q>) -> (Vec<u64>, Vec<Hit>) {
        let (buf, off) = pack(queries);
        let nq = off.len() - 1;
        let mut offsets = vec![0u64; nq + 1];
        let mut total = 0u64;
        let rc = unsafe {
            gdx_locate_many(self.raw, buf.as_ptr(), off.as_ptr(), nq as u64, offsets.as_mut_ptr(), std::ptr::null_mut(), 0,
                            &mut total, std::ptr::null_mut())
        };
        if rc != GDX_ERR_CAPACITY { check(rc); }
        let mut hits = vec![Hit { text_id: 0, position: 0 }; total as usize];
        if total > 0 {
            check(unsafe {
                gdx_locate_many(self.raw, buf.as_ptr(), off.as_ptr(), nq as u64, offsets.as_mut_ptr(), hits.as_mut_ptr(),
                                total, &mut total, std::ptr::null_mut())
            });
        }
        (offsets, hits)
    }

    /// lib.rs:241-246
    pub fn cursors_for_many_queries<'a, Q: AsRef<[u8]>>(&'a self, queries: impl IntoIterator<Item = Q>) -> Vec<GpuCursor<'a>> {
        let (buf, off) = pack(queries);
        let nq = off.len() - 1;
        let (mut s, mut e) = (vec![0u64; nq], vec![0u64; nq]);
        check(unsafe {
            gdx_cursors_for_many_queries(self.raw, buf.as_ptr(), off.as_ptr(), nq as u64, s.as_mut_ptr(), e.as_mut_ptr(),
                                         std::ptr::null_mut())
        });
        s.into_iter().zip(e).map(|(start, end)| GpuCursor { index: self, start, end }).collect()
    }

    /// lib.rs:202-210
    pub fn cursor_empty(&self) -> GpuCursor<'_> {
        let (mut start, mut end) = (0u64, 0u64);
        check(unsafe { gdx_cursor_empty(self.raw, &mut start, &mut end) });
        GpuCursor { index: self, start, end }
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
    pub fn locate(&self) -> Vec<Hit> {
        let mut offsets = [0u64; 2];
        let mut total = 0u64;
        let mut hits = vec![Hit { text_id: 0, position: 0 }; self.count()];
        check(unsafe {
            gdx_cursor_locate_many(self.index.raw, &self.start, &self.end, 1, offsets.as_mut_ptr(), hits.as_mut_ptr(),
                                   hits.len() as u64, &mut total)
        });
        hits
    }
}
