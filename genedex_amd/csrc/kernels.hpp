// kernels.hpp -- host-callable launchers of the HIP kernels (defined in the .hip files)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/gdx.h"
#include "layout.hpp"

namespace gdx {

// Kernel choices of a query call (gdx_query_options_t); every combination returns identical results.
// Negative / zero = default; the GDX_* environment variables only override defaults (debugging aid).
struct QueryOptions {
    int search_variant = -1;   // -1 default (pair lines when present), 0 quad, 1 one lane per query, 2 pair
    int search_lanes = 0;      // 0 default (4), 4 or 8 lanes per query in the pair kernels
    int load_policy = -1;      // -1 default (0 plain), 1 sc1
    int length_schedule = -1;  // -1 default (1: blocks order spread-out ranges by length), 0 off
    int locate_variant = -1;   // (ignored: the lock-step locate variants of rounds 1-4 are gone, the chunk kernels remain)
    int locate_jump_walk = -1; // -1 default (1: the queue kernel walks through the jump table), 0 rank lines only
    int search_fast = -1;         // count / locate searches run the fast-path kernel first (1; 2 = with 16-row jumps) or not (0); -1 = per index
    int search_defer_after = -1;  // -1 default (3): load rounds beyond its allowance after which a query is parked and
                                  // finished in the block's straggler pass; 0 = never
    int search_exact = -1;        // exact-interval and cursor searches run search_exact_kernel4 first (default) or not (0)
    int search_seed = -1;         // count / locate searches start from the seed table when the index has one (default) or not (0)
    uint32_t max_hits_per_query = 0;  // host-pointer locate calls: at most this many hits per query, the first ones in
                                      // suffix-array order (locate(q).take(k) of the reference's lazy iterator); 0 = all
};

// Active lists of the cursor-extension mode (search mode 2): the cursors to extend are those listed in active_in
// (*n_active_in of them, a device value; null = all), the ones still non-empty afterwards are appended to active_out
// (*n_active_out must be zeroed by the caller beforehand).
struct CursorArgs {
    const uint32_t *active_in = nullptr;
    const uint32_t *n_active_in = nullptr;
    uint32_t *active_out = nullptr;
    uint32_t *n_active_out = nullptr;
    // chunk view (gdx_cursor_extend_front_chunk_dev): when chunk_symbols != 0, cursor i is extended by chunk
    // chunk_index (counted from the end) of query [qbeg[i], qend[i]) and stays in the live list only while symbols
    // are left of that chunk
    uint32_t chunk_symbols = 0, chunk_index = 0;
    // where the fast-path kernel left the listed queries: {lo, hi, symbols left, 1} to go on from there, {.., 0} to
    // start from the beginning; indexed by query, null = all start from the beginning
    const uint4 *resume_state = nullptr;
};

// Search records (16 bytes per query, mode 1): {start, end, hint row, hint symbols | status << 24} -- end - start is
// the count; with a hint (row != 0xffffffff, count 1) the hit is SA[hint row] - hint symbols.  A MASKED record
// (kRecMasked set in the fourth word; search_fast_kernel4 on reads that end on several rows) is {first row, first row
// + count, mask, symbols | kRecMasked}: the hits are SA[first row + j] - symbols for the set bits j of the mask, in
// that order, which is the reference's.
constexpr uint32_t kRecMasked = 1u << 23;
// A RESOLVED record (32-byte jump entries carry SA[row]): {start, end, text position of the one hit, kRecResolved |
// status << 24}, end - start == 1 -- locate only turns the position into (text id, offset).  A resolved record OF TWO
// (search_fast_kernel4, a read that ends on two rows whose occurrences it has just verified): {second position, second
// position + 2, first position, kRecResolved} -- end - start is the count as everywhere (in 32-bit arithmetic), hit slot 0 is
// the third word, hit slot 1 the first; no row is named and locate fetches no suffix-array line.
constexpr uint32_t kRecResolved = 1u << 22;

// COMPACT results (optional, beside the records): one u32 per query = the text position of its only hit, kCompactNone = no
// occurrence, kCompactSee = look at the query's 16-byte record (several hits, an unresolved row, a status).  The seed
// kernel answers nearly every read of a text without repeats this way, and scan and locate then stream 4 bytes per
// query instead of 16 (the record slots of such queries are not written at all).
constexpr uint32_t kCompactNone = 0xffffffffu;
constexpr uint32_t kCompactSee = 0xfffffffeu;

// One search launch.  Query i = d_qbuf[d_qbeg[i] .. d_qend[i]) (d_qend = d_qbeg + 1 for the usual offsets array).
// mode 0: exact intervals (cursors_for_many_queries); mode 1: count / locate (end - start is the count, start / end
// are only meaningful through the hint or record; see search_pair_body); mode 2: cursor extension, start / end are
// in / out.  Any output may be null.  rec: one 16-byte record per query {start, end, hint row, hint symbols |
// status << 24} for launch_hit_offsets_rec / launch_locate.
constexpr uint32_t kSumTile = 2048;  // queries per tile of the two-pass offsets scan (locate.hip: kScan2Tile)

// Several small device regions zeroed by ONE launch: the list counters, tile sums, totals and chunk flags of a count +
// locate step were seven fills of ~4.6 us each -- a twentieth of the step a rank of eight runs on its 12.5 M reads.
// Regions are 4-byte aligned; sizes are rounded up to whole 32-bit words.
struct ZeroSet {
    static constexpr int kMax = 8;
    uint32_t *p[kMax] = {};
    uint32_t words[kMax] = {};
    int n = 0;
    void add(void *ptr, size_t bytes);
    void flush(hipStream_t stream);  // zeroes what was added (one kernel; nothing added: nothing launched) and empties the set
};

struct SearchCall {
    const uint8_t *d_qbuf = nullptr;
    const uint64_t *d_qbeg = nullptr, *d_qend = nullptr;
    uint64_t nq = 0;
    uint32_t *d_start = nullptr, *d_end = nullptr, *d_count = nullptr;
    uint8_t *d_status = nullptr;
    uint2 *d_hint = nullptr;
    uint4 *d_rec = nullptr;
    uint32_t *d_compact = nullptr;  // mode 1 with d_rec: compact results (above)
    unsigned long long *d_step_stats = nullptr;
    int mode = 0;
    bool packed = false;  // d_qbuf holds 2-bit codes, d_qbeg / d_qend count symbols (rank-line layout, sigma <= 8)
    // count / locate with compact results, optional: the hit totals of a locate folded into the search.  d_tile_sums[t] (one per
    // kSumTile queries, written by the call) = the hit slots of the queries of tile t (RecordSize with tile_max_hits),
    // *d_tile_rest (zeroed by the caller) += those of the queries whose compact result says "see the record"; *tile_sums_done
    // (host) tells whether the call did it -- the seed table's lane kernel on an index without pair lines does, by counting
    // what it answers and adding its (short) lists afterwards; otherwise launch_scan_totals reads the results once more
    unsigned long long *d_tile_sums = nullptr, *d_tile_rest = nullptr;
    uint32_t tile_max_hits = 0;
    bool *tile_sums_done = nullptr;
    uint32_t uniform_len = 0;  // != 0: a uniform batch -- every query has this many symbols, query i starts at symbol
                               // i * uniform_len; d_qbeg / d_qend may be null (gdx_query_layout_t)
    const ZeroSet *also_zero = nullptr;  // regions of the caller's to be zeroed before the call's first kernel (same launch
                                         // as the call's own counters)
    CursorArgs cursors;
};
void launch_search_call(const IndexView &ix, const SearchCall &call, hipStream_t stream,
                        const QueryOptions &qo = QueryOptions());

// ---- search.hip ---------------------------------------------------------------------------
// Backward search of nq queries (lookup jump + LF loop), one lane per query.
// Any of out_start/out_end/out_count/out_status may be null.  d_hint (optional, uint2[nq]): locate hints for
// launch_locate of exactly these intervals ({0xffffffff, 0} = none; see locate_queue_kernel).
void launch_search(const IndexView &ix, const uint8_t *d_qbuf, const uint64_t *d_qoff, uint64_t nq,
                   uint32_t *d_out_start, uint32_t *d_out_end, uint32_t *d_out_count, uint8_t *d_out_status,
                   hipStream_t stream, unsigned long long *d_step_stats = nullptr, uint2 *d_hint = nullptr,
                   const QueryOptions &qo = QueryOptions());
void set_search_variant(int v);  // 0 quad, 1 lane, 2 pair (default), -1 re-read the environment
// ASCII -> 2-bit packed queries on the device (include/gdx.h "packed queries")
void launch_pack_queries(const IndexView &ix, const uint8_t *d_qbuf, uint64_t n_symbols, uint8_t *d_packed,
                         uint8_t *d_bad_flags, unsigned long long *d_bad_symbols, hipStream_t stream);
void launch_extend_front(const IndexView &ix, uint32_t *d_start, uint32_t *d_end, const uint8_t *d_io_symbols,
                         uint64_t m, uint8_t *d_out_status, hipStream_t stream);
// d_error (u32, pre-zeroed) is set to 1 when an argument is out of range
void launch_rank_many(const IndexView &ix, const uint8_t *d_symbols, const uint32_t *d_idx, uint64_t m,
                      uint32_t *d_out, uint32_t *d_error, hipStream_t stream);
void launch_symbol_at_many(const IndexView &ix, const uint32_t *d_idx, uint64_t m, uint8_t *d_out,
                           uint32_t *d_error, hipStream_t stream);
// gdx_bench_lf_walk_dev (gdx_bench.h): `steps` LF steps from every start row, the BWT symbols met on the way
void launch_lf_walk(const IndexView &ix, const uint32_t *d_rows, uint64_t m, uint32_t steps, uint8_t *d_symbols,
                    uint32_t *d_end_rows, hipStream_t stream);
// fills lookup table `depth` (entries k^depth) of ix.lookup; d_lookup is the writable alias of ix.lookup
void launch_fill_lookup(const IndexView &ix, uint2 *d_lookup, int depth, hipStream_t stream);
// Top table of the pair kernels (IndexView::top): 4^depth entries, dense symbols 1..4 only, rank-line layout.
void launch_fill_top(const IndexView &ix, uint2 *d_top, uint32_t depth, hipStream_t stream);
// *d_sum (pre-zeroed) += widths of the top-table entries wider than `rows` rows
void launch_top_wide(const uint2 *d_top, uint32_t depth, uint32_t rows, unsigned long long *d_sum, hipStream_t stream);

// ---- locate.hip ---------------------------------------------------------------------------
size_t hit_offsets_temp_bytes(uint64_t m);
void launch_hit_offsets(const uint32_t *d_start, const uint32_t *d_end, uint64_t m, uint64_t *d_hit_offsets,
                        void *d_temp, size_t temp_bytes, hipStream_t stream);
// the same scan over 16-byte search records (SearchCall::d_rec)
size_t hit_offsets_rec_temp_bytes(uint64_t m);
// max_hits != 0: queries with more occurrences get no hit slots (counted, not located) -- or, with `take`, slots for
// their first max_hits rows
void launch_hit_offsets_rec(const uint4 *d_rec, uint64_t m, uint64_t *d_hit_offsets, void *d_temp, size_t temp_bytes,
                            hipStream_t stream, uint32_t max_hits = 0, bool take = false, const uint32_t *d_compact = nullptr);
// exclusive scan of m u32 counts into m + 1 u64 offsets (the root of gdx_multi_locate_many_gather_dev)
size_t count_offsets_temp_bytes(uint64_t m);
void launch_count_offsets(const uint32_t *d_counts, uint64_t m, uint64_t *d_offsets, void *d_temp, size_t temp_bytes,
                          hipStream_t stream);
// counts (end - start) and status bytes out of search records; d_any_status (optional): |= 1 when a status byte is not 0
void launch_unpack_records(const uint4 *d_rec, uint64_t m, uint32_t *d_counts, uint8_t *d_status, hipStream_t stream,
                           const uint32_t *d_compact = nullptr, unsigned long long *d_any_status = nullptr);
// compact results -> text id bytes and positions in the text (-1 none, -2 see the exceptions); at most 256 texts
void launch_compact_split(const IndexView &ix, const uint32_t *d_compact, uint64_t m, uint8_t *d_ids, int32_t *d_pos,
                          hipStream_t stream);
// the queries whose compact result says "see the record": unordered list (up to `capacity`), *d_n = how many there are
void launch_compact_exceptions(const uint32_t *d_compact, uint64_t m, uint32_t *d_list, uint64_t capacity,
                               unsigned long long *d_n, hipStream_t stream);
// the "found bitmap" wire of the multi-GPU gather (gdx_wire_pack_dev / gdx_wire_split_dev, locate.hip)
// The form the host-pointer call gdx_locate_many_alloc_layout32 sends across PCIe (host_api.hip, wire_host.hpp): the exceptions'
// hits as they are, the hit offset of every tile's first read (host threads expand tiles independently), and -- collections of
// 2..256 texts -- the found reads' text ids as bytes beside positions IN their text (the device has the text table in LDS;
// a host thread would spend more on that lookup than on everything else).
struct WireHostForm {
    gdx_hit32_t *d_exc_hits32 = nullptr;  // instead of d_exc_ids / d_exc_pos
    uint32_t *d_tile_off = nullptr;       // n_tiles + 1 entries
    uint8_t *d_found_ids = nullptr;       // != null: found_capacity bytes; d_found_pos then holds positions in the text
    const IndexView *ix = nullptr;        // (with d_found_ids)
    uint64_t hits_stored = ~0ull;         // hit slots of d_hits that were written (a step into too small a buffer leaves the rest)
};
size_t wire_pack_workspace_bytes(uint64_t m);
void launch_wire_pack(const uint32_t *d_compact, const void *d_hit_offsets, bool narrow_offsets, const gdx_hit32_t *d_hits, uint64_t m,
                      uint8_t *d_bitmap, uint32_t *d_tile_found, uint32_t *d_found_pos, uint64_t found_cap, uint32_t *d_exc_q,
                      uint32_t *d_exc_cnt, uint64_t exc_cap, uint8_t *d_exc_ids, int32_t *d_exc_pos, uint64_t exc_hits_cap,
                      uint32_t *d_meta, void *d_workspace, hipStream_t stream, const struct WireHostForm *host_form = nullptr);
constexpr uint32_t kWireTileReads = 2048;  // reads per tile of the wire (locate.hip kWireTile)
void launch_wire_split(const IndexView &ix, const uint8_t *d_bitmap, const uint32_t *d_tile_found, const uint32_t *d_found_pos,
                       uint64_t found_cap, uint64_t m, const uint32_t *d_exc_q, const uint32_t *d_meta, uint64_t exc_cap, uint8_t *d_ids,
                       int32_t *d_pos, hipStream_t stream);
size_t locate_workspace_bytes(uint64_t total_hits);
// HitT = gdx_hit32_t (wide == false) or gdx_hit_t (wide == true)
void launch_locate(const IndexView &ix, const uint32_t *d_start, const uint32_t *d_end, uint64_t m,
                   const uint64_t *d_hit_offsets, uint64_t total_hits, void *d_hits, bool wide,
                   void *d_workspace, hipStream_t stream, unsigned long long *d_step_stats = nullptr,
                   const uint2 *d_hint = nullptr, const QueryOptions &qo = QueryOptions(),
                   const uint4 *d_rec = nullptr, bool reference_walk = false,
                   const uint32_t *d_compact = nullptr, bool compact_stored = false, const uint8_t *d_chunk_flags = nullptr,
                   bool narrow_offsets = false,  // d_hit_offsets is u32[m + 1] (records path only)
                   const unsigned long long *d_total = nullptr);  // != null: the number of hit slots is read on the device (no host
                                                                  // round trip); total_hits is then the hit buffer's capacity
// Offsets scan in two steps with the one host round trip between them: launch_scan_totals leaves d_totals (u64[2]) = {all
// hit slots, the slots of queries whose compact result says "see the record"} and the tile bases in d_scan_workspace
// (scan_totals_workspace_bytes); launch_scan_offsets_store then writes the offsets and, in the same pass, the hit of every
// query whose compact result is its position.  launch_locate(..., compact_stored = true) fills what is left (totals[1]).
size_t scan_totals_workspace_bytes(uint64_t m);
void launch_scan_totals_finish(void *d_scan_workspace, uint64_t m, unsigned long long *d_totals, hipStream_t stream);
void launch_scan_totals(const uint4 *d_rec, const uint32_t *d_compact, uint64_t m, uint32_t max_hits, bool take,
                        void *d_scan_workspace, unsigned long long *d_totals, hipStream_t stream);
void launch_scan_offsets_store(const IndexView &ix, const uint4 *d_rec, const uint32_t *d_compact, uint64_t m, uint32_t max_hits,
                               bool take, const void *d_scan_workspace, uint64_t *d_hit_offsets, void *d_hits,
                               uint64_t hits_capacity, bool wide, hipStream_t stream, bool store = true,
                               uint8_t *d_chunk_flags = nullptr, bool narrow_offsets = false, bool flags_zeroed = false,
                               bool entry_sa = false,  // locate_entry_sa(ix, qo): the pass may locate small "see the record" queries itself
                               const unsigned long long *d_totals = nullptr);  // != null: ... only if the totals say they are few
bool locate_entry_sa(const IndexView &ix, const QueryOptions &qo);
// The whole count + locate step of a batch, enqueued without a host round trip (gdx_locate_many_step_compact_layout_dev, and
// every chunk of the host-pointer call gdx_locate_many_alloc_layout32): search (+ hit totals), offsets scan + the hits the
// compact results answer, the rest from the records.  call: the query side, d_rec and (optionally) d_compact, mode 1.
struct LocateStep {
    SearchCall call;
    uint32_t max_hits = 0;
    bool take = false;  // max_hits: the first max_hits rows of a query (locate(q).take(k)) instead of none
    void *d_scan_workspace = nullptr;          // scan_totals_workspace_bytes(nq)
    unsigned long long *d_totals = nullptr;    // u64[2]: all hit slots, those behind "see the record"
    void *d_hit_offsets = nullptr;             // u64[nq + 1], or u32[nq + 1] with narrow
    bool narrow = false;
    void *d_hits = nullptr;                    // gdx_hit32_t[hits_capacity]; what lies beyond is not stored
    uint64_t hits_capacity = 0;
    void *d_workspace = nullptr;               // locate_workspace_bytes(hits_capacity)
    hipEvent_t event_after_search = nullptr;   // recorded between the two halves
};
void launch_locate_step(const IndexView &ix, const LocateStep &step, hipStream_t stream, const QueryOptions &qo);
// the second half of a step whose totals the host has read (gdx_locate_many_offsets_hits_compact_dev): offsets + the hits the
// compact results answer in one pass, then the rest_hits slots behind "see the record" from the records
void launch_offsets_hits(const IndexView &ix, const uint4 *d_rec, const uint32_t *d_compact, uint64_t nq, uint32_t max_hits, bool take,
                         const void *d_scan_workspace, void *d_hit_offsets, bool narrow, uint64_t total_hits, uint64_t rest_hits,
                         void *d_hits, void *d_workspace, hipStream_t stream, const QueryOptions &qo);
size_t locate_chunk_flags_bytes(uint64_t total_hits);
void *locate_flags_region(uint8_t *d_chunk_flags);          // what is zeroed before the store pass: an "any flagged" word + the flags
size_t locate_flags_region_bytes(uint64_t total_hits);
// store == false: offsets only.  d_chunk_flags (inside the locate workspace at locate_chunk_flags_offset(total_hits), filled by
// the store pass): launch_locate then only visits the chunks of hit slots in which that pass left something open
size_t locate_chunk_flags_offset(uint64_t total_hits);
// reference_walk: walk one LF step at a time (sampled_suffix_array.rs:118-131) so that the steps counted through
// d_step_stats are the reference's
// d_rec != null: start / hint come from the search records instead of d_start / d_hint (d_start, d_end unused)

}  // namespace gdx
