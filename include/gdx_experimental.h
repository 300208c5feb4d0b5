/* gdx_experimental.h -- entry points of libgdx.so beside the core ABI of gdx.h: the steps of rounds 1-5 that the core calls have
 * superseded (separate search / offsets / hits calls around host round trips, the `_packed` calls that gdx_query_layout_t
 * replaced, hint arrays, compact-result plumbing of the multi-GPU gather) and building blocks the tests and bench.py still
 * drive one by one.  Exported and tested like the core (tests/test_abi.py, tests/test_gpu_*.py), but not what a binding of the
 * reference's API needs (INTEGRATION.md binds gdx.h only), and free to change.  Every call returns the results of the core call
 * it is a part of: lib.rs:155-246 of the reference. */
#ifndef GDX_EXPERIMENTAL_H
#define GDX_EXPERIMENTAL_H

#include "gdx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The same import for the reference's other table variants (lib.rs:102-113): table_kind 0 =
 * CondensedTextWithRankSupport (condensed.rs:24-30), 1 = FlatTextWithRankSupport (flat.rs:27-33: one
 * indicator block per symbol, 16-bit block offset in the low bits of each block); block_bits 64 = Block64,
 * 512 = Block512 (block.rs).  interleaved_blocks is passed as u64 words (a Block512 is 8 words). */
int gdx_index_from_parts_ex(int table_kind, int block_bits, const uint64_t *count,
                            const uint64_t *interleaved_blocks, uint64_t n, const uint32_t *sa_samples,
                            uint64_t sa_rate, const uint64_t *border_keys, const uint64_t *border_vals,
                            const uint64_t *sentinel_indices, uint64_t n_texts, const uint8_t *io_to_dense,
                            int sigma, int n_searchable, int lookup_depth, int index_width, int device_id,
                            gdx_index_t **out);

int gdx_cursors_for_many_queries_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff /*u64*/,
                                     uint64_t nq, void *d_out_start /*u32*/, void *d_out_end /*u32*/,
                                     void *d_out_status /*u8 or NULL*/, void *stream);

int gdx_count_many_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                       void *d_out_counts /*u32*/, void *d_out_status, void *stream);

/* exclusive scan of (end-start) into d_hit_offsets (u64[m+1]); needs no workspace sizing */
int gdx_hit_offsets_dev(const gdx_index_t *ix, const void *d_start, const void *d_end, uint64_t m,
                        void *d_hit_offsets, void *stream);

int gdx_locate_intervals_dev(const gdx_index_t *ix, const void *d_start, const void *d_end, uint64_t m,
                             const void *d_hit_offsets, uint64_t total_hits, void *d_hits /*gdx_hit32_t*/,
                             void *d_workspace, void *stream);

/* The same two calls with a locate hint carried from the search to the locate of the SAME intervals (the device
 * form of the fused gdx_locate_many): d_hint is an opaque device array of 8 bytes per query (8-byte aligned),
 * written by gdx_cursors_for_many_queries_hint_dev and read by gdx_locate_intervals_hint_dev.  For a query whose
 * interval is one row wide it may name a sampled suffix-array row the search passed through and its distance
 * to the hit, which saves the walk of sampled_suffix_array.rs:118-131; results are identical with or without it.
 * The hint is only valid together with the d_start / d_end arrays of the call that produced it. */
int gdx_cursors_for_many_queries_hint_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                          void *d_out_start, void *d_out_end, void *d_out_status, void *d_hint,
                                          void *stream);

int gdx_locate_intervals_hint_dev(const gdx_index_t *ix, const void *d_start, const void *d_end, uint64_t m,
                                  const void *d_hit_offsets, uint64_t total_hits, void *d_hits, void *d_workspace,
                                  const void *d_hint, void *stream);

/* ---- fused count + locate on the device (the device form of FmIndex::locate_many, lib.rs:179-185) -----------------
 * The search writes one opaque 16-byte record per query (d_records: 16-byte aligned, 16 * nq bytes) that carries the
 * query's number of occurrences, its status and what locate needs.  Because only the hits matter here -- not the
 * suffix-array interval itself -- the search may finish a query from a jump-table entry it already holds instead of
 * fetching one more line ("lazy tail"); counts and hits are the reference's, bit for bit and in
 * the same order.
 *   1. gdx_locate_many_search_dev     qbuf, qoff -> records
 *   2. gdx_locate_many_offsets_dev    records -> d_hit_offsets (u64[nq+1], exclusive scan of the counts);
 *                                     d_hit_offsets[nq] is the number of hits (read it back to size d_hits)
 *   3. gdx_locate_many_hits_dev       records + offsets -> d_hits (gdx_hit32_t[total]), hits of query i at
 *                                     [off[i], off[i+1]) in suffix-array order (lib.rs:187-197);
 *                                     d_workspace: gdx_locate_workspace_bytes(total) bytes
 * gdx_locate_many_unpack_dev extracts counts (u32) and / or status bytes from the records (either may be NULL). */
int gdx_locate_many_search_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff /*u64*/, uint64_t nq,
                               void *d_records, void *stream);

int gdx_locate_many_offsets_dev(const gdx_index_t *ix, const void *d_records, uint64_t nq, void *d_hit_offsets,
                                void *stream);

/* the same scan with a per-query limit: a query with more than max_hits occurrences gets no hit slots (it is still
 * counted -- gdx_locate_many_unpack_dev reports its count -- but not located: what read mappers do with reads that
 * fall into repeats); max_hits = 0 means no limit */
int gdx_locate_many_offsets_capped_dev(const gdx_index_t *ix, const void *d_records, uint64_t nq, uint32_t max_hits,
                                       void *d_hit_offsets, void *stream);

int gdx_locate_many_hits_dev(const gdx_index_t *ix, const void *d_records, uint64_t nq, const void *d_hit_offsets,
                             uint64_t total_hits, void *d_hits, void *d_workspace, void *stream);

int gdx_locate_many_unpack_dev(const gdx_index_t *ix, const void *d_records, uint64_t nq, void *d_out_counts,
                               void *d_out_status, void *stream);

/* COMPACT results beside the records: d_compact (u32[nq], device) holds per query the text position of its ONLY hit
 * (concatenated texts incl. sentinels, as the resolved records), 0xffffffff = no occurrence, or 0xfffffffe = "see
 * d_records[q]" (several hits, a row that still has to be located, a status).  On an index with a seed table
 * (gdx_build_options_t.seed_symbols) the search answers nearly every read of a text without repeats this way and leaves
 * their 16-byte records untouched; offsets and hits then stream 4 bytes per query instead of 16.  On any other index every
 * entry says "see the record" (same results, no gain).  The four calls mirror gdx_locate_many_search_dev /
 * _offsets_capped_dev / _hits_dev / _unpack_dev; results are identical to theirs (lib.rs:155-185). */
int gdx_locate_many_search_compact_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                       void *d_records, void *d_compact, void *stream);

int gdx_locate_many_offsets_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                        uint32_t max_hits, void *d_hit_offsets, void *stream);

int gdx_locate_many_hits_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                     const void *d_hit_offsets, uint64_t total_hits, void *d_hits, void *d_workspace, void *stream);

int gdx_locate_many_unpack_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                       void *d_out_counts, void *d_out_status, void *stream);

/* Compact results in the form they travel in between devices (the multi-GPU gather): per query one
 * text id byte and one int32 -- the position in that text of the query's only hit (lib.rs:155-185 Hit { text_id, position }),
 * -1 = no occurrence, -2 = "see the record" (the sender ships those queries' counts and hits beside).  5 bytes per query on
 * the receiving side from 4 on the wire.  Collections of at most 256 texts (else GDX_ERR_UNSUPPORTED); d_compact and
 * d_out_positions 16-byte aligned, d_out_text_ids 4-byte aligned. */
int gdx_compact_split_hits_dev(const gdx_index_t *ix, const void *d_compact, uint64_t nq, void *d_out_text_ids,
                               void *d_out_positions, void *stream);

/* The queries whose compact result says "see the record": their numbers (u32) into d_out_queries in NO particular order, as
 * many as `capacity` holds; *d_out_n (u64, device) = how many there are in all.  One streaming pass over d_compact (16-byte
 * aligned).  The sender of the multi-GPU gather lists its exceptions with this (on a text without repeats a few in a
 * million queries) and sorts them. */
int gdx_compact_exceptions_dev(const gdx_index_t *ix, const void *d_compact, uint64_t nq, void *d_out_queries,
                               uint64_t capacity, void *d_out_n, void *stream);

/* Offsets and hits in two calls around the ONE host round trip of a count + locate step (instead of offsets, round trip,
 * hits): gdx_locate_many_totals_compact_dev sums the counts -- d_totals (u64[2], device) = {all hit slots, the slots of the
 * queries whose compact result says "see the record"} -- and leaves the bases of its tiles in d_scan_workspace
 * (gdx_locate_many_totals_workspace_bytes(nq) bytes); the caller reads d_totals back, sizes d_hits (total_hits entries of
 * gdx_hit32_t) and d_workspace (gdx_locate_workspace_bytes(total_hits), only needed when rest_hits != 0) and calls
 * gdx_locate_many_offsets_hits_compact_dev, which writes d_hit_offsets (u64[nq + 1]) and in the SAME pass over the compact
 * results stores the hit of every query they answer, then locates the remaining rest_hits slots from the records.
 * d_compact may be NULL (records only: the second call is then the plain offsets + hits).  Same results as the other
 * record calls. */
uint64_t gdx_locate_many_totals_workspace_bytes(uint64_t nq);

int gdx_locate_many_totals_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                       uint32_t max_hits, void *d_scan_workspace, void *d_totals, void *stream);

int gdx_locate_many_offsets_hits_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                             uint32_t max_hits, const void *d_scan_workspace, void *d_hit_offsets,
                                             uint64_t total_hits, uint64_t rest_hits, void *d_hits, void *d_workspace, void *stream);

/* the same with NARROW hit offsets: d_hit_offsets32 is u32[nq + 1] (total_hits < 2^32, else GDX_ERR_INVALID_ARGUMENT): 4
 * bytes per query less to write -- a fifth of this pass's traffic on a batch of reads with one hit each */
int gdx_locate_many_offsets32_hits_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                               uint32_t max_hits, const void *d_scan_workspace, void *d_hit_offsets32,
                                               uint64_t total_hits, uint64_t rest_hits, void *d_hits, void *d_workspace,
                                               void *stream);

/* gdx_count_many / gdx_cursors_for_many_queries on packed host buffers (the same chunked pipeline) */
int gdx_count_many_packed(const gdx_index_t *ix, const uint8_t *packed, const uint64_t *qoff, uint64_t nq,
                          uint64_t *out_counts, uint8_t *out_status);

int gdx_cursors_for_many_queries_packed(const gdx_index_t *ix, const uint8_t *packed, const uint64_t *qoff,
                                        uint64_t nq, uint64_t *out_start, uint64_t *out_end, uint8_t *out_status);

/* the device-resident search calls on packed buffers; the rest of a locate (gdx_locate_many_offsets_dev,
 * gdx_locate_many_hits_dev) is unchanged */
int gdx_cursors_for_many_queries_packed_dev(const gdx_index_t *ix, const void *d_packed, const void *d_qoff,
                                            uint64_t nq, void *d_out_start, void *d_out_end, void *d_out_status,
                                            void *stream);

int gdx_count_many_packed_dev(const gdx_index_t *ix, const void *d_packed, const void *d_qoff, uint64_t nq,
                              void *d_out_counts, void *d_out_status, void *stream);

int gdx_locate_many_search_packed_dev(const gdx_index_t *ix, const void *d_packed, const void *d_qoff, uint64_t nq,
                                      void *d_records, void *stream);

/* gdx_locate_many_search_compact_dev / gdx_locate_many_search_dev / gdx_count_many_dev / gdx_cursors_for_many_queries_dev
 * on a batch in the given layout (FmIndex::locate_many / count_many / cursors_for_many_queries, lib.rs:155-246); the rest
 * of a locate (gdx_locate_many_totals_compact_dev, ..._offsets_hits_compact_dev, ..._hits_dev) never looks at the queries */
int gdx_locate_many_search_compact_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                              const gdx_query_layout_t *layout, void *d_records, void *d_compact, void *stream);

int gdx_locate_many_search_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                      const gdx_query_layout_t *layout, void *d_records, void *stream);

/* gdx_locate_many_search_compact_layout_dev and gdx_locate_many_totals_compact_dev in one call: d_scan_workspace
 * (gdx_locate_many_totals_workspace_bytes(nq)) and d_totals (u64[2]: all hit slots, those the compact results leave open)
 * are what gdx_locate_many_offsets_hits_compact_dev takes next.  On an index whose count / locate search is the seed
 * table's lane kernel the search counts its hits per scan tile while it stores them, and the separate pass over the compact
 * results (4 bytes per query) is gone; on any other index the call is the two calls one after the other. */
int gdx_locate_many_search_totals_compact_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                                     const gdx_query_layout_t *layout, uint32_t max_hits, void *d_records,
                                                     void *d_compact, void *d_scan_workspace, void *d_totals, void *stream);

int gdx_multi_from_indexes(gdx_index_t **replicas, int n_replicas, gdx_multi_t **out);

#ifdef __cplusplus
}
#endif
#endif
