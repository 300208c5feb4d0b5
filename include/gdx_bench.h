/*
 * gdx_bench.h -- measurement helpers exported by libgdx.so next to the query ABI of gdx.h:
 * synthetic workloads generated directly in HBM, roofline
 * micro-benchmarks (streaming copy, random line gather) and step counters that give the
 * algorithmic byte counts of a run.  None of this is part of the
 * drop-in boundary.
 */
#ifndef GDX_BENCH_H
#define GDX_BENCH_H

#include <stdint.h>

#include "gdx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t sa_initial_order;      /* symbols fixed by the first key sort of the suffix sorter */
    uint64_t sa_pending_after_sort; /* suffixes still tied after it                             */
    uint64_t sa_rounds;             /* prefix-doubling rounds that followed                     */
    double seconds_encode, seconds_sa, seconds_bwt, seconds_table, seconds_lookup, seconds_pairs;
} gdx_build_stats_t;

int gdx_index_build_stats(const gdx_index_t *ix, gdx_build_stats_t *out);

/* IO text of n symbols: symbol i depends only on (seed, i): r = splitmix64 output number i of the
 * stream seeded with `seed`; 'N' if (r >> 32) * 1e6 < n_per_million * 2^32, else "ACGT"[(r >> 8) & 3]. */
int gdx_synth_text_dev(void *d_out, uint64_t n, uint64_t seed, uint32_t n_per_million, void *stream);

/* nq queries with lengths uniform in [len_min, len_max]; a query is, with probability
 * sampled_per_million / 1e6, a substring of one text without 'N' (up to 8 draws, else random),
 * otherwise uniform random over ACGT.  d_text_offsets: u64[n_texts+1] offsets into d_io_text.
 * Writes d_qoff (u64[nq+1]) and d_qbuf; *out_total_bytes = qoff[nq].  Synchronises the stream. */
int gdx_synth_queries_dev(const void *d_io_text, const void *d_text_offsets, uint64_t n_texts, uint64_t nq,
                          uint32_t len_min, uint32_t len_max, uint32_t sampled_per_million, uint64_t seed,
                          void *d_qoff, void *d_qbuf, uint64_t qbuf_capacity, uint64_t *out_total_bytes,
                          void *stream);

/* search kernel used on rank lines: 2 = pair lines (two LF steps per 128-byte line fetch; default),
 * 0 = quad (four lanes fetch one 64-byte line), 1 = one lane per query, -1 = re-read GDX_SEARCH_VARIANT.
 * All variants return identical results; the switch exists for A/B measurements and parity tests. */
int gdx_debug_set_search_variant(int variant);

/* tests: index_width 64 takes the 64-bit engine (wide.hip) even when the collection would fit 32-bit rows */
int gdx_debug_force_wide(int on);

/* chunk size of the pipeline behind the host-pointer query calls (queries and query bytes per chunk; 0 = default
 * 2^20 queries / 32 MB): tests force many small chunks through it */
int gdx_debug_set_host_chunking(uint64_t queries, uint64_t bytes);

/* streaming copy of `bytes` (multiple of 16): the "measured HBM bandwidth" denominator */
int gdx_bench_stream_copy(void *d_dst, const void *d_src, uint64_t bytes, void *stream);
/* streaming read of `bytes` (multiple of 16) */
int gdx_bench_stream_read(const void *d_src, uint64_t bytes, void *d_sink, void *stream);
/* n_accesses independent reads of random aligned lines of line_bytes (64 or 128) out of n_lines.
 * mode 0: one lane reads a whole line (the access shape of the lane-per-query rank);
 * mode 1: line_bytes/16 adjacent lanes read one line, 16 bytes each;
 * mode 2: as mode 0 but every next line depends on the data just loaded (a dependent chain per lane,
 *         the shape of an LF loop);
 * mode 3: every lane reads one entry of line_bytes = 8, 16 or 32 (the shape of a one-lane-per-query search reading
 *         top-table and jump-table entries).  n_accesses should be a multiple of 2^20.  d_sink: u32[1]. */
int gdx_bench_random_gather(const void *d_src, uint64_t n_lines, uint32_t line_bytes, uint64_t n_accesses,
                            uint64_t seed, uint32_t mode, void *d_sink, void *stream);

/* d_steps (u64[3], pre-zeroed by the caller): [0] += LF steps the search of these queries executes (the
 * reference's count: a pair step is 2, a jump is 8 to 32, the top table its depth); on pair lines also [1] += line fetches summed over the
 * queries and [2] += the fetch slots their wavefronts spent on them (lanes of a finished query idle until the
 * longest query of the wave ends), so [1] / [2] is the active-lane fraction of the search. */
int gdx_search_step_stats_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                              void *d_steps, void *stream);
/* same call as gdx_locate_intervals_dev; d_steps (u64[2], pre-zeroed): [0] += locate walk steps executed,
 * [1] += hits that had to walk at all */
int gdx_locate_step_stats_dev(const gdx_index_t *ix, const void *d_start, const void *d_end, uint64_t m,
                              const void *d_hit_offsets, uint64_t total_hits, void *d_hits, void *d_workspace,
                              void *d_steps, void *stream);

/* the same call as gdx_locate_many_hits_dev; d_steps (u64[2], pre-zeroed): [0] += walk steps actually executed
 * (with the records' hints), [1] += hits that had to walk at all */
int gdx_locate_many_hits_stats_dev(const gdx_index_t *ix, const void *d_records, uint64_t nq, const void *d_hit_offsets,
                                   uint64_t total_hits, void *d_hits, void *d_workspace, void *d_steps, void *stream);

/* Independent check of a built index (tests/test_gpu_fullsize.py): from each of the m start rows d_rows[i] (u32) walk
 * `steps` LF steps on the occurrence table and write the BWT symbol met at every step (dense code) to
 * d_symbols[i * steps + j] -- the text read backwards from position SA[row] - 1.  A chain that meets the sentinel stops:
 * that entry is 0, the rest of the chain 0xff.  d_end_rows (u32[m], may be null): the row each chain stopped at. */
int gdx_bench_lf_walk_dev(const gdx_index_t *ix, const void *d_rows, uint64_t m, uint32_t steps, void *d_symbols,
                          void *d_end_rows, void *stream);

/* Query acceleration structures the index carries beside the reference's arrays (DESIGN.md "HBM layout"):
 * out[0] = 1 if pair lines are present, out[1] = bytes per jump-table entry (0 = none, 8 or 16),
 * out[2] = depth of the top table (0 = none), out[3] = bit 0: full suffix array present, bit 1: text units present, bit 2: inverse
 * suffix array present, bit 3: the index has the library's DEFAULT SHAPE (gdx.h gdx_build_options_t: every option was left at its
 * default and the shape fitted the budget). */
int gdx_index_aux_info(const gdx_index_t *ix, uint32_t out[4]);

/* Drops the pair lines / jump table / top table of an index and builds them again with other options, without
 * repeating the suffix sort (the ladder of design points in bench.py).  NOT safe against concurrent queries. */
int gdx_index_rebuild_aux(gdx_index_t *ix, const gdx_build_options_t *opts);

#ifdef __cplusplus
}
#endif
#endif
