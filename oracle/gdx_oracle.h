/*
 * gdx_oracle.h -- CPU restatement of the genedex v0.2.2 query hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under genedex_amd/ may include, link or
 * call this.  Allowed callers: tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py.
 *
 * The reference (/root/reference, Rust) cannot be compiled in this image (no
 * rustc/cargo, un-vendored libsais 0.2.0 / psacak 0.1.0), so this file restates
 * the algorithm in plain C, one function per reference function, each citing
 * the reference file:line it follows.  Parity is pinned by the reference's own
 * known-answer tests and naive-search/naive-rank property tests
 * (tests/test_oracle_*.py, tests/golden/).
 *
 * Third-party arithmetic not in /root/reference: suffix sorting (libsais
 * 0.2.0, Cargo.lock:343-362).  libsais returns THE suffix array of the byte
 * string (sentinels are ordinary symbols 0, end-of-string is smallest), which
 * is unique, so any correct suffix sorter reproduces it; the one here is a
 * plain prefix-doubling sorter, checked against brute force in the tests.
 *
 * Storage note: every `I`-typed array (superblock offsets, SA samples, lookup
 * tables) is held as uint32_t; this restatement supports n < 2^32 only, and
 * index_width only drives the `n <= I::MAX` assertion (construction/mod.rs:34).
 */
#ifndef GDX_ORACLE_H
#define GDX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gdxo_index gdxo_index;

/* status of one query (the reference panics instead) */
enum {
    GDXO_OK = 0,
    GDXO_INVALID_SYMBOL = 1,      /* alphabet.rs:195-198 expect() panic          */
    GDXO_UNSEARCHABLE_IN_LOOKUP = 2 /* lookup_table.rs:154-158 aliasing / OOB     */
};

/* index_width: 32 = u32, -32 = i32, 64 = i64 (construction/mod.rs:156-252) */
gdxo_index *gdxo_build(const uint8_t *texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                       const uint8_t *io_to_dense /*256*/, int sigma, int n_searchable,
                       uint64_t sa_rate, int lookup_depth, int index_width);

/* Import of a BWT + sampled SA produced elsewhere (the GPU builder), so that the
 * CPU baseline can run on hg38-scale input.  The occurrence table is built here
 * from the BWT exactly as TextWithRankSupport::construct does. */
gdxo_index *gdxo_from_bwt(const uint8_t *bwt, uint64_t n, const uint32_t *sa_samples,
                          uint64_t sa_rate, const uint64_t *border_keys,
                          const uint64_t *border_vals, const uint64_t *sentinel_indices,
                          uint64_t n_texts, const uint8_t *io_to_dense, int sigma,
                          int n_searchable, int lookup_depth, int index_width, int n_threads);

/* bare occurrence table = TextWithRankSupport::construct(text, alphabet_size) */
gdxo_index *gdxo_table_only(const uint8_t *dense_text, uint64_t n, int sigma);

void gdxo_free(gdxo_index *ix);

uint64_t gdxo_n(const gdxo_index *ix);
uint64_t gdxo_num_texts(const gdxo_index *ix);
int gdxo_sigma(const gdxo_index *ix);
const uint64_t *gdxo_count(const gdxo_index *ix);              /* sigma+1 entries */
const uint8_t *gdxo_dense_text(const gdxo_index *ix);          /* n, or NULL */
const uint8_t *gdxo_bwt(const gdxo_index *ix);                 /* n, or NULL */
const uint32_t *gdxo_full_sa(const gdxo_index *ix);            /* n, or NULL */
const uint64_t *gdxo_blocks(const gdxo_index *ix, uint64_t *len);
const uint16_t *gdxo_block_offsets(const gdxo_index *ix, uint64_t *len);
const uint32_t *gdxo_superblock_offsets(const gdxo_index *ix, uint64_t *len);
const uint32_t *gdxo_sa_samples(const gdxo_index *ix, uint64_t *len);
const uint64_t *gdxo_border_keys(const gdxo_index *ix);        /* num_texts, sorted */
const uint64_t *gdxo_border_vals(const gdxo_index *ix);
const uint64_t *gdxo_sentinel_indices(const gdxo_index *ix);
/* lookup table of one depth as (start,end) u32 pairs; len = n_searchable^depth entries */
const uint32_t *gdxo_lookup_table(const gdxo_index *ix, int depth, uint64_t *len);

/* operator level: text_with_rank_support/mod.rs:106-133 */
int gdxo_rank(const gdxo_index *ix, int symbol, uint64_t idx, uint64_t *out);
int gdxo_symbol_at(const gdxo_index *ix, uint64_t idx, uint8_t *out);
/* batched rank, condensed.rs:137-287 (m <= 64 borders pairs) */
int gdxo_replace_many_interval_borders_with_ranks(const gdxo_index *ix, uint64_t *starts,
                                                  uint64_t *ends, const uint8_t *symbols,
                                                  uint64_t m);

/* lib.rs:217-235 single-query path; returns status */
int gdxo_cursor_for_query(const gdxo_index *ix, const uint8_t *q, uint64_t len, uint64_t *start,
                          uint64_t *end);
/* cursor.rs:34-51 */
int gdxo_extend_query_front(const gdxo_index *ix, uint64_t *start, uint64_t *end,
                            uint8_t io_symbol);
/* batch_computed_cursors.rs:36-172, N = 64.  Returns 0, or the status of the first
 * query that would have panicked (outputs are then unspecified, like a panic). */
int gdxo_cursors_for_many_queries(const gdxo_index *ix, const uint8_t *qbuf,
                                  const uint64_t *qoff, uint64_t nq, uint64_t *starts,
                                  uint64_t *ends, int n_threads);
/* single-query path looped, per-query status (used to cross-check the GPU status codes) */
void gdxo_cursors_single_path(const gdxo_index *ix, const uint8_t *qbuf, const uint64_t *qoff,
                              uint64_t nq, uint64_t *starts, uint64_t *ends, uint8_t *status,
                              int n_threads);

/* sampled_suffix_array.rs:110-138 + text_id_search_tree.rs:35-64, SA order */
void gdxo_locate_interval(const gdxo_index *ix, uint64_t start, uint64_t end,
                          uint64_t *text_ids, uint64_t *positions);
/* many intervals; hit_offsets[nq+1] is given (exclusive scan of end-start) */
void gdxo_locate_intervals(const gdxo_index *ix, const uint64_t *starts, const uint64_t *ends,
                           uint64_t nq, const uint64_t *hit_offsets, uint64_t *text_ids,
                           uint64_t *positions, int n_threads);
uint64_t gdxo_lookup_text_id(const gdxo_index *ix, uint64_t concatenated_text_index);
/* recover_range only: concatenated-text position per SA index */
void gdxo_recover_range(const gdxo_index *ix, uint64_t start, uint64_t end, uint64_t *out);

/* ---------------------------------------------------------------------------------------------
 * The four occurrence-table variants of the reference as stand-alone tables (lib.rs:102-113):
 * kind 0 = CondensedTextWithRankSupport (condensed.rs), kind 1 = FlatTextWithRankSupport (flat.rs);
 * block_bits 64 = Block64, 512 = Block512 (block.rs).  TextWithRankSupport::construct / rank / symbol_at. */
typedef struct gdxo_table gdxo_table;
gdxo_table *gdxo_table_construct(const uint8_t *dense_text, uint64_t n, int sigma, int kind, int block_bits);
void gdxo_table_free(gdxo_table *t);
int gdxo_table_rank(const gdxo_table *t, int symbol, uint64_t idx, uint64_t *out);
int gdxo_table_symbol_at(const gdxo_table *t, uint64_t idx, uint8_t *out);
/* interleaved_blocks as u64 words (Block512 = 8 words), block offsets (condensed only), superblock offsets */
const uint64_t *gdxo_table_blocks(const gdxo_table *t, uint64_t *n_words);
const uint16_t *gdxo_table_block_offsets(const gdxo_table *t, uint64_t *len);
const uint32_t *gdxo_table_superblock_offsets(const gdxo_table *t, uint64_t *len);

/* brute-force helpers used only to pin the oracle itself */
void gdxo_naive_suffix_array(const uint8_t *text, uint64_t n, uint32_t *sa);

#ifdef __cplusplus
}
#endif
#endif
