/*
 * gdx.h -- C ABI of the MI355X-native FM-index query engine (libgdx.so).
 *
 * Drop-in boundary for the query hot path of feldroop/genedex v0.2.2.  The reference is a
 * pure-Rust crate with no FFI of its own, so each entry point below names the public Rust
 * item (file:line relative to the reference root) it stands in for; INTEGRATION.md shows
 * the `extern "C"` block and the safe wrappers a maintainer would add on the Rust side.
 *
 * Conventions
 *  - every function returns a gdx_status (0 = ok); gdx_last_error() gives the text of the
 *    last failure on the calling thread.  The reference panics instead.
 *  - handles are immutable after construction: concurrent query calls on one handle are
 *    allowed (FmIndex is Send+Sync in the reference, all query methods take &self).
 *  - the caller owns every input and output buffer.  Plain entry points take HOST pointers;
 *    `_dev` entry points take DEVICE pointers plus a hipStream_t (passed as void*) and
 *    enqueue work without synchronising.
 *  - counts, positions and SA indices are uint64_t at the host ABI (usize in the reference).
 *    Inside, and in the `_dev` entry points, they are uint32_t for index storage types i32 and
 *    u32 (n <= 2^32-1) and for i64 when n fits.  index_width 64 with a collection that does NOT
 *    fit (the reference: IndexStorage for i64, construction/mod.rs:225-252) builds an index with
 *    64-bit rows on the reference's own arrays (wide.hip): gdx_index_build[_dev][_ex],
 *    gdx_index_info, gdx_count_many, gdx_cursors_for_many_queries, gdx_locate_many[_alloc],
 *    gdx_cursor_empty, gdx_cursor_extend_front_many, gdx_cursor_locate_many, gdx_rank_many,
 *    gdx_symbol_at_many and gdx_index_export_bwt serve it (same intervals, counts, hits and hit order
 *    as the reference's algorithm; lookup depth 0, alphabets of up to 7 symbols + sentinel; a plain
 *    engine, ~25x slower than the 32-bit one); every other call returns GDX_ERR_UNSUPPORTED on it.
 *    Collections that split at text borders are served faster by the partitioned index (gdx_parts_*).
 *  - a set of queries is one byte buffer `qbuf` plus `qoff[nq+1]` byte offsets
 *    (query i = qbuf[qoff[i] .. qoff[i+1])), IO symbols (ASCII), any mix of lengths.
 */
#ifndef GDX_H
#define GDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gdx_index gdx_index_t;

typedef enum {
    GDX_OK = 0,
    GDX_ERR_INVALID_ARGUMENT = 1,    /* reference: assert!/panic on the argument          */
    GDX_ERR_INVALID_TEXT_SYMBOL = 2, /* alphabet.rs:195-198 while encoding the texts      */
    GDX_ERR_TEXT_TOO_LONG = 3,       /* construction/mod.rs:34  n <= I::MAX               */
    GDX_ERR_DEVICE = 4,              /* HIP runtime failure, no GPU, out of device memory */
    GDX_ERR_CAPACITY = 5,            /* output buffer too small; required size reported   */
    GDX_ERR_QUERY_STATUS = 6,        /* >= 1 query has a non-zero status (see below); the
                                        outputs of all other queries are valid            */
    GDX_ERR_UNSUPPORTED = 7
} gdx_status;

/* per-query status byte; the reference panics for 1 and silently aliases / panics for 2 */
typedef enum {
    GDX_Q_OK = 0,
    GDX_Q_INVALID_SYMBOL = 1,        /* alphabet.rs:195-198: symbol not in the alphabet, reached
                                        while the interval was still non-empty              */
    GDX_Q_UNSEARCHABLE_IN_LOOKUP = 2 /* lookup_table.rs:154-158: a valid but non-searchable
                                        symbol (e.g. N) inside the lookup-table suffix      */
} gdx_query_status;

/* lib.rs:331-335 Hit { text_id: usize, position: usize } */
typedef struct {
    uint64_t text_id;
    uint64_t position;
} gdx_hit_t;

/* device-side hit record (index storage types i32/u32) */
typedef struct {
    uint32_t text_id;
    uint32_t position;
} gdx_hit32_t;

/* lib.rs:283-294 alphabet(), num_texts(), total_text_len() + config.rs:72-82 knobs */
typedef struct {
    uint64_t total_text_len; /* n, includes one sentinel per text */
    uint64_t num_texts;
    int32_t sigma;        /* num_dense_symbols, includes the sentinel */
    int32_t n_searchable; /* num_searchable_dense_symbols             */
    int32_t lookup_depth;
    int32_t index_width; /* 32 = u32, -32 = i32, 64 = i64 */
    uint64_t sa_rate;
    uint64_t device_bytes; /* HBM held by the handle */
    int32_t device_id;
    int32_t table_layout; /* 0 = 64-byte rank lines (sigma <= 8), 1 = generic planes */
} gdx_index_info_t;

const char *gdx_last_error(void);
int gdx_device_count(void);

/* ---------------------------------------------------------------------------------------
 * construction   (FmIndexConfig::construct_index config.rs:63-69 -> FmIndex::new lib.rs:118-142)
 * texts: concatenated IO symbols of all texts + text_offsets[n_texts+1]; io_to_dense = the
 * alphabet's 256-entry table (alphabet.rs:24-28, 0 = not in the alphabet).  The index is
 * built on the GPU `device_id` from scratch (own suffix sorter).                         */
int gdx_index_build(const uint8_t *texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                    const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate,
                    int lookup_depth, int index_width, int device_id, gdx_index_t **out);

/* same, the concatenated IO text already resides in device memory (text_offsets on host) */
int gdx_index_build_dev(const void *d_texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                        const uint8_t *io_to_dense, int sigma, int n_searchable,
                        uint64_t sa_rate, int lookup_depth, int index_width, int device_id,
                        gdx_index_t **out);

/* Import of an index a Rust host already owns, in the REFERENCE's logical layout
 * (condensed.rs:24-30 interleaved_blocks / Block64, lib.rs:95 count,
 * sampled_suffix_array.rs:18-23, text_id_search_tree.rs:8).  Block and superblock offsets
 * are recomputed on the device from the bit planes; lookup tables are refilled.  */
int gdx_index_from_parts(const uint64_t *count /*sigma+1*/, const uint64_t *interleaved_blocks,
                         uint64_t n /*text_len*/, const uint32_t *sa_samples, uint64_t sa_rate,
                         const uint64_t *border_keys, const uint64_t *border_vals,
                         const uint64_t *sentinel_indices, uint64_t n_texts,
                         const uint8_t *io_to_dense, int sigma, int n_searchable,
                         int lookup_depth, int index_width, int device_id, gdx_index_t **out);


/* ---- build options (FmIndexConfig is a value in the reference, config.rs:17-82; the knobs below are this
 * implementation's own and have no counterpart there).  Besides the reference's arrays an index may carry three
 * derived acceleration structures: pair lines (two LF steps per 128-byte fetch, 16 bits per
 * symbol), a jump table (8 to 32 LF steps of a narrow interval per fetch, 8 to 32 bytes per symbol) and a top table
 * (the first D symbols of a DNA query in one fetch, 8 * 4^D bytes).  Results are identical with any combination.
 * A field left at -1 / 0 takes its default; the GDX_* environment variables documented in DESIGN.md only override
 * fields left at their default (debugging aid).  Initialise with gdx_build_options_init(). */
/* THE DEFAULT SHAPE (round 6).  With jump_entry_bytes, full_suffix_array, text_units, seed_symbols and inverse_suffix_array all
 * left at -1 (and pair lines not switched off) on a DNA-like alphabet -- rank-line layout, dense symbols 1..4 searchable -- the
 * library builds ONE index that serves every call at its best measured speed: seed table (k from the text length, load 60 %) +
 * text units + full suffix array + inverse suffix array + pair lines + a top table of depth <= 14, no jump table (3.1 G symbols:
 * 104 GB).  Count / locate go through the seed table, exact intervals and cursors through seed entry / text / ISA with the pair
 * lines for the steps that empty an interval, locate has SA[row] one fetch away.  It needs 8.5 bytes per symbol + the seed
 * table inside the budget for auxiliary structures; where that does not fit, or when any of those fields is set, the fields mean
 * what they say and -1 stands for the tables of rounds 1-3 (32-byte jump entries, top table, shrunk to the budget).
 * gdx_index_aux_info (gdx_bench.h) bit 3 tells whether an index has the default shape. */
typedef struct {
    uint32_t struct_size;      /* sizeof(gdx_build_options_t), lets the struct grow compatibly         */
    int32_t pair_lines;        /* -1 default (on when sigma <= 8), 0 off, 1 on                         */
    int32_t jump_entry_bytes;  /* -1 default (none in the default shape, else 32), 0 no jump table, 8, 16 or 32 */
    int32_t top_table_depth;   /* -1 default (largest even D <= 16 with 4^D <= 2 n; at most 14 in the default shape), 0 none, 1..16 */
    uint64_t aux_budget_bytes; /* cap for jump + top table together; 0 = default: free device memory
                                  minus a reserve for query batches, at most half of the device memory.
                                  Tables that do not fit shrink (gdx_index_aux reports what was built). */
    int32_t full_suffix_array; /* -1 default (on in the default shape, else off), 0 off, 1: SA[row] of EVERY row as its own array (4 bytes per symbol); with
                                  32-byte jump entries the same values already sit inside the entries                 */
    int32_t text_units;        /* -1 default (on in the default shape, else off), 0 off, 1: the concatenated text itself, 4 bits per symbol (2-bit code + a
                                  "not A C G T" bit, 16 bytes per 32 symbols): once a search is down to a few rows, the rest
                                  of the query is compared with the text at SA[row] in one fetch per row instead of LF
                                  steps -- the low-memory alternative to the jump table (count / locate searches)      */
    int32_t seed_symbols;      /* -1 default (as 1 in the default shape, else off), 0 off; 1 = a SEED TABLE with k chosen from the text length (ceil(log4 n) + 8, at
                                  most 24), 8..24 = that k; implies text_units.  A bucketed hash table over the distinct k-mers of
                                  the text (16 bytes each over the load factor: 71 GB for 3.1 G symbols): count / locate searches
                                  fetch ONE 128-byte bucket for the last k symbols of a read, and when that k-mer occurs once in
                                  the text -- nearly every read of a text without repeats -- the entry also holds its position and
                                  the 32 symbols in front of it, so a read of up to k + 32 symbols is counted AND located with that
                                  single fetch; longer reads go on against the text units, k-mers on several rows hand over their
                                  suffix-array interval (results are the reference's either way); a k-mer on TWO to FOUR rows also gets a record
                                  (32 / 64 bytes) with its rows' positions and contexts when the budget has room, and decides its reads the same way.
                                  gdx_index_seed_info reports.
                                  The table has at least 2^(2k - 21) buckets of 128 bytes whatever the text (17 GB for k = 24, 1 GB
                                  for k = 22, 67 MB for k = 20): an explicit k whose table does not fit the budget for auxiliary
                                  structures is refused (GDX_ERR_INVALID_ARGUMENT); 1 picks a k the text fills                    */
    int32_t seed_load_percent; /* 0 default (60 in the default shape, else 70): slots of the seed table filled on average, 20..100 (fewer: more memory, fewer
                                  reads that need a second bucket)                                                                */
    int32_t inverse_suffix_array; /* -1 default (on in the default shape, else off), 0 off, 1: ISA[position] = row as its own array (4 bytes per symbol).  With it and a
                                  seed table, cursors_for_many_queries answers every read that occurs exactly once (its seed's
                                  entry, the text in front, then ONE fetch of the row) without LF steps; other reads take the usual
                                  route, so the intervals -- frozen empty ones included -- stay the reference's               */
    int32_t reference_table_layout; /* -1 / 0 default: this library's own occurrence table (rank lines for sigma <= 8, the
                                  reference's Condensed / Block64 arrays beyond); 1..4: the REFERENCE's table in the variant named
                                  -- 1 CondensedTextWithRankSupport<Block64>, 2 <Block512>, 3 FlatTextWithRankSupport<Block64>,
                                  4 <Block512> (lib.rs:102-113; bit for bit the arrays genedex builds: interleaved blocks, block
                                  offsets -- inside the blocks for the flat variants --, superblock offsets) and kernels that
                                  query it as it is, one lane per query: the reference's own speed / memory points
                                  (condensed.rs:291-341, flat.rs:221-266) instead of this library's.  No pair lines, jump, top
                                  or seed table on such an index (they belong to the rank-line layout).                  */
} gdx_build_options_t;
void gdx_build_options_init(gdx_build_options_t *opts);

/* gdx_index_build / gdx_index_build_dev / gdx_index_from_parts_ex / gdx_index_load with build options
 * (opts == NULL: defaults, i.e. exactly the plain calls) */
int gdx_index_build_ex(const uint8_t *texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                       const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate, int lookup_depth,
                       int index_width, int device_id, const gdx_build_options_t *opts, gdx_index_t **out);
int gdx_index_build_dev_ex(const void *d_texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                           const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate,
                           int lookup_depth, int index_width, int device_id, const gdx_build_options_t *opts,
                           gdx_index_t **out);
int gdx_index_from_parts_ex2(int table_kind, int block_bits, const uint64_t *count,
                             const uint64_t *interleaved_blocks, uint64_t n, const uint32_t *sa_samples,
                             uint64_t sa_rate, const uint64_t *border_keys, const uint64_t *border_vals,
                             const uint64_t *sentinel_indices, uint64_t n_texts, const uint8_t *io_to_dense,
                             int sigma, int n_searchable, int lookup_depth, int index_width, int device_id,
                             const gdx_build_options_t *opts, gdx_index_t **out);
int gdx_index_load_ex(const char *path, int device_id, const gdx_build_options_t *opts, gdx_index_t **out);

/* what an index carries beside the reference's arrays, and what was asked for before the budget was applied */
typedef struct {
    int32_t pair_lines;              /* 1 if present                                  */
    int32_t jump_entry_bytes;        /* 0 = no jump table                             */
    int32_t top_table_depth;         /* 0 = no top table                              */
    int32_t wanted_jump_entry_bytes; /* != jump_entry_bytes: the budget shrank it     */
    int32_t wanted_top_table_depth;
    int32_t wide_permille;           /* thousandths of the text positions whose top-table interval is wider than
                                        4 rows: how repetitive the text is as the search sees it (i.i.d. 3.1 G
                                        symbols: 3; genome-like: 95); above 50 the search parks stragglers
                                        by default (gdx_query_options_t.search_defer_after)                */
    uint64_t aux_bytes;              /* jump + top table                              */
    uint64_t aux_budget_bytes;       /* the budget that applied                       */
} gdx_index_aux_t;
int gdx_index_aux(const gdx_index_t *ix, gdx_index_aux_t *out);
/* the seed table of an index: out[0] = k (0 = none), [1] = buckets of 128 bytes, [2] = k-mers that occur once with 32 symbols
 * A C G T in front (answered by their entry alone), [3] = other k-mers (entries that hold a suffix-array interval), [4] = buckets
 * that turned an entry away (a miss there looks into the next bucket as well), [5] = largest displacement in buckets,
 * [6] = bytes (the table and the records below), [7] = tag bits */
int gdx_index_seed_info(const gdx_index_t *ix, uint64_t out[8]);
/* the records beside the seed table: a k-mer on two to four rows, each with 32 symbols A C G T in front, has a record with the
 * positions and the contexts of its rows (32 bytes for two rows, 64 for three and four) when the budget has room for them -- a
 * count / locate read from such a repeat is decided by that record.  out[0] = records of two-copy repeats, [1] = of three- and
 * four-copy repeats, [2] = their bytes (part of gdx_index_seed_info's [6]), [3] = 0 */
int gdx_index_seed_records(const gdx_index_t *ix, uint64_t out[4]);

/* ---- query options: which kernel variant the query calls on this handle use.  Every combination returns
 * identical results (the parity tests run them all); the defaults are the measured fastest.  The setting is
 * per handle, takes effect for calls that start afterwards and may be changed while other threads query. */
typedef struct {
    uint32_t struct_size;    /* sizeof(gdx_query_options_t)                                                   */
    int32_t search_kernel;   /* -1 default (2 when pair lines exist), 0 four lanes per query on rank lines,
                                1 one lane per query on rank lines, 2 pair lines + jump / top tables          */
    int32_t search_lanes;    /* 0 default (4), 4 or 8 lanes per query in the pair-line kernel                 */
    int32_t load_policy;     /* -1 default (0 plain loads), 1 = sc1 (no L1 allocation)                        */
    int32_t length_schedule; /* -1 default (1: a block orders its queries by length when they differ), 0 off  */
    int32_t locate_kernel;   /* ignored since round 5 (-1 .. 2 accepted): the chunk kernels are the only ones   */
    int32_t locate_jump_walk; /* -1 default (1: the locate walk goes through the jump table), 0 rank lines only */
    int32_t search_defer_after; /* a query still unfinished this many load rounds after the allowance of a query
                                that jumps is parked and finished in its block's straggler pass, where all lanes work
                                on such queries (reads from repeats); 0 = never park; -1 default: 3 on repetitive
                                texts (gdx_index_aux_t.wide_permille > 50), else 0 (the bookkeeping is not free)    */
    int32_t search_fast;     /* 1: count / locate searches first run a slim kernel that only knows the top table,
                                jumps and tails decided from jump entries, and hand what it cannot finish to the
                                general kernel; 2: the same with jumps over intervals of up to 16 rows (reads from
                                repeat families) instead of 4; 0 = general kernel only; -1 default, by
                                gdx_index_aux_t.wide_permille: > 500 (a top table that is shallow for the text) 0,
                                > 20 (a repetitive text) 2, else 1                                              */
    int32_t search_exact;    /* -1 default (1): cursors_for_many_queries and the cursor extension calls first run a slim
                                kernel for clean input (symbols A C G T only: top table, whole jump levels, exact pair-line
                                steps) and hand what it cannot finish to the general kernel; 0 = general kernel only    */
    uint32_t max_hits_per_query; /* gdx_locate_many / gdx_locate_many_alloc / gdx_multi_locate_many_alloc: at most this many
                                hits per query, the first ones in suffix-array order -- locate(q).take(k) of the reference's
                                lazy iterator (lib.rs:187-197): one poly-A read on a genome would otherwise materialise
                                gigabytes of hits.  hit_offsets then counts the hits RETURNED.  0 = all (default).
                                gdx_parts_locate_many_alloc: at most this many per query over ALL parts (the first k in
                                part order, suffix-array order inside a part)                                        */
    int32_t search_seed;     /* -1 default (1): count / locate searches start from the seed table when the index has one
                                (gdx_build_options_t.seed_symbols); 0 = as if it had none                                    */
} gdx_query_options_t;
void gdx_query_options_init(gdx_query_options_t *opts);
int gdx_index_set_query_options(gdx_index_t *ix, const gdx_query_options_t *opts);
int gdx_index_get_query_options(const gdx_index_t *ix, gdx_query_options_t *out);

/* Persistence (FmIndex::save_to_file / load_from_file, lib.rs:296-327).  The reference's wire format belongs to
 * the un-vendored `savefile` crate and no test of the reference inspects it, so this is an own format: a small
 * header followed by the index in the REFERENCE's logical layout (count, Condensed/Block64 bit planes, sampled
 * suffix array, text borders, sentinel positions); loading goes through the same path as gdx_index_from_parts. */
int gdx_index_save(const gdx_index_t *ix, const char *path);
int gdx_index_load(const char *path, int device_id, gdx_index_t **out);

void gdx_index_free(gdx_index_t *ix);
int gdx_index_info(const gdx_index_t *ix, gdx_index_info_t *out);

/* exports in the reference's logical layout (host buffers), for interchange and parity checks */
int gdx_index_export_count(const gdx_index_t *ix, uint64_t *count /*sigma+1*/);
int gdx_index_export_bwt(const gdx_index_t *ix, uint8_t *bwt /*n*/);
int gdx_index_export_sa_samples(const gdx_index_t *ix, uint32_t *samples /*ceil(n/rate)*/);
int gdx_index_export_borders(const gdx_index_t *ix, uint64_t *keys, uint64_t *vals /*num_texts*/);
int gdx_index_export_sentinel_indices(const gdx_index_t *ix, uint64_t *out /*num_texts*/);
int gdx_index_export_lookup_table(const gdx_index_t *ix, int depth, uint32_t *pairs /*2*k^depth*/);
/* condensed.rs:24-30: blocks ceil((n+1)/64)*nbits u64, block offsets ceil((n+1)/64)*sigma u16,
 * superblock offsets ceil((n+1)/65536)*sigma u32 */
int gdx_index_export_condensed_table(const gdx_index_t *ix, uint64_t *interleaved_blocks,
                                     uint16_t *interleaved_block_offsets,
                                     uint32_t *interleaved_superblock_offsets);
/* an index built with gdx_build_options_t.reference_table_layout: its occurrence table as it sits in HBM -- the
 * reference's interleaved blocks of that variant (flat: block offsets inside the blocks, flat.rs:30-52) and u32 superblock
 * offsets.  NULL buffers: only the sizes (64-bit words / offsets) are returned. */
int gdx_index_export_reference_table(const gdx_index_t *ix, uint64_t *interleaved_blocks, uint64_t capacity_words,
                                     uint64_t *out_n_words, uint32_t *interleaved_superblock_offsets, uint64_t capacity_offsets,
                                     uint64_t *out_n_offsets);

/* ---------------------------------------------------------------------------------------
 * operator level   (TextWithRankSupport, text_with_rank_support/mod.rs:88-133)           */
/* rank (mod.rs:106-110): out[i] = #symbols[i] in bwt[0..idx[i]).  INVALID_ARGUMENT if any
 * symbol >= sigma or idx > n (the reference asserts). */
int gdx_rank_many(const gdx_index_t *ix, const uint8_t *symbols, const uint64_t *idx, uint64_t m,
                  uint64_t *out);
/* symbol_at (condensed.rs:343-362); INVALID_ARGUMENT if any idx >= n */
int gdx_symbol_at_many(const gdx_index_t *ix, const uint64_t *idx, uint64_t m, uint8_t *out);

/* ---------------------------------------------------------------------------------------
 * queries                                                                                 */
/* FmIndex::count_many lib.rs:155-161 (order preserving) */
int gdx_count_many(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                   uint64_t *out_counts, uint8_t *out_status /*nq or NULL*/);
/* FmIndex::cursors_for_many_queries lib.rs:241-246 / BatchComputedCursors: the half-open
 * SA interval per query, bit-identical to the reference also for empty results */
int gdx_cursors_for_many_queries(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff,
                                 uint64_t nq, uint64_t *out_start, uint64_t *out_end,
                                 uint8_t *out_status);
/* FmIndex::locate_many lib.rs:179-185.  Hits of query i are hits[out_hit_offsets[i] ..
 * out_hit_offsets[i+1]) in suffix-array order (lib.rs:187-197).  *out_total is always set;
 * if hits == NULL or hits_capacity < total the call returns GDX_ERR_CAPACITY after filling
 * out_hit_offsets (sizing call).  The locate calls on host pointers run the batch in chunks, each ONE fused step on the device
 * whose results cross PCIe packed (the found-bitmap wire of gdx_wire_pack_dev: 3.7-4.6 bytes per read) and are expanded into the
 * caller's arrays by host threads; a chunk may hold at most 2^32 - 2 hits (GDX_ERR_CAPACITY names the remedy).  Environment
 * GDX_HOST_RESULTS=dma: search, total, locate as separate launches and offsets + hits copied out by the device (rounds 1-4;
 * also what collections of more than 256 texts and hosts with fewer than four worker threads get). */
int gdx_locate_many(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                    uint64_t *out_hit_offsets /*nq+1*/, gdx_hit_t *hits, uint64_t hits_capacity,
                    uint64_t *out_total, uint8_t *out_status);

/* The same in ONE pass: the library allocates the hit array (*out_hits, `*out_total` entries; release it with
 * gdx_free_hits).  gdx_locate_many with a caller-owned buffer needs a sizing call first, i.e. searches twice. */
int gdx_locate_many_alloc(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                          uint64_t *out_hit_offsets /*nq+1*/, gdx_hit_t **out_hits, uint64_t *out_total,
                          uint8_t *out_status);
void gdx_free_hits(gdx_hit_t *hits);
/* (the library keeps ONE released array -- the largest it has seen -- for the next gdx_locate_many_alloc, whose pages are
 * then already there: a caller that locates batch after batch does not pay the first touch of gigabytes per call) */
/* All three host-pointer query calls above run as a pipeline over chunks of the batch (copy-in, kernels and copy-out
 * of neighbouring chunks overlap, pinned staging filled by a few host threads; GDX_HOST_THREADS overrides their
 * number), so a call costs about max(PCIe in, PCIe out, kernels) rather than their sum. */

/* ---------------------------------------------------------------------------------------
 * batched cursor API   (Cursor, cursor.rs:16-73; the reference has only the scalar form,
 * ROADMAP.md:33 lists the batched one as future work)                                     */
/* FmIndex::cursor_empty lib.rs:202-210: [0, n) */
int gdx_cursor_empty(const gdx_index_t *ix, uint64_t *start, uint64_t *end);
/* Cursor::extend_query_front cursor.rs:34-51 for m independent cursors, in place; a cursor
 * whose interval is already empty is left untouched */
int gdx_cursor_extend_front_many(const gdx_index_t *ix, uint64_t *start, uint64_t *end,
                                 const uint8_t *io_symbols, uint64_t m, uint8_t *out_status);
/* Cursor::locate cursor.rs:71-73 for m cursors; same output convention as gdx_locate_many */
int gdx_cursor_locate_many(const gdx_index_t *ix, const uint64_t *start, const uint64_t *end,
                           uint64_t m, uint64_t *out_hit_offsets, gdx_hit_t *hits,
                           uint64_t hits_capacity, uint64_t *out_total);

/* ---------------------------------------------------------------------------------------
 * device-resident entry points: all pointers are DEVICE pointers on the handle's GPU, work
 * is enqueued on `stream` (hipStream_t) and the call returns without synchronising.
 * d_qbuf must be 8-byte aligned and its allocation padded to a multiple of 8 bytes.
 * The calls make the handle's device current for their duration and restore the caller's.   */
int gdx_cursor_extend_front_many_dev(const gdx_index_t *ix, void *d_start /*u32*/, void *d_end /*u32*/,
                                     const void *d_io_symbols, uint64_t m, void *d_out_status,
                                     void *stream);
/* locate for m intervals whose offsets were produced by gdx_hit_offsets_dev; total =
 * d_hit_offsets[m] (the caller reads it back to size d_hits and d_workspace:
 * gdx_locate_workspace_bytes(total)). */
uint64_t gdx_locate_workspace_bytes(uint64_t total_hits);

/* The "found bitmap" form of a located shard on its way to another device (the multi-GPU gather): one
 * BIT per read -- set: the read's compact result is a position, i.e. it has exactly one hit -- in d_bitmap (bit q & 7 of byte
 * q >> 3; gdx_wire_bitmap_bytes(nq) bytes), the text positions of those reads back to back in read order in d_found_pos
 * (u32[found_capacity]), the number of found reads before every tile of 2048 reads in d_tile_found (u32[nq / 2048 + 2], the
 * last entry = all of them), and the EXCEPTIONS -- the reads whose compact result says "see the record" -- in read order:
 * d_exc_queries / d_exc_counts (u32[exc_capacity]: read number, number of hits) and their hits d_exc_text_ids (u8) /
 * d_exc_positions (i32) [exc_hits_capacity], taken from the shard's hit offsets (u32 or u64: offsets_width) and hits
 * (gdx_hit32_t).  d_meta (u32[4]) = {exceptions, their hits, found reads, 0}: the true numbers -- what exceeds a capacity is
 * dropped, and the receiver sees it there.  3.73 bytes per read where nine reads in ten are found, against 4 for the compact
 * words themselves (a position needs its 32 bits, a miss does not).  Collections of at most 256 texts.  d_workspace:
 * gdx_wire_pack_workspace_bytes(nq) bytes.  Three launches: tile counts, their scan, the pack. */
uint64_t gdx_wire_bitmap_bytes(uint64_t nq);
uint64_t gdx_wire_pack_workspace_bytes(uint64_t nq);
int gdx_wire_pack_dev(const gdx_index_t *ix, const void *d_compact, const void *d_hit_offsets, uint32_t offsets_width,
                      const void *d_hits, uint64_t nq, void *d_bitmap, void *d_tile_found, void *d_found_pos, uint64_t found_capacity,
                      void *d_exc_queries, void *d_exc_counts, uint64_t exc_capacity, void *d_exc_text_ids, void *d_exc_positions,
                      uint64_t exc_hits_capacity, void *d_meta, void *d_workspace, void *stream);
/* The receiver's side: a shard in that form -> per read one text id byte and one int32, the position in that text of the
 * read's only hit (lib.rs:331-335 Hit { text_id, position }), -1 = no occurrence, -2 = an exception (the reads d_exc_queries
 * lists, ascending; d_meta[0] of them, at most exc_capacity) -- what gdx_compact_split_hits_dev produces from compact words.
 * d_out_text_ids 8-byte, d_out_positions 16-byte aligned. */
int gdx_wire_split_dev(const gdx_index_t *ix, const void *d_bitmap, const void *d_tile_found, const void *d_found_pos,
                       uint64_t found_capacity, uint64_t nq, const void *d_exc_queries, const void *d_meta, uint64_t exc_capacity,
                       void *d_out_text_ids, void *d_out_positions, void *stream);

/* ---- packed queries ("allow 2-bit host packing as an optional input format") ------------------------
 * Four symbols per byte instead of one: symbol j of the buffer sits in bits 2 (j & 3) .. 2 (j & 3) + 1 of byte j >> 2
 * and holds (dense code - 1) of one of the dense symbols 1..4 (A, C, G, T of the DNA alphabets); offsets count SYMBOLS
 * (the very qoff array of the ASCII form when the packed buffer was made from the whole ASCII buffer).  A query with
 * any other symbol (N, IUPAC codes, bytes outside the alphabet) cannot be expressed: it is an EXCEPTION, its packed
 * symbols are 0 and its results from the packed calls are meaningless -- run the exceptions through the ASCII calls.
 * Packed input quarters the PCIe traffic of the host calls and the query-byte DRAM traffic of the search and needs
 * no alphabet translation; results are identical.  Needs the rank-line layout (sigma <= 8, dense symbols 1..4 searchable:
 * the DNA alphabets) -- with or without pair lines, seed table, text units.
 * A packed buffer for n symbols takes gdx_packed_bytes(n) bytes (padded: the kernels read whole 16-bit units), must
 * be 2-byte aligned (device) and is made by
 *   gdx_pack_queries      on the host, by a few threads; out_exceptions receives the sorted indices of the exception
 *                         queries (GDX_ERR_CAPACITY and the needed count in *out_n_exceptions if there are more)
 *   gdx_pack_queries_dev  on the device from a device-resident ASCII buffer; *d_bad_symbols (u64, zeroed by the
 *                         caller, may be NULL) counts symbols outside 1..4, d_bad_flags (u8 per 64 symbols, zeroed by
 *                         the caller, may be NULL) marks where they are */
uint64_t gdx_packed_bytes(uint64_t n_symbols);
int gdx_pack_queries(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint8_t *out_packed,
                     uint64_t *out_exceptions, uint64_t exceptions_capacity, uint64_t *out_n_exceptions);
int gdx_pack_queries_dev(const gdx_index_t *ix, const void *d_qbuf, uint64_t n_symbols, void *d_packed,
                         void *d_bad_flags, void *d_bad_symbols, void *stream);
/* gdx_pack_queries with the alphabet's 256-entry table instead of an index (host only, no device needed): for a reader that
 * packs what it parses -- gdx_fastx_next_batch -> gdx_pack_queries_table -> gdx_*_layout -- before or without an index */
int gdx_pack_queries_table(const uint8_t *io_to_dense, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                           uint8_t *out_packed, uint64_t *out_exceptions, uint64_t exceptions_capacity,
                           uint64_t *out_n_exceptions);

/* ---- query layouts: how a batch of queries lies in its buffer ---------------------------------------------------------
 * The reference's batch calls take any iterator of byte slices (lib.rs:155-161, 179-185, 241-246); a device has one buffer.
 * Beside the plain form -- IO symbols, one byte each, with an offsets array -- a batch may be
 *   packed   2-bit codes, four symbols per byte ("packed queries" above; ROADMAP.md:35-37), and / or
 *   uniform  every query has the same number of symbols and query i starts at symbol i * uniform_len: no offsets array at
 *            all (d_qoff may be NULL) -- a batch of sequencer reads.  The kernels then compute where a read lies instead of
 *            loading 8 bytes of offsets per read.
 * A len-50 read costs the search 58 bytes of query traffic in the plain form, 20.5 packed, 12.5 packed + uniform, next to
 * the one 128-byte bucket of the seed table it looks at.  Results are identical in every form (exceptions of the packed
 * form as above).  On every index of the rank-line layout (sigma <= 8); the seed-table kernels, the text-unit kernels and
 * the rank-line kernel read all forms natively, the pair-line kernels read offsets (a uniform batch gets them written into
 * scratch, once per call).  Initialise with gdx_query_layout_init(); layout == NULL means the plain form. */
typedef struct {
    uint32_t struct_size;  /* sizeof(gdx_query_layout_t) */
    int32_t packed;        /* 0: IO symbols (d_qbuf 8-byte aligned); 1: 2-bit codes (d_qbuf 2-byte aligned) */
    uint64_t uniform_len;  /* 0: offsets in d_qoff (u64[nq + 1], counting symbols); 1 .. 2^21 - 1: uniform batch */
} gdx_query_layout_t;
void gdx_query_layout_init(gdx_query_layout_t *layout);
/* The WHOLE count + locate step of a batch in one call and without a host round trip (FmIndex::locate_many, lib.rs:179-185, on
 * device-resident reads): search, hit totals, hit offsets and hits are enqueued behind each other on `stream`; the number of
 * hits stays on the device.  The caller offers a hit buffer of hits_capacity entries (gdx_hit32_t) -- sized from what it knows
 * of its batches: one hit per read and a margin, or the largest total seen so far -- and a workspace of
 * gdx_locate_workspace_bytes(hits_capacity) bytes.  d_totals (u64[2], device) = {all hit slots, those behind "see the record"}
 * is valid once the stream has passed the call: the caller reads it when it reads the results.  totals[0] <= hits_capacity:
 * d_hit_offsets and d_hits are complete.  totals[0] > hits_capacity: the offsets are right (u64; the u32 form wraps beyond
 * 2^32), hits at and beyond the capacity were not stored -- the records, compact results and tile bases are intact, so
 * gdx_locate_many_offsets_hits_compact_dev with a buffer of totals[0] entries finishes the step.  offsets_width 32: d_hit_offsets
 * is u32[nq + 1] (hits_capacity < 2^32), 64: u64[nq + 1].  d_compact as in gdx_locate_many_search_compact_layout_dev (NULL on an
 * index without seed table: records only).  A step of 12.5 M reads (what a rank of eight runs of a 100 M batch) is a dozen
 * launches; the three calls around a round trip were two dozen and 45 us of waiting in 0.57 ms.  event_after_search
 * (hipEvent_t or NULL) is recorded on the stream between the search (incl. its hit totals) and the offsets + hits, for
 * callers that time the two halves. */
int gdx_locate_many_step_compact_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                            const gdx_query_layout_t *layout, uint32_t max_hits, void *d_records, void *d_compact,
                                            void *d_scan_workspace, void *d_totals, void *d_hit_offsets, uint32_t offsets_width,
                                            void *d_hits, uint64_t hits_capacity, void *d_workspace, void *event_after_search,
                                            void *stream);
int gdx_count_many_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                              const gdx_query_layout_t *layout, void *d_out_counts, void *d_out_status, void *stream);
int gdx_cursors_for_many_queries_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                            const gdx_query_layout_t *layout, void *d_out_start, void *d_out_end,
                                            void *d_out_status, void *stream);

/* gdx_locate_many_alloc_layout with NARROW results in pinned host memory the library owns (FmIndex::locate_many, lib.rs:179-185;
 * Hit lib.rs:331-335 as two u32): hit_offsets is u32[nq + 1], hits gdx_hit32_t[total_hits] -- 8 + 4 bytes per result instead of
 * 16 + 8 (the wide call spends its time widening once the reads come as 2-bit codes: 12.5 bytes in, 23 bytes out per read).
 * The link bounds this call and its two directions share one rate, so a chunk's results cross it packed -- the found-bitmap
 * wire of gdx_wire_pack_dev below: a bit per read, position (+ a text id byte) per read with one hit, the others' hits, 3.7-4.6
 * bytes per read instead of 12.2 -- and host threads write offsets and hits into the arrays (AVX2, streaming stores; status bytes
 * stay on the device unless a read of the chunk has one; GDX_HOST_RESULTS=dma as above: the device writes offsets and hits
 * itself).  Fewer than 2^32 hits in all,
 * else GDX_ERR_CAPACITY (the wide call has no such limit).  Release the arrays with gdx_free_hits32: the library keeps one pair
 * for the caller's next batch (pinning memory costs about a millisecond per 10 MB); gdx_release_cached_hits() frees what the
 * library holds back, this pair and the array of gdx_free_hits.  layout may be NULL (IO symbols + offsets).  When qbuf itself
 * is pinned host memory (hipHostMalloc / hipHostRegister), every host-pointer call copies from it directly, without staging. */
typedef struct {
    uint32_t *hit_offsets;
    gdx_hit32_t *hits;
    uint64_t total_hits, nq;
    uint64_t reserved[2];
} gdx_hits32_t;
int gdx_locate_many_alloc_layout32(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                   const gdx_query_layout_t *layout, gdx_hits32_t *out_results, uint8_t *out_status);
void gdx_free_hits32(gdx_hits32_t *results);
void gdx_release_cached_hits(void);

/* The host-pointer calls on a batch in a layout: gdx_count_many / gdx_cursors_for_many_queries / gdx_locate_many_alloc
 * (FmIndex::count_many lib.rs:155, cursors_for_many_queries :241, locate_many :179) through the same chunked pipeline.  A
 * packed + uniform batch of len-50 reads moves 12.5 bytes per read over PCIe instead of 58, and no offsets are staged. */
int gdx_count_many_layout(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                          const gdx_query_layout_t *layout, uint64_t *out_counts, uint8_t *out_status);
int gdx_cursors_for_many_queries_layout(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                        const gdx_query_layout_t *layout, uint64_t *out_start, uint64_t *out_end,
                                        uint8_t *out_status);
int gdx_locate_many_alloc_layout(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                 const gdx_query_layout_t *layout, uint64_t *out_hit_offsets, gdx_hit_t **out_hits,
                                 uint64_t *out_total, uint8_t *out_status);

/* ---- batched cursor extension by strings (Cursor::extend_query_front, cursor.rs:34-51, applied to every symbol of
 * a string from its last to its first; ROADMAP.md:33 "API to use batched search with cursors") --------------------
 * Cursor i is extended by string i = d_qbuf[d_qbeg[i] .. d_qend[i]) (for a plain offsets array pass d_qoff and
 * d_qoff + 1; two arrays allow chunk views into longer queries without copying bytes).  d_start / d_end are
 * in / out.  A cursor whose interval is (or becomes) empty is left as it is and the rest of its string is not
 * looked at (the reference would still translate those symbols and panic on an invalid one, cursor.rs:35-38).
 * An invalid symbol stops the cursor where it stands and sets d_status[i] = GDX_Q_INVALID_SYMBOL; a stopped cursor
 * ignores later calls (d_status is in / out when given: zero it together with gdx_cursor_empty).
 * Active lists (optional, device-side compaction so that a caller feeding long queries in chunks only touches live
 * cursors): d_active_in = indices (u32) of the cursors to extend, *d_n_active_in (u32 in device memory) of them,
 * NULL = all m; the cursors that are still non-empty and not stopped afterwards are appended to d_active_out
 * (u32[m], any order) and counted in *d_n_active_out (u32 in device memory, zeroed by this call).  m remains the
 * upper bound the launch is sized for.  One launch advances a cursor by its whole string through the pair lines
 * and the jump table (up to 32 LF steps per fetch), not one LF step per launch. */
int gdx_cursor_extend_front_strings_dev(const gdx_index_t *ix, void *d_start /*u32*/, void *d_end /*u32*/,
                                        const void *d_qbuf, const void *d_qbeg /*u64*/, const void *d_qend /*u64*/,
                                        uint64_t m, void *d_status /*u8 or NULL*/, const void *d_active_in,
                                        const void *d_n_active_in, void *d_active_out, void *d_n_active_out,
                                        void *stream);
/* The same for callers that feed whole queries a chunk at a time (what a read mapper does): call k = 0, 1, 2, ...
 * extends cursor i by chunk k of query i, counted from the query's END in chunks of chunk_symbols symbols -- the
 * symbols [max(qoff[i], qoff[i+1] - (k + 1) c), qoff[i+1] - k c), nothing when the query is shorter than k c.  No
 * per-call offset arrays, and the live list is sharper: a cursor is appended to d_active_out only if it is
 * non-empty, not stopped AND its query has symbols left of this chunk, so a call never visits a finished query. */
int gdx_cursor_extend_front_chunk_dev(const gdx_index_t *ix, void *d_start /*u32*/, void *d_end /*u32*/,
                                      const void *d_qbuf, const void *d_qoff /*u64[m+1]*/, uint64_t m,
                                      uint32_t chunk_symbols, uint32_t chunk_index, void *d_status,
                                      const void *d_active_in, const void *d_n_active_in, void *d_active_out,
                                      void *d_n_active_out, void *stream);
/* host form: strings = qbuf + qoff[m+1]; start / end in / out; status (in / out, may be NULL) */
int gdx_cursor_extend_front_strings(const gdx_index_t *ix, uint64_t *start, uint64_t *end, const uint8_t *qbuf,
                                    const uint64_t *qoff, uint64_t m, uint8_t *status);

/* Unlike gdx_rank_many, which returns GDX_ERR_INVALID_ARGUMENT, the device form cannot report an argument
 * error without synchronising: an entry with symbol >= sigma or idx > n yields d_out[i] = 0 and, if d_error
 * (u32, may be NULL) is given, *d_error is set to 1 (the caller zeroes it beforehand). */
int gdx_rank_many_dev(const gdx_index_t *ix, const void *d_symbols, const void *d_idx /*u32*/, uint64_t m,
                      void *d_out /*u32*/, void *d_error /*u32 or NULL*/, void *stream);

/* ---- several GPUs of one node behind one handle ------------------------------------------
 * For a host that is one process (the Rust caller), not a set of torch.distributed ranks: the index is replicated,
 * one replica per device; a query call cuts the batch into contiguous shards (shard r = queries [nq r / g,
 * nq (r + 1) / g)), every replica runs the host-pointer pipeline on its shard over its own PCIe link from its own
 * thread and writes into the caller's arrays at the shard's offset.  The devices exchange nothing (the results live
 * in host memory), and the output is bit for bit the one-GPU output.  gdx_multi_from_indexes CONSUMES the replica
 * handles it is given (they are set to NULL); all must come from the same texts and configuration. */
typedef struct gdx_multi gdx_multi_t;
int gdx_multi_build(const uint8_t *texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                    const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate, int lookup_depth,
                    int index_width, const int *device_ids, int n_devices, const gdx_build_options_t *opts,
                    gdx_multi_t **out);
void gdx_multi_free(gdx_multi_t *m);
int gdx_multi_replicas(const gdx_multi_t *m);
/* gdx_index_set_query_options on every replica (e.g. max_hits_per_query for gdx_multi_locate_many_alloc) */
int gdx_multi_set_query_options(gdx_multi_t *m, const gdx_query_options_t *opts);
int gdx_multi_count_many(const gdx_multi_t *m, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                         uint64_t *out_counts, uint8_t *out_status);
int gdx_multi_cursors_for_many_queries(const gdx_multi_t *m, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                       uint64_t *out_start, uint64_t *out_end, uint8_t *out_status);
/* ---- collections beyond 2^32 - 1 symbols (the reference: IndexStorage = i64, construction/mod.rs:225-252): a
 * PARTITIONED index.  Rows and text positions are 32-bit in every table and kernel of this library, so a collection that
 * does not fit one index is cut at text borders into parts of at most max_part_symbols (0 = 2^32 - 1; sentinels count),
 * every part gets its own index on the device, and a query runs against each: count = the sum of the parts' counts,
 * locate = the parts' hits part after part with global text ids.  Counts and hit sets are the reference's (an occurrence
 * lies inside one text); the single suffix-array interval of the whole collection (cursors) is not available, and a
 * query's hits come in suffix-array order per part (the reference leaves the order of locate() open, lib.rs:163).  A
 * single text that does not fit a part is refused (GDX_ERR_TEXT_TOO_LONG).  texts_on_device != 0: texts_buf is a device
 * pointer (text_offsets stays on the host). */
typedef struct gdx_parts gdx_parts_t;
int gdx_parts_build(const void *texts_buf, int texts_on_device, const uint64_t *text_offsets, uint64_t n_texts,
                    const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate, int lookup_depth,
                    int device_id, uint64_t max_part_symbols, const gdx_build_options_t *opts, gdx_parts_t **out);
void gdx_parts_free(gdx_parts_t *p);
/* out[0] = number of parts, out[1] = total symbols incl. sentinels, out[2] = number of texts, out[3] = device bytes */
int gdx_parts_info(const gdx_parts_t *p, uint64_t out[4]);
int gdx_parts_set_query_options(gdx_parts_t *p, const gdx_query_options_t *opts);
int gdx_parts_count_many(const gdx_parts_t *p, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                         uint64_t *out_counts, uint8_t *out_status);
int gdx_parts_locate_many_alloc(const gdx_parts_t *p, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                uint64_t *out_hit_offsets, gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status);

/* Device-resident form: the queries of shard r already sit in the HBM of replica r's device; every
 * replica runs search -> scan -> locate on its shard, then the per-query counts (u32) and the hits (gdx_hit32_t) of all
 * shards are gathered into buffers on the ROOT replica's device by RCCL point-to-point transfers over xGMI (ncclSend /
 * ncclRecv of every shard inside one group: one exchange step, no collective, no reduction), and the root scans the
 * counts into global hit offsets.  Order preserving: shard r's queries and hits follow shard r - 1's, so the output is
 * the one-GPU output of the concatenated batch.  The result buffers belong to the handle and stay valid until the next
 * gather call on it or gdx_multi_free.  Replicas must sit on distinct devices (RCCL, loaded on first use), or all on
 * one device (plain device copies; what a single-GPU box can test).  The shards' buffers must be complete before the call
 * (the call works on streams of its own: synchronise the stream that filled them, or the device); the call returns when
 * the gathered results are in place.  d_qbuf of every non-empty shard must be 8-byte aligned. */
typedef struct {
    const void *d_qbuf; /* query bytes, on the device of replica r (8-byte aligned)  */
    const void *d_qoff; /* u64[nq + 1] offsets into d_qbuf, on the same device */
    uint64_t nq;
} gdx_device_shard_t;
typedef struct {
    void *d_counts;      /* u32[nq]         occurrences of every query, shard after shard */
    void *d_hit_offsets; /* u64[nq + 1]     hits of query i at [off[i], off[i + 1])       */
    void *d_hits;        /* gdx_hit32_t[total_hits], suffix-array order within a query    */
    void *d_status;      /* u8[nq]                                                        */
    uint64_t nq, total_hits;
    int32_t device_id;   /* the root replica's device, where all of the above live        */
    int32_t used_rccl;   /* 1: the transfers went through ncclSend / ncclRecv             */
} gdx_gathered_t;
int gdx_multi_locate_many_gather_dev(gdx_multi_t *m, const gdx_device_shard_t *shards, int n_shards, int root,
                                     gdx_gathered_t *out);
int gdx_multi_locate_many_alloc(const gdx_multi_t *m, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                uint64_t *out_hit_offsets, gdx_hit_t **out_hits, uint64_t *out_total,
                                uint8_t *out_status);

/* ---- query / text ingestion (host only) ---------------------------------------------------------------------
 * Streaming FASTA / FASTQ reader that fills the layout the calls above take: sequences appended to qbuf, offsets
 * to qoff[0 .. n] (qoff[0] = 0).  The reference leaves reading to its callers (ROADMAP.md:35-37 notes it can cost
 * more than searching).  FASTA: '>' headers, sequences over any number of lines; FASTQ: '@' header, sequence
 * lines, '+' line, as many quality characters as symbols; '\r' dropped; bytes copied as they are (the alphabet
 * decides what is valid).  gdx_fastx_next_batch reads up to max_records records that fit into qbuf_capacity
 * bytes, *n_out = 0 at the end of the file; a record larger than the whole buffer is GDX_ERR_CAPACITY, and the reader stays
 * at that record: call again with a larger buffer (genedex_amd.fastx.read_sequences doubles it). */
typedef struct gdx_fastx gdx_fastx_t;
int gdx_fastx_open(const char *path, gdx_fastx_t **out);
int gdx_fastx_next_batch(gdx_fastx_t *reader, uint8_t *qbuf, uint64_t qbuf_capacity, uint64_t *qoff,
                         uint64_t max_records, uint64_t *n_out);
/* the same; *out_uniform_len (may be NULL) = the common length of the batch's records, 0 when they differ -- what
 * gdx_query_layout_t.uniform_len takes.  A regular file is memory-mapped and a batch parsed by several threads (tiles of 2 MB cut
 * at record starts, every tile checked to end where the next one starts, its records copied while they are in the caches; round 6:
 * one thread delivered 60 M reads a second to calls that take 1-3 G, sixteen deliver 800 M); a batch may hold fewer records than
 * would fit.  Environment: GDX_FASTX_THREADS = their number (default: the CPUs the process may use, at most 32), 0 = the streaming
 * reader (which pipes and stdin get anyway); for tests and experiments GDX_FASTX_BLOCK_BYTES (tile size), GDX_FASTX_NEWLINE_INDEX=0,
 * GDX_FASTX_POPULATE=0, GDX_FASTX_TIMING=1 (a batch's thread-seconds per phase on stderr). */
int gdx_fastx_next_batch_ex(gdx_fastx_t *reader, uint8_t *qbuf, uint64_t qbuf_capacity, uint64_t *qoff,
                            uint64_t max_records, uint64_t *n_out, uint64_t *out_uniform_len);
void gdx_fastx_close(gdx_fastx_t *reader);

#ifdef __cplusplus
}
#endif
#endif
