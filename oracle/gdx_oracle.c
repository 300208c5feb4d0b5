/*
 * gdx_oracle.c -- CPU restatement of the genedex v0.2.2 query hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY -- see gdx_oracle.h.  Every function cites the
 * reference file:line (relative to /root/reference) whose behaviour it follows.
 */
#define _GNU_SOURCE
#include "gdx_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SUPERBLOCK 65536u /* condensed.rs:34  (u16::MAX + 1) */
#define BLOCK_BITS 64u    /* block.rs:153     Block64::NUM_BITS */
#define BATCH 64          /* lib.rs:115       BATCH_SIZE */

typedef struct {
    int64_t data; /* text_id_search_tree.rs:129-158: threshold >= 0, or !text_id < 0 */
} tree_node;

struct gdxo_index {
    /* alphabet.rs:24-28 */
    uint8_t io_to_dense[256];
    int sigma;        /* num_dense_symbols (incl. sentinel) */
    int n_searchable; /* num_searchable_dense_symbols       */
    int nbits;        /* ilog2_ceil(sigma), condensed.rs:417-419 */
    int width;
    /* lib.rs:93-100 */
    uint64_t n; /* text_len, incl. one sentinel per text */
    uint64_t count[258];
    /* condensed.rs:24-30 */
    uint64_t *blocks;
    uint64_t n_blocks;
    uint16_t *block_offsets;
    uint64_t n_block_offsets;
    uint32_t *superblock_offsets;
    uint64_t n_superblock_offsets;
    /* sampled_suffix_array.rs:18-23 */
    uint32_t *sa_samples;
    uint64_t n_samples;
    uint64_t sa_rate;
    uint64_t *border_keys; /* text_border_lookup keys, sorted  */
    uint64_t *border_vals; /* text_border_lookup values        */
    /* text_id_search_tree.rs:6-9 */
    uint64_t n_texts;
    uint64_t *sentinel_indices;
    tree_node *nodes;
    uint64_t n_nodes;
    /* lookup_table.rs:19-23 */
    int max_depth_plus1; /* tables.len() */
    uint64_t *factors;
    uint32_t **tables; /* tables[d] = (start,end) pairs */
    uint64_t *table_len;
    /* kept for small inputs only (debug / cross-checks) */
    uint8_t *text;
    uint8_t *bwt;
    uint32_t *sa;
};

/* ------------------------------------------------------------------------- */
/* small helpers                                                              */

/* condensed.rs:417-419 ilog2_ceil_for_nonzero */
static int ilog2_ceil(uint64_t v)
{
    int bits = 64 - __builtin_clzll(v);
    int pow2 = (v & (v - 1)) == 0;
    return bits - pow2;
}

static uint64_t div_ceil(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

/* ------------------------------------------------------------------------- */
/* suffix array (third-party libsais in the reference, construction/mod.rs:88-103)
 * Plain Manber-Myers prefix doubling with two counting-sort passes per round.
 * End of string compares smallest (rank 0 for positions >= n).               */

static void suffix_array_doubling(const uint8_t *text, uint64_t n, uint32_t *sa)
{
    if (n == 0) return;
    uint32_t *rank = malloc((n + 1) * sizeof(uint32_t));
    uint32_t *tmp = malloc((n + 1) * sizeof(uint32_t));
    uint32_t *sa2 = malloc((n + 1) * sizeof(uint32_t));
    uint64_t nbuckets = (n > 257 ? n : 257) + 2;
    uint32_t *cnt = malloc(nbuckets * sizeof(uint32_t));

    /* round 0: sort by first symbol; rank = 1 + first slot of the symbol's group */
    memset(cnt, 0, nbuckets * sizeof(uint32_t));
    for (uint64_t i = 0; i < n; i++) cnt[text[i] + 1]++;
    for (uint64_t c = 1; c < 258; c++) cnt[c] += cnt[c - 1];
    for (uint64_t i = 0; i < n; i++) rank[i] = cnt[text[i]] + 1;
    {
        uint32_t *pos = malloc(258 * sizeof(uint32_t));
        memcpy(pos, cnt, 258 * sizeof(uint32_t));
        for (uint64_t i = 0; i < n; i++) sa[pos[text[i]]++] = (uint32_t)i;
        free(pos);
    }

    for (uint64_t h = 1;; h <<= 1) {
        /* pass 1: order by second key rank[i+h] (0 when i+h >= n) */
        memset(cnt, 0, nbuckets * sizeof(uint32_t));
        for (uint64_t i = 0; i < n; i++) {
            uint32_t k2 = (i + h < n) ? rank[i + h] : 0;
            cnt[k2 + 1]++;
        }
        for (uint64_t c = 1; c < nbuckets; c++) cnt[c] += cnt[c - 1];
        for (uint64_t i = 0; i < n; i++) {
            uint32_t k2 = (i + h < n) ? rank[i + h] : 0;
            sa2[cnt[k2]++] = (uint32_t)i;
        }
        /* pass 2: stable order by first key rank[i] */
        memset(cnt, 0, nbuckets * sizeof(uint32_t));
        for (uint64_t i = 0; i < n; i++) cnt[rank[i] + 1]++;
        for (uint64_t c = 1; c < nbuckets; c++) cnt[c] += cnt[c - 1];
        for (uint64_t j = 0; j < n; j++) {
            uint32_t i = sa2[j];
            sa[cnt[rank[i]]++] = i;
        }
        /* new ranks: 1 + first slot of the (rank, rank2) group */
        uint64_t groups = 0;
        uint32_t prev1 = 0, prev2 = 0, cur = 0;
        for (uint64_t j = 0; j < n; j++) {
            uint32_t i = sa[j];
            uint32_t k1 = rank[i];
            uint32_t k2 = (i + h < n) ? rank[i + h] : 0;
            if (j == 0 || k1 != prev1 || k2 != prev2) {
                cur = (uint32_t)j + 1;
                groups++;
            }
            tmp[i] = cur;
            prev1 = k1;
            prev2 = k2;
        }
        memcpy(rank, tmp, n * sizeof(uint32_t));
        if (groups == n) break;
    }
    free(rank);
    free(tmp);
    free(sa2);
    free(cnt);
}

static const uint8_t *g_naive_text;
static uint64_t g_naive_n;
static int naive_cmp(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    uint64_t lx = g_naive_n - x, ly = g_naive_n - y;
    uint64_t l = lx < ly ? lx : ly;
    int c = memcmp(g_naive_text + x, g_naive_text + y, l);
    if (c != 0) return c;
    return (lx < ly) ? -1 : (lx > ly);
}

void gdxo_naive_suffix_array(const uint8_t *text, uint64_t n, uint32_t *sa)
{
    for (uint64_t i = 0; i < n; i++) sa[i] = (uint32_t)i;
    g_naive_text = text;
    g_naive_n = n;
    qsort(sa, n, sizeof(uint32_t), naive_cmp);
}

/* ------------------------------------------------------------------------- */
/* occurrence table construction: condensed.rs:59-124 + fill_superblock :365-415
 * One superblock at a time; the superblock totals are turned into exclusive
 * prefix sums afterwards (:104-115).                                          */

/* Zeroed allocation for the big arrays of the CPU baseline: 2 MiB aligned and advised for transparent huge pages (a
 * 1.7 GB table under random access on 4 KiB pages spends its time in TLB misses), first touched in parallel so that
 * the pages spread over the NUMA nodes of the threads that will read them.  free() releases it.                  */
static void *big_calloc(uint64_t count, size_t size, int n_threads)
{
    size_t bytes = (size_t)(count ? count : 1) * size;
    void *p = NULL;
    if (bytes < ((size_t)4 << 20)) return calloc(count ? count : 1, size);
    size_t rounded = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (posix_memalign(&p, (size_t)2 << 20, rounded) != 0 || !p) return calloc(count ? count : 1, size);
    (void)madvise(p, rounded, MADV_HUGEPAGE);
    int64_t n_pages = (int64_t)(rounded >> 21);
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
    for (int64_t i = 0; i < n_pages; i++) memset((char *)p + ((size_t)i << 21), 0, (size_t)2 << 20);
    return p;
}

static void construct_table(gdxo_index *ix, const uint8_t *text, uint64_t text_len, int n_threads)
{
    int sigma = ix->sigma;
    int nb = ix->nbits;
    uint64_t len = text_len + 1; /* :69 */
    uint64_t blocks_total = div_ceil(len, BLOCK_BITS);
    ix->n_blocks = blocks_total * nb;                     /* :72 */
    ix->n_block_offsets = blocks_total * sigma;           /* :73 */
    ix->n_superblock_offsets = div_ceil(len, SUPERBLOCK) * sigma; /* :74 */
    ix->blocks = big_calloc(ix->n_blocks, sizeof(uint64_t), n_threads);
    ix->block_offsets = big_calloc(ix->n_block_offsets, sizeof(uint16_t), n_threads);
    ix->superblock_offsets = big_calloc(ix->n_superblock_offsets, sizeof(uint32_t), n_threads);

    uint64_t n_sb = div_ceil(len, SUPERBLOCK);
    uint64_t blocks_per_sb = SUPERBLOCK / BLOCK_BITS;
    (void)n_threads;
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
    for (uint64_t sb = 0; sb < n_sb; sb++) {
        /* the text chunk of this superblock; may be empty or missing (:92 zip stops) */
        uint64_t t0 = sb * SUPERBLOCK;
        if (t0 >= text_len) continue; /* no text chunk: zip yields nothing for it */
        uint64_t t1 = t0 + SUPERBLOCK < text_len ? t0 + SUPERBLOCK : text_len;
        uint32_t *sbo = ix->superblock_offsets + sb * sigma;
        uint16_t sums[256];
        memset(sums, 0, sizeof(sums));
        /* blocks that exist in the allocated arrays for this superblock */
        uint64_t first_block = sb * blocks_per_sb;
        uint64_t blocks_here = blocks_total - first_block;
        if (blocks_here > blocks_per_sb) blocks_here = blocks_per_sb;
        uint64_t text_blocks = div_ceil(t1 - t0, BLOCK_BITS);
        for (uint64_t k = 0; k < text_blocks; k++) {
            uint64_t blk = first_block + k;
            uint16_t *bo = ix->block_offsets + blk * sigma;
            for (int c = 0; c < sigma; c++) bo[c] = sums[c]; /* :384 */
            uint64_t *planes = ix->blocks + blk * nb;
            uint64_t p0 = t0 + k * BLOCK_BITS;
            uint64_t p1 = p0 + BLOCK_BITS < t1 ? p0 + BLOCK_BITS : t1;
            for (uint64_t p = p0; p < p1; p++) {
                uint8_t s = text[p];
                sbo[s]++;                           /* :389-390 */
                sums[s] = (uint16_t)(sums[s] + 1u); /* :395 wrapping_add */
                uint8_t t = s;
                for (int b = 0; b < nb; b++) { /* :397-400 */
                    planes[b] |= (uint64_t)(t & 1u) << (p - p0);
                    t >>= 1;
                }
            }
        }
        /* :404-413 blocks_overshoot: the array holds one more block than the text has */
        if (text_blocks < blocks_here) {
            uint16_t *bo = ix->block_offsets + (first_block + blocks_here - 1) * sigma;
            for (int c = 0; c < sigma; c++) bo[c] = sums[c];
        }
    }
    /* :104-115 accumulate superblocks in a single thread */
    uint64_t sum_prev[256];
    memset(sum_prev, 0, sizeof(sum_prev));
    for (uint64_t sb = 0; sb < n_sb; sb++) {
        uint32_t *sbo = ix->superblock_offsets + sb * sigma;
        for (int c = 0; c < sigma; c++) {
            uint64_t temp = sbo[c];
            sbo[c] = (uint32_t)sum_prev[c];
            sum_prev[c] += temp;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* rank / symbol_at: condensed.rs:291-362, block.rs:152-178                    */

static inline uint64_t block64_count_ones_before(uint64_t data, uint64_t idx)
{
    /* block.rs:175-178: data & !(u64::MAX << idx); idx < 64 */
    uint64_t masked = data & ~(UINT64_MAX << idx);
    return (uint64_t)__builtin_popcountll(masked);
}

static inline uint64_t rank_unchecked(const gdxo_index *ix, uint8_t symbol, uint64_t idx)
{
    uint64_t sbo = ix->superblock_offsets[(idx / SUPERBLOCK) * ix->sigma + symbol]; /* :294-304 */
    uint64_t bo = ix->block_offsets[(idx / BLOCK_BITS) * ix->sigma + symbol];       /* :306-311 */
    const uint64_t *planes = ix->blocks + (idx / BLOCK_BITS) * ix->nbits;           /* :313-315 */
    uint64_t acc = planes[0];
    if ((symbol & 1) == 0) acc = ~acc; /* :323-325 */
    for (int b = 1; b < ix->nbits; b++) {
        uint64_t blk = planes[b];
        symbol >>= 1;
        if ((symbol & 1) == 0) blk = ~blk;
        acc &= blk; /* :327-335 */
    }
    return sbo + bo + block64_count_ones_before(acc, idx % BLOCK_BITS); /* :337-340 */
}

static inline uint8_t symbol_at_unchecked(const gdxo_index *ix, uint64_t idx)
{
    const uint64_t *planes = ix->blocks + (idx / BLOCK_BITS) * ix->nbits;
    uint64_t in_block = idx % BLOCK_BITS;
    uint8_t symbol = 0;
    for (int b = 0; b < ix->nbits; b++) symbol |= (uint8_t)(((planes[b] >> in_block) & 1u) << b);
    return symbol; /* :343-362 */
}

int gdxo_rank(const gdxo_index *ix, int symbol, uint64_t idx, uint64_t *out)
{
    if (!(symbol >= 0 && symbol < ix->sigma && idx <= ix->n)) return -1; /* mod.rs:107-108 */
    *out = rank_unchecked(ix, (uint8_t)symbol, idx);
    return 0;
}

int gdxo_symbol_at(const gdxo_index *ix, uint64_t idx, uint8_t *out)
{
    if (!(idx < ix->n)) return -1; /* condensed.rs:344 */
    *out = symbol_at_unchecked(ix, idx);
    return 0;
}

/* lib.rs:273-275 */
static inline uint64_t lf_mapping_step(const gdxo_index *ix, uint8_t symbol, uint64_t idx)
{
    return ix->count[symbol] + rank_unchecked(ix, symbol, idx);
}

/* batched rank: condensed.rs:137-287 -- the same seven staged loops */
typedef struct {
    uint64_t start[BATCH], end[BATCH];
    const uint8_t *q[BATCH];
    uint64_t qlen[BATCH];
    int q_some[BATCH];
    uint64_t query_at_idx[BATCH];
    uint8_t symbols[BATCH];
    uint64_t buffer1[BATCH], buffer2[BATCH], buffer3[BATCH], buffer4[BATCH];
} buffers_t; /* batch_computed_cursors.rs:202-211 */

static void replace_many_unchecked(const gdxo_index *ix, buffers_t *bf, uint64_t m)
{
    const int sigma = ix->sigma, nb = ix->nbits;
    uint64_t *sb_s = bf->buffer1, *sb_e = bf->buffer2, *bo_s = bf->buffer3, *bo_e = bf->buffer4;
    const uint64_t *pl_s[BATCH], *pl_e[BATCH];
    uint64_t acc_s[BATCH], acc_e[BATCH];

    for (uint64_t i = 0; i < m; i++) { /* :159-163 */
        sb_s[i] = (bf->start[i] / SUPERBLOCK) * sigma + bf->symbols[i];
        sb_e[i] = (bf->end[i] / SUPERBLOCK) * sigma + bf->symbols[i];
    }
    for (uint64_t i = 0; i < m; i++) { /* :167-186 */
        sb_s[i] = ix->superblock_offsets[sb_s[i]];
        sb_e[i] = ix->superblock_offsets[sb_e[i]];
    }
    for (uint64_t i = 0; i < m; i++) { /* :189-192 */
        bo_s[i] = (bf->start[i] / BLOCK_BITS) * sigma + bf->symbols[i];
        bo_e[i] = (bf->end[i] / BLOCK_BITS) * sigma + bf->symbols[i];
    }
    for (uint64_t i = 0; i < m; i++) { /* :196-208 */
        bo_s[i] = ix->block_offsets[bo_s[i]];
        bo_e[i] = ix->block_offsets[bo_e[i]];
    }
    for (uint64_t i = 0; i < m; i++) { /* :215-223 */
        pl_s[i] = ix->blocks + (bf->start[i] / BLOCK_BITS) * nb;
        pl_e[i] = ix->blocks + (bf->end[i] / BLOCK_BITS) * nb;
    }
    for (uint64_t i = 0; i < m; i++) { /* :227-248 first plane */
        acc_s[i] = pl_s[i][0];
        acc_e[i] = pl_e[i][0];
    }
    for (uint64_t i = 0; i < m; i++) { /* :250-275 negate / AND the planes */
        uint8_t symbol = bf->symbols[i];
        if ((symbol & 1) == 0) {
            acc_s[i] = ~acc_s[i];
            acc_e[i] = ~acc_e[i];
        }
        for (int b = 1; b < nb; b++) {
            uint64_t bs = pl_s[i][b], be = pl_e[i][b];
            symbol >>= 1;
            if ((symbol & 1) == 0) {
                bs = ~bs;
                be = ~be;
            }
            acc_s[i] &= bs;
            acc_e[i] &= be;
        }
    }
    for (uint64_t i = 0; i < m; i++) { /* :277-286 */
        uint64_t cs = block64_count_ones_before(acc_s[i], bf->start[i] % BLOCK_BITS);
        uint64_t ce = block64_count_ones_before(acc_e[i], bf->end[i] % BLOCK_BITS);
        bf->start[i] = sb_s[i] + bo_s[i] + cs;
        bf->end[i] = sb_e[i] + bo_e[i] + ce;
    }
}

int gdxo_replace_many_interval_borders_with_ranks(const gdxo_index *ix, uint64_t *starts,
                                                  uint64_t *ends, const uint8_t *symbols,
                                                  uint64_t m)
{
    if (m > BATCH) return -1; /* condensed.rs:146 */
    buffers_t bf;
    memset(&bf, 0, sizeof(bf));
    for (uint64_t i = 0; i < m; i++) {
        /* mod.rs:41-52 validity asserts */
        if (!(symbols[i] < ix->sigma && starts[i] <= ix->n && ends[i] <= ix->n)) return -1;
        bf.start[i] = starts[i];
        bf.end[i] = ends[i];
        bf.symbols[i] = symbols[i];
    }
    replace_many_unchecked(ix, &bf, m);
    for (uint64_t i = 0; i < m; i++) {
        starts[i] = bf.start[i];
        ends[i] = bf.end[i];
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* lookup tables: lookup_table.rs                                              */

static inline uint64_t max_depth(const gdxo_index *ix) { return (uint64_t)ix->max_depth_plus1 - 1; }

/* :99-113 / :147-161 with alphabet translation.  status 1 = invalid symbol panic,
 * 2 = a valid but non-searchable symbol (dense-1 >= k) => aliasing or OOB in the reference. */
static int compute_lookup_idx(const gdxo_index *ix, const uint8_t *suffix, uint64_t len,
                              uint64_t *idx_out)
{
    uint64_t idx = 0;
    int status = GDXO_OK;
    for (uint64_t j = 0; j < len; j++) {
        uint8_t dense = ix->io_to_dense[suffix[j]];
        if (dense == 0) return GDXO_INVALID_SYMBOL; /* alphabet.rs:195-198 */
        uint64_t d = (uint64_t)dense - 1;           /* :154-156 */
        if (d >= (uint64_t)ix->n_searchable) status = GDXO_UNSEARCHABLE_IN_LOOKUP;
        idx += d * ix->factors[j];
    }
    *idx_out = idx;
    return status;
}

/* :115-127 */
static uint64_t compute_lookup_idx_dense(const gdxo_index *ix, const uint8_t *suffix, uint64_t len)
{
    uint64_t idx = 0;
    for (uint64_t j = 0; j < len; j++) idx += ((uint64_t)suffix[j] - 1) * ix->factors[j];
    return idx;
}

/* cursor.rs:40-51 */
static inline void extend_front_dense(const gdxo_index *ix, uint64_t *start, uint64_t *end,
                                      uint8_t symbol)
{
    if (*start != *end) {
        uint64_t s = lf_mapping_step(ix, symbol, *start);
        uint64_t e = lf_mapping_step(ix, symbol, *end);
        *start = s;
        *end = e;
    }
}

/* lib.rs:248-271 cursor_for_query_without_alphabet_translation */
static void cursor_for_dense_query(const gdxo_index *ix, const uint8_t *q, uint64_t len,
                                   uint64_t *start, uint64_t *end)
{
    uint64_t depth = len < max_depth(ix) ? len : max_depth(ix); /* lib.rs:277-281 */
    uint64_t suffix_idx = len - depth;
    uint64_t idx = compute_lookup_idx_dense(ix, q + suffix_idx, depth);
    *start = ix->tables[depth][2 * idx];
    *end = ix->tables[depth][2 * idx + 1];
    for (uint64_t r = suffix_idx; r-- > 0;) {
        extend_front_dense(ix, start, end, q[r]);
        if (*end - *start == 0) break;
    }
}

/* :163-181 fill_lookup_tables, :191-212 LookupTable::new, :225-258 fill_table */
static void fill_lookup_tables(gdxo_index *ix, int depth_max)
{
    uint64_t k = (uint64_t)ix->n_searchable;
    ix->factors = malloc(((size_t)depth_max + 1) * sizeof(uint64_t));
    ix->tables = calloc((size_t)depth_max + 1, sizeof(uint32_t *));
    ix->table_len = calloc((size_t)depth_max + 1, sizeof(uint64_t));
    uint64_t f = 1;
    for (int e = 0; e <= depth_max; e++) {
        ix->factors[e] = f;
        f *= k;
    }
    ix->max_depth_plus1 = 0;
    for (int depth = 0; depth <= depth_max; depth++) {
        uint64_t num_values = ix->factors[depth];
        uint32_t *data = calloc(num_values * 2, sizeof(uint32_t));
        if (depth == 0) {
            data[0] = 0;
            data[1] = (uint32_t)ix->n; /* :205-208 */
        } else {
            uint8_t *query = malloc((size_t)depth);
            for (uint64_t idx = 0; idx < num_values; idx++) {
                uint64_t t = idx;
                for (int j = 0; j < depth; j++) { /* digit j <-> query[j], +1 offsets the sentinel */
                    query[j] = (uint8_t)(t % k + 1);
                    t /= k;
                }
                uint64_t s, e;
                cursor_for_dense_query(ix, query, (uint64_t)depth, &s, &e);
                data[2 * idx] = (uint32_t)s;
                data[2 * idx + 1] = (uint32_t)e;
            }
            free(query);
        }
        ix->tables[depth] = data;
        ix->table_len[depth] = num_values;
        ix->max_depth_plus1 = depth + 1; /* tables.push() */
    }
}

/* ------------------------------------------------------------------------- */
/* search drivers                                                              */

/* lib.rs:217-235 */
int gdxo_cursor_for_query(const gdxo_index *ix, const uint8_t *q, uint64_t len, uint64_t *start,
                          uint64_t *end)
{
    uint64_t depth = len < max_depth(ix) ? len : max_depth(ix);
    uint64_t suffix_idx = len - depth;
    uint64_t idx;
    int st = compute_lookup_idx(ix, q + suffix_idx, depth, &idx);
    if (st != GDXO_OK) {
        *start = 0;
        *end = 0;
        return st;
    }
    *start = ix->tables[depth][2 * idx];
    *end = ix->tables[depth][2 * idx + 1];
    for (uint64_t r = suffix_idx; r-- > 0;) {
        uint8_t dense = ix->io_to_dense[q[r]]; /* cursor.rs:34-38 */
        if (dense == 0) {
            *start = 0;
            *end = 0;
            return GDXO_INVALID_SYMBOL;
        }
        extend_front_dense(ix, start, end, dense);
        if (*end - *start == 0) break; /* lib.rs:229-231 */
    }
    return GDXO_OK;
}

int gdxo_extend_query_front(const gdxo_index *ix, uint64_t *start, uint64_t *end,
                            uint8_t io_symbol)
{
    uint8_t dense = ix->io_to_dense[io_symbol];
    if (dense == 0) return GDXO_INVALID_SYMBOL;
    extend_front_dense(ix, start, end, dense);
    return GDXO_OK;
}

/* batch_computed_cursors.rs:131-158 */
static void move_finished_queries_to_end(buffers_t *bf, uint64_t next_idx_in_queries,
                                         uint64_t *num_unfinished)
{
    uint64_t i = 0;
    while (i < *num_unfinished) {
        if (bf->q_some[i] && bf->qlen[i] > next_idx_in_queries && bf->start[i] != bf->end[i]) {
            i++;
            continue;
        }
        uint64_t j = *num_unfinished - 1;
        const uint8_t *tq = bf->q[i];
        bf->q[i] = bf->q[j];
        bf->q[j] = tq;
        uint64_t tl = bf->qlen[i];
        bf->qlen[i] = bf->qlen[j];
        bf->qlen[j] = tl;
        int ts = bf->q_some[i];
        bf->q_some[i] = bf->q_some[j];
        bf->q_some[j] = ts;
        uint64_t t;
        t = bf->start[i]; bf->start[i] = bf->start[j]; bf->start[j] = t;
        t = bf->end[i]; bf->end[i] = bf->end[j]; bf->end[j] = t;
        t = bf->query_at_idx[i]; bf->query_at_idx[i] = bf->query_at_idx[j]; bf->query_at_idx[j] = t;
        *num_unfinished -= 1;
    }
}

/* batch_computed_cursors.rs:36-73 for one batch of <= 64 queries */
static int compute_batch(const gdxo_index *ix, buffers_t *bf, const uint8_t *qbuf,
                         const uint64_t *qoff, uint64_t first, uint64_t batch_size,
                         uint64_t *starts, uint64_t *ends)
{
    for (uint64_t i = 0; i < batch_size; i++) { /* :41-47 */
        bf->q[i] = qbuf + qoff[first + i];
        bf->qlen[i] = qoff[first + i + 1] - qoff[first + i];
        bf->q_some[i] = 1;
        bf->query_at_idx[i] = i;
    }
    /* :75-96 batched_lookup_jumps */
    for (uint64_t i = 0; i < batch_size; i++) {
        uint64_t depth = bf->qlen[i] < max_depth(ix) ? bf->qlen[i] : max_depth(ix);
        uint64_t suffix_idx = bf->qlen[i] - depth;
        uint64_t idx;
        int st = compute_lookup_idx(ix, bf->q[i] + suffix_idx, depth, &idx);
        if (st != GDXO_OK) return st;
        bf->buffer1[i] = depth;
        bf->buffer2[i] = idx;
    }
    for (uint64_t i = 0; i < batch_size; i++) { /* lookup_table.rs:131-140 */
        bf->start[i] = ix->tables[bf->buffer1[i]][2 * bf->buffer2[i]];
        bf->end[i] = ix->tables[bf->buffer1[i]][2 * bf->buffer2[i] + 1];
    }
    uint64_t next_idx_in_queries = max_depth(ix); /* :53 */
    uint64_t num_unfinished = batch_size;
    move_finished_queries_to_end(bf, next_idx_in_queries, &num_unfinished);
    while (num_unfinished > 0) { /* :62-70 */
        /* :98-129 batched_lf_mappings */
        for (uint64_t i = 0; i < num_unfinished; i++) {
            uint64_t rev_idx = bf->qlen[i] - next_idx_in_queries - 1;
            uint8_t dense = ix->io_to_dense[bf->q[i][rev_idx]];
            if (dense == 0) return GDXO_INVALID_SYMBOL;
            bf->symbols[i] = dense;
        }
        replace_many_unchecked(ix, bf, num_unfinished);
        for (uint64_t i = 0; i < num_unfinished; i++) {
            bf->start[i] += ix->count[bf->symbols[i]];
            bf->end[i] += ix->count[bf->symbols[i]];
        }
        next_idx_in_queries += 1;
        move_finished_queries_to_end(bf, next_idx_in_queries, &num_unfinished);
    }
    /* :160-172 move_queries_back_to_initial_order */
    uint64_t i = 0;
    while (i < batch_size) {
        uint64_t j = bf->query_at_idx[i];
        if (i == j) {
            i++;
            continue;
        }
        uint64_t t;
        t = bf->start[i]; bf->start[i] = bf->start[j]; bf->start[j] = t;
        t = bf->end[i]; bf->end[i] = bf->end[j]; bf->end[j] = t;
        t = bf->query_at_idx[i]; bf->query_at_idx[i] = bf->query_at_idx[j]; bf->query_at_idx[j] = t;
    }
    for (uint64_t k = 0; k < batch_size; k++) {
        starts[first + k] = bf->start[k];
        ends[first + k] = bf->end[k];
    }
    return GDXO_OK;
}

static void thread_range(uint64_t nq, int t, int nt, uint64_t *lo, uint64_t *hi)
{
    *lo = nq * (uint64_t)t / (uint64_t)nt;
    *hi = nq * (uint64_t)(t + 1) / (uint64_t)nt;
}

int gdxo_cursors_for_many_queries(const gdxo_index *ix, const uint8_t *qbuf,
                                  const uint64_t *qoff, uint64_t nq, uint64_t *starts,
                                  uint64_t *ends, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    int rc = GDXO_OK;
    /* one contiguous chunk of queries per thread; each thread owns one iterator + Buffers,
     * the way a genedex user would parallelise count_many (the library itself is serial). */
#pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int t = 0, nt = 1;
#endif
        uint64_t lo, hi;
        thread_range(nq, t, nt, &lo, &hi);
        buffers_t *bf = calloc(1, sizeof(buffers_t));
        for (uint64_t first = lo; first < hi; first += BATCH) {
            uint64_t bs = hi - first < BATCH ? hi - first : BATCH;
            int st = compute_batch(ix, bf, qbuf, qoff, first, bs, starts, ends);
            if (st != GDXO_OK) {
#pragma omp critical
                rc = st;
                break;
            }
        }
        free(bf);
    }
    return rc;
}

void gdxo_cursors_single_path(const gdxo_index *ix, const uint8_t *qbuf, const uint64_t *qoff,
                              uint64_t nq, uint64_t *starts, uint64_t *ends, uint8_t *status,
                              int n_threads)
{
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (uint64_t i = 0; i < nq; i++) {
        int st = gdxo_cursor_for_query(ix, qbuf + qoff[i], qoff[i + 1] - qoff[i], &starts[i],
                                       &ends[i]);
        if (status) status[i] = (uint8_t)st;
    }
}

/* ------------------------------------------------------------------------- */
/* locate: sampled_suffix_array.rs:110-138, text_id_search_tree.rs             */

static uint64_t border_lookup(const gdxo_index *ix, uint64_t key)
{
    /* HashMap<usize, I> probe (sampled_suffix_array.rs:124); here a sorted array */
    uint64_t lo = 0, hi = ix->n_texts;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (ix->border_keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return ix->border_vals[lo];
}

static inline uint64_t recover_one(const gdxo_index *ix, uint64_t i)
{
    uint64_t num_steps_done = 0;
    while (i % ix->sa_rate != 0) {
        uint8_t bwt_symbol = symbol_at_unchecked(ix, i);
        if (bwt_symbol == 0) return border_lookup(ix, i) + num_steps_done; /* :123-126 */
        i = lf_mapping_step(ix, bwt_symbol, i);
        num_steps_done += 1;
    }
    return (uint64_t)ix->sa_samples[i / ix->sa_rate] + num_steps_done; /* :133-136 */
}

void gdxo_recover_range(const gdxo_index *ix, uint64_t start, uint64_t end, uint64_t *out)
{
    for (uint64_t i = start; i < end; i++) out[i - start] = recover_one(ix, i);
}

static inline uint64_t left_child(uint64_t i) { return i * 2 + 1; }
static inline uint64_t right_child(uint64_t i) { return (i + 1) * 2; }

/* text_id_search_tree.rs:67-117 add_nodes */
static void add_nodes(tree_node *nodes, uint64_t cur, const uint64_t *indices, uint64_t num,
                      uint64_t indices_offset, uint64_t *max_index_used)
{
    if (cur > *max_index_used) *max_index_used = cur;
    if (num == 1) {
        nodes[cur].data = (int64_t)(~indices_offset); /* new_leaf :147-151 */
        return;
    }
    uint64_t p2 = 1;
    while (p2 < num) p2 <<= 1; /* next_power_of_two */
    uint64_t cur_offset = ((num & (num - 1)) == 0) ? num / 2 : p2 / 2;
    uint64_t threshold = indices[cur_offset - 1];
    nodes[cur].data = (int64_t)threshold;
    add_nodes(nodes, left_child(cur), indices, cur_offset, indices_offset, max_index_used);
    add_nodes(nodes, right_child(cur), indices + cur_offset, num - cur_offset,
              indices_offset + cur_offset, max_index_used);
}

/* :13-33 */
static void build_text_id_tree(gdxo_index *ix)
{
    uint64_t p2 = 1;
    while (p2 < ix->n_texts) p2 <<= 1;
    uint64_t max_needed = p2 * 2 - 1;
    ix->nodes = calloc(max_needed, sizeof(tree_node));
    uint64_t max_used = 0;
    add_nodes(ix->nodes, 0, ix->sentinel_indices, ix->n_texts, 0, &max_used);
    ix->n_nodes = max_used + 1;
}

/* :50-64 */
uint64_t gdxo_lookup_text_id(const gdxo_index *ix, uint64_t pos)
{
    uint64_t cur = 0;
    while (ix->nodes[cur].data >= 0) {
        cur = (pos <= (uint64_t)ix->nodes[cur].data) ? left_child(cur) : right_child(cur);
    }
    return (uint64_t)(~ix->nodes[cur].data);
}

/* :35-48 */
static inline void backtransform(const gdxo_index *ix, uint64_t pos, uint64_t *text_id,
                                 uint64_t *position)
{
    uint64_t t = gdxo_lookup_text_id(ix, pos);
    *text_id = t;
    *position = (t == 0) ? pos : pos - ix->sentinel_indices[t - 1] - 1;
}

/* lib.rs:187-197 locate_interval */
void gdxo_locate_interval(const gdxo_index *ix, uint64_t start, uint64_t end,
                          uint64_t *text_ids, uint64_t *positions)
{
    for (uint64_t i = start; i < end; i++) {
        uint64_t pos = recover_one(ix, i);
        backtransform(ix, pos, &text_ids[i - start], &positions[i - start]);
    }
}

void gdxo_locate_intervals(const gdxo_index *ix, const uint64_t *starts, const uint64_t *ends,
                           uint64_t nq, const uint64_t *hit_offsets, uint64_t *text_ids,
                           uint64_t *positions, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(n_threads)
    for (uint64_t q = 0; q < nq; q++) {
        gdxo_locate_interval(ix, starts[q], ends[q], text_ids + hit_offsets[q],
                             positions + hit_offsets[q]);
    }
}

/* ------------------------------------------------------------------------- */
/* construction plumbing                                                       */

static gdxo_index *alloc_index(const uint8_t *io_to_dense, int sigma, int n_searchable,
                               int index_width)
{
    gdxo_index *ix = calloc(1, sizeof(gdxo_index));
    if (io_to_dense) memcpy(ix->io_to_dense, io_to_dense, 256);
    ix->sigma = sigma;
    ix->n_searchable = n_searchable;
    ix->nbits = ilog2_ceil((uint64_t)sigma);
    ix->width = index_width;
    return ix;
}

static int fits_width(uint64_t n, int width)
{
    /* construction/mod.rs:34 */
    if (width == -32) return n <= 0x7fffffffull;
    if (width == 32) return n <= 0xffffffffull;
    if (width == 64) return n <= 0xffffffffull; /* storage limit of this restatement */
    return 0;
}

/* construction/mod.rs:318-336 frequency_table_to_count */
static void frequency_table_to_count(gdxo_index *ix, const uint64_t *freq)
{
    uint64_t sum = 0;
    for (int c = 0; c < ix->sigma + 1; c++) {
        ix->count[c] = sum;
        sum += freq[c];
    }
}

gdxo_index *gdxo_table_only(const uint8_t *dense_text, uint64_t n, int sigma)
{
    if (sigma < 2) return NULL; /* condensed.rs:64 */
    gdxo_index *ix = alloc_index(NULL, sigma, sigma - 1, 64);
    ix->n = n;
    construct_table(ix, dense_text, n, 1);
    return ix;
}

gdxo_index *gdxo_build(const uint8_t *texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                       const uint8_t *io_to_dense, int sigma, int n_searchable,
                       uint64_t sa_rate, int lookup_depth, int index_width)
{
    if (n_texts == 0 || sa_rate == 0 || sigma < 2) return NULL;
    gdxo_index *ix = alloc_index(io_to_dense, sigma, n_searchable, index_width);
    /* construction/mod.rs:255-308: concatenate, densely encode, one sentinel (0) per text */
    uint64_t total = text_offsets[n_texts] - text_offsets[0];
    uint64_t n = total + n_texts;
    if (!fits_width(n, index_width)) {
        free(ix);
        return NULL;
    }
    ix->n = n;
    ix->n_texts = n_texts;
    ix->sa_rate = sa_rate;
    ix->text = malloc(n ? n : 1);
    ix->sentinel_indices = malloc(n_texts * sizeof(uint64_t));
    uint64_t freq[258];
    memset(freq, 0, sizeof(freq));
    uint64_t w = 0;
    for (uint64_t t = 0; t < n_texts; t++) {
        for (uint64_t p = text_offsets[t]; p < text_offsets[t + 1]; p++) {
            uint8_t d = io_to_dense[texts_buf[p]];
            if (d == 0) { /* alphabet.rs:195-198 panic */
                gdxo_free(ix);
                return NULL;
            }
            ix->text[w++] = d;
            freq[d]++;
        }
        ix->sentinel_indices[t] = w; /* :266-273 */
        ix->text[w++] = 0;
    }
    freq[0] = n_texts; /* :305 */
    frequency_table_to_count(ix, freq);
    build_text_id_tree(ix);

    /* suffix array (libsais in the reference) */
    ix->sa = malloc((n ? n : 1) * sizeof(uint32_t));
    suffix_array_doubling(ix->text, n, ix->sa);

    /* bwt.rs:93-116: bwt[i] = text[SA[i]-1], SA[i]==0 wraps to the last symbol;
     * text_border_lookup = { i -> SA[i] : bwt[i] == 0 } */
    ix->bwt = malloc(n ? n : 1);
    ix->border_keys = malloc(n_texts * sizeof(uint64_t));
    ix->border_vals = malloc(n_texts * sizeof(uint64_t));
    uint64_t nb = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t ti = ix->sa[i] > 0 ? ix->sa[i] : n;
        ix->bwt[i] = ix->text[ti - 1];
        if (ix->bwt[i] == 0) {
            ix->border_keys[nb] = i;
            ix->border_vals[nb] = ix->sa[i];
            nb++;
        }
    }
    /* sampled_suffix_array.rs:37-43: keep SA[i] for i % rate == 0 */
    ix->n_samples = div_ceil(n, sa_rate);
    ix->sa_samples = malloc((ix->n_samples ? ix->n_samples : 1) * sizeof(uint32_t));
    for (uint64_t i = 0, k = 0; i < n; i += sa_rate) ix->sa_samples[k++] = ix->sa[i];

    construct_table(ix, ix->bwt, n, 1);
    fill_lookup_tables(ix, lookup_depth);
    return ix;
}

gdxo_index *gdxo_from_bwt(const uint8_t *bwt, uint64_t n, const uint32_t *sa_samples,
                          uint64_t sa_rate, const uint64_t *border_keys,
                          const uint64_t *border_vals, const uint64_t *sentinel_indices,
                          uint64_t n_texts, const uint8_t *io_to_dense, int sigma,
                          int n_searchable, int lookup_depth, int index_width, int n_threads)
{
    if (n_texts == 0 || sa_rate == 0 || sigma < 2 || !fits_width(n, index_width)) return NULL;
    gdxo_index *ix = alloc_index(io_to_dense, sigma, n_searchable, index_width);
    ix->n = n;
    ix->n_texts = n_texts;
    ix->sa_rate = sa_rate;
    uint64_t freq[258];
    memset(freq, 0, sizeof(freq));
    for (uint64_t i = 0; i < n; i++) freq[bwt[i]]++; /* a permutation of the text */
    frequency_table_to_count(ix, freq);
    ix->sentinel_indices = malloc(n_texts * sizeof(uint64_t));
    memcpy(ix->sentinel_indices, sentinel_indices, n_texts * sizeof(uint64_t));
    build_text_id_tree(ix);
    ix->border_keys = malloc(n_texts * sizeof(uint64_t));
    ix->border_vals = malloc(n_texts * sizeof(uint64_t));
    memcpy(ix->border_keys, border_keys, n_texts * sizeof(uint64_t));
    memcpy(ix->border_vals, border_vals, n_texts * sizeof(uint64_t));
    ix->n_samples = div_ceil(n, sa_rate);
    ix->sa_samples = big_calloc(ix->n_samples, sizeof(uint32_t), n_threads);
    {
        int64_t n_chunks = (int64_t)div_ceil(ix->n_samples, (uint64_t)1 << 19);
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
        for (int64_t c = 0; c < n_chunks; c++) {
            uint64_t a = (uint64_t)c << 19, b = a + ((uint64_t)1 << 19);
            if (b > ix->n_samples) b = ix->n_samples;
            memcpy(ix->sa_samples + a, sa_samples + a, (b - a) * sizeof(uint32_t));
        }
    }
    construct_table(ix, bwt, n, n_threads);
    fill_lookup_tables(ix, lookup_depth);
    return ix;
}

void gdxo_free(gdxo_index *ix)
{
    if (!ix) return;
    free(ix->blocks);
    free(ix->block_offsets);
    free(ix->superblock_offsets);
    free(ix->sa_samples);
    free(ix->border_keys);
    free(ix->border_vals);
    free(ix->sentinel_indices);
    free(ix->nodes);
    free(ix->factors);
    if (ix->tables) {
        for (int d = 0; d < ix->max_depth_plus1; d++) free(ix->tables[d]);
        free(ix->tables);
    }
    free(ix->table_len);
    free(ix->text);
    free(ix->bwt);
    free(ix->sa);
    free(ix);
}

/* ------------------------------------------------------------------------- */
/* the four table variants (condensed / flat) x (Block64 / Block512) as stand-alone tables      */

#define NUM_BLOCK_OFFSET_BITS 16u /* block.rs:3 */

struct gdxo_table {
    int kind, block_bits, sigma, nbits;
    uint64_t n;               /* text_len */
    uint64_t words_per_block; /* block.rs:24 NUM_U64 */
    uint64_t superblock_size; /* condensed: 65536; flat: flat.rs:71-72 */
    uint64_t *blocks;
    uint64_t n_words;
    uint16_t *block_offsets;
    uint64_t n_block_offsets;
    uint32_t *sbo;
    uint64_t n_sbo;
};

/* block.rs:96-100 / :162-164 set_bit_assuming_zero */
static inline void blk_set_bit(uint64_t *blk, uint64_t idx, uint64_t bit) { blk[idx / 64] |= bit << (idx % 64); }
/* block.rs:90-94 / :158-160 get_bit */
static inline uint8_t blk_get_bit(const uint64_t *blk, uint64_t idx) { return (uint8_t)((blk[idx / 64] >> (idx % 64)) & 1u); }
/* block.rs:102-112 (BLOCK512_MASKS :194-226) and :175-178: ones among the first idx bits */
static inline uint64_t blk_count_ones_before(const uint64_t *blk, uint64_t words, uint64_t idx)
{
    uint64_t sum = 0;
    for (uint64_t w = 0; w < words; w++) {
        uint64_t mask;
        if (w < idx / 64) mask = UINT64_MAX;
        else if (w == idx / 64) mask = ~(UINT64_MAX << (idx % 64));
        else mask = 0;
        sum += (uint64_t)__builtin_popcountll(blk[w] & mask);
    }
    return sum;
}

gdxo_table *gdxo_table_construct(const uint8_t *text, uint64_t n, int sigma, int kind, int block_bits)
{
    if (sigma < 2 || (kind != 0 && kind != 1) || (block_bits != 64 && block_bits != 512)) return NULL;
    gdxo_table *t = calloc(1, sizeof(gdxo_table));
    t->kind = kind;
    t->block_bits = block_bits;
    t->sigma = sigma;
    t->nbits = ilog2_ceil((uint64_t)sigma);
    t->n = n;
    t->words_per_block = (uint64_t)block_bits / 64;
    const uint64_t wpb = t->words_per_block;
    const uint64_t len = n + 1;
    if (kind == 0) {
        /* condensed.rs:59-124 with B::NUM_BITS = block_bits */
        const uint64_t bb = (uint64_t)block_bits;
        t->superblock_size = SUPERBLOCK;
        const uint64_t nblk = div_ceil(len, bb);
        t->n_words = nblk * t->nbits * wpb;
        t->n_block_offsets = nblk * sigma;
        t->n_sbo = div_ceil(len, SUPERBLOCK) * sigma;
        t->blocks = calloc(t->n_words ? t->n_words : 1, sizeof(uint64_t));
        t->block_offsets = calloc(t->n_block_offsets ? t->n_block_offsets : 1, sizeof(uint16_t));
        t->sbo = calloc(t->n_sbo ? t->n_sbo : 1, sizeof(uint32_t));
        const uint64_t blocks_per_sb = SUPERBLOCK / bb;
        for (uint64_t sb = 0; sb * SUPERBLOCK < n; sb++) {
            const uint64_t t0 = sb * SUPERBLOCK, t1 = t0 + SUPERBLOCK < n ? t0 + SUPERBLOCK : n;
            uint16_t sums[256];
            memset(sums, 0, sizeof(sums));
            const uint64_t first = sb * blocks_per_sb;
            uint64_t here = nblk - first;
            if (here > blocks_per_sb) here = blocks_per_sb;
            const uint64_t text_blocks = div_ceil(t1 - t0, bb);
            for (uint64_t k = 0; k < text_blocks; k++) {
                uint16_t *bo = t->block_offsets + (first + k) * sigma;
                for (int c = 0; c < sigma; c++) bo[c] = sums[c];
                uint64_t *planes = t->blocks + (first + k) * t->nbits * wpb;
                const uint64_t p0 = t0 + k * bb, p1 = p0 + bb < t1 ? p0 + bb : t1;
                for (uint64_t p = p0; p < p1; p++) {
                    uint8_t s = text[p];
                    t->sbo[sb * sigma + s]++;
                    sums[s] = (uint16_t)(sums[s] + 1u);
                    for (int b = 0; b < t->nbits; b++) {
                        blk_set_bit(planes + (uint64_t)b * wpb, p - p0, (uint64_t)(s & 1u));
                        s >>= 1;
                    }
                }
            }
            if (text_blocks < here) {
                uint16_t *bo = t->block_offsets + (first + here - 1) * sigma;
                for (int c = 0; c < sigma; c++) bo[c] = sums[c];
            }
        }
    } else {
        /* flat.rs:59-126, fill_superblock :268-315 */
        const uint64_t used = (uint64_t)block_bits - NUM_BLOCK_OFFSET_BITS;
        t->superblock_size = ((1u << NUM_BLOCK_OFFSET_BITS) / used) * used;
        const uint64_t nblk = div_ceil(len, used);
        t->n_words = nblk * sigma * wpb;
        t->n_sbo = div_ceil(len, t->superblock_size) * sigma;
        t->blocks = calloc(t->n_words ? t->n_words : 1, sizeof(uint64_t));
        t->sbo = calloc(t->n_sbo ? t->n_sbo : 1, sizeof(uint32_t));
        const uint64_t blocks_per_sb = t->superblock_size / used;
        for (uint64_t sb = 0; sb * t->superblock_size < n; sb++) {
            const uint64_t t0 = sb * t->superblock_size;
            const uint64_t t1 = t0 + t->superblock_size < n ? t0 + t->superblock_size : n;
            uint64_t sums[256];
            memset(sums, 0, sizeof(sums));
            const uint64_t first = sb * blocks_per_sb;
            uint64_t here = nblk - first;
            if (here > blocks_per_sb) here = blocks_per_sb;
            const uint64_t text_blocks = div_ceil(t1 - t0, used);
            for (uint64_t k = 0; k < text_blocks; k++) {
                uint64_t *blks = t->blocks + (first + k) * sigma * wpb;
                for (int c = 0; c < sigma; c++) blks[(uint64_t)c * wpb] = sums[c]; /* integrate_block_offset */
                const uint64_t p0 = t0 + k * used, p1 = p0 + used < t1 ? p0 + used : t1;
                for (uint64_t p = p0; p < p1; p++) {
                    const uint8_t s = text[p];
                    t->sbo[sb * sigma + s]++;
                    sums[s]++;
                    blk_set_bit(blks + (uint64_t)s * wpb, p - p0 + NUM_BLOCK_OFFSET_BITS, 1);
                }
            }
            if (text_blocks < here) {
                uint64_t *blks = t->blocks + (first + here - 1) * sigma * wpb;
                for (int c = 0; c < sigma; c++) blks[(uint64_t)c * wpb] = sums[c];
            }
        }
    }
    /* accumulate superblocks (condensed.rs:104-115, flat.rs:105-116) */
    uint64_t prev[256];
    memset(prev, 0, sizeof(prev));
    for (uint64_t sb = 0; sb * sigma < t->n_sbo; sb++)
        for (int c = 0; c < sigma; c++) {
            const uint64_t tmp = t->sbo[sb * sigma + c];
            t->sbo[sb * sigma + c] = (uint32_t)prev[c];
            prev[c] += tmp;
        }
    return t;
}

void gdxo_table_free(gdxo_table *t)
{
    if (!t) return;
    free(t->blocks);
    free(t->block_offsets);
    free(t->sbo);
    free(t);
}

int gdxo_table_rank(const gdxo_table *t, int symbol, uint64_t idx, uint64_t *out)
{
    if (!(symbol >= 0 && symbol < t->sigma && idx <= t->n)) return -1; /* mod.rs:107-108 */
    const uint64_t wpb = t->words_per_block;
    if (t->kind == 0) { /* condensed.rs:291-341 */
        const uint64_t bb = (uint64_t)t->block_bits;
        const uint64_t sbo = t->sbo[(idx / SUPERBLOCK) * t->sigma + symbol];
        const uint64_t bo = t->block_offsets[(idx / bb) * t->sigma + symbol];
        const uint64_t *planes = t->blocks + (idx / bb) * t->nbits * wpb;
        uint64_t acc[8];
        uint8_t s = (uint8_t)symbol;
        for (uint64_t w = 0; w < wpb; w++) acc[w] = (s & 1) ? planes[w] : ~planes[w];
        for (int b = 1; b < t->nbits; b++) {
            s >>= 1;
            for (uint64_t w = 0; w < wpb; w++) {
                const uint64_t v = planes[(uint64_t)b * wpb + w];
                acc[w] &= (s & 1) ? v : ~v;
            }
        }
        *out = sbo + bo + blk_count_ones_before(acc, wpb, idx % bb);
    } else { /* flat.rs:221-246 */
        const uint64_t used = (uint64_t)t->block_bits - NUM_BLOCK_OFFSET_BITS;
        const uint64_t sbo = t->sbo[(idx / t->superblock_size) * t->sigma + symbol];
        const uint64_t *blk = t->blocks + ((idx / used) * t->sigma + symbol) * wpb;
        uint64_t copy[8];
        memcpy(copy, blk, wpb * sizeof(uint64_t));
        const uint64_t mask = ~(UINT64_MAX << NUM_BLOCK_OFFSET_BITS); /* block.rs:126-133 / :184-191 */
        const uint64_t bo = copy[0] & mask;
        copy[0] &= ~mask;
        *out = sbo + bo + blk_count_ones_before(copy, wpb, idx % used + NUM_BLOCK_OFFSET_BITS);
    }
    return 0;
}

int gdxo_table_symbol_at(const gdxo_table *t, uint64_t idx, uint8_t *out)
{
    if (!(idx < t->n)) return -1;
    const uint64_t wpb = t->words_per_block;
    if (t->kind == 0) { /* condensed.rs:343-362 */
        const uint64_t bb = (uint64_t)t->block_bits;
        const uint64_t *planes = t->blocks + (idx / bb) * t->nbits * wpb;
        uint8_t s = 0;
        for (int b = 0; b < t->nbits; b++) s |= (uint8_t)(blk_get_bit(planes + (uint64_t)b * wpb, idx % bb) << b);
        *out = s;
    } else { /* flat.rs:248-266 */
        const uint64_t used = (uint64_t)t->block_bits - NUM_BLOCK_OFFSET_BITS;
        const uint64_t *blks = t->blocks + (idx / used) * t->sigma * wpb;
        for (int c = 0; c < t->sigma; c++)
            if (blk_get_bit(blks + (uint64_t)c * wpb, idx % used + NUM_BLOCK_OFFSET_BITS)) {
                *out = (uint8_t)c;
                return 0;
            }
        return -2; /* unreachable!() */
    }
    return 0;
}

const uint64_t *gdxo_table_blocks(const gdxo_table *t, uint64_t *n_words)
{
    *n_words = t->n_words;
    return t->blocks;
}
const uint16_t *gdxo_table_block_offsets(const gdxo_table *t, uint64_t *len)
{
    *len = t->n_block_offsets;
    return t->block_offsets;
}
const uint32_t *gdxo_table_superblock_offsets(const gdxo_table *t, uint64_t *len)
{
    *len = t->n_sbo;
    return t->sbo;
}

/* accessors ---------------------------------------------------------------- */
uint64_t gdxo_n(const gdxo_index *ix) { return ix->n; }
uint64_t gdxo_num_texts(const gdxo_index *ix) { return ix->n_texts; }
int gdxo_sigma(const gdxo_index *ix) { return ix->sigma; }
const uint64_t *gdxo_count(const gdxo_index *ix) { return ix->count; }
const uint8_t *gdxo_dense_text(const gdxo_index *ix) { return ix->text; }
const uint8_t *gdxo_bwt(const gdxo_index *ix) { return ix->bwt; }
const uint32_t *gdxo_full_sa(const gdxo_index *ix) { return ix->sa; }
const uint64_t *gdxo_blocks(const gdxo_index *ix, uint64_t *len)
{
    *len = ix->n_blocks;
    return ix->blocks;
}
const uint16_t *gdxo_block_offsets(const gdxo_index *ix, uint64_t *len)
{
    *len = ix->n_block_offsets;
    return ix->block_offsets;
}
const uint32_t *gdxo_superblock_offsets(const gdxo_index *ix, uint64_t *len)
{
    *len = ix->n_superblock_offsets;
    return ix->superblock_offsets;
}
const uint32_t *gdxo_sa_samples(const gdxo_index *ix, uint64_t *len)
{
    *len = ix->n_samples;
    return ix->sa_samples;
}
const uint64_t *gdxo_border_keys(const gdxo_index *ix) { return ix->border_keys; }
const uint64_t *gdxo_border_vals(const gdxo_index *ix) { return ix->border_vals; }
const uint64_t *gdxo_sentinel_indices(const gdxo_index *ix) { return ix->sentinel_indices; }
const uint32_t *gdxo_lookup_table(const gdxo_index *ix, int depth, uint64_t *len)
{
    if (depth < 0 || depth >= ix->max_depth_plus1) {
        *len = 0;
        return NULL;
    }
    *len = ix->table_len[depth];
    return ix->tables[depth];
}
