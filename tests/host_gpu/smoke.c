/* smoke.c -- a GPU host of libgdx.so that is neither Python nor torch: plain C, `-lgdx`, nothing else.
 * It is what a Rust / C caller of the reference's FmIndex::count_many / cursors_for_many_queries / locate_many
 * (/root/reference/src/lib.rs:147-246) looks like through include/gdx.h: host pointers in, host pointers out,
 * the library brings up the HIP runtime itself (the system's libamdhip64, not torch's copy).
 *
 *   smoke <vectors.bin>
 *
 * vectors.bin is written by tests/test_gpu_host_c.py: texts, queries and the oracle's answers for them (intervals,
 * hit offsets, hits in suffix-array order).  Every answer of every index variant below must equal the oracle's bit
 * for bit.  Prints one "ok ..." line per variant and "PASS"; any difference: "FAIL ..." and exit code 1.
 *
 * File layout (little endian u64 unless noted): magic "GDXVEC01", n_texts, text_total, nq, q_total, total_hits,
 * sigma, n_searchable, sa_rate, lookup_depth, io_to_dense u8[256], text_offsets[n_texts+1], texts u8[text_total]
 * (padded to 8), qoff[nq+1], qbuf u8[q_total] (padded to 8), start[nq], end[nq], hit_off[nq+1],
 * hit_text[total_hits], hit_pos[total_hits]. */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gdx.h"

typedef struct {
    uint64_t n_texts, text_total, nq, q_total, total_hits, sigma, n_searchable, sa_rate, lookup_depth;
    uint8_t io_to_dense[256];
    uint64_t *text_offsets, *qoff, *start, *end, *hit_off, *hit_text, *hit_pos;
    uint8_t *texts, *qbuf;
} vectors_t;

static void *read_block(FILE *f, size_t bytes)
{
    const size_t padded = (bytes + 7) & ~(size_t)7;
    void *p = malloc(padded ? padded : 8);
    if (!p || (padded && fread(p, 1, padded, f) != padded)) {
        fprintf(stderr, "FAIL short vector file\n");
        exit(2);
    }
    return p;
}

static void load_vectors(const char *path, vectors_t *v)
{
    FILE *f = fopen(path, "rb");
    char magic[8];
    if (!f || fread(magic, 1, 8, f) != 8 || memcmp(magic, "GDXVEC01", 8) != 0) {
        fprintf(stderr, "FAIL cannot read %s\n", path);
        exit(2);
    }
    if (fread(v, 8, 9, f) != 9 || fread(v->io_to_dense, 1, 256, f) != 256) exit(2);
    v->text_offsets = read_block(f, 8 * (v->n_texts + 1));
    v->texts = read_block(f, v->text_total);
    v->qoff = read_block(f, 8 * (v->nq + 1));
    v->qbuf = read_block(f, v->q_total);
    v->start = read_block(f, 8 * v->nq);
    v->end = read_block(f, 8 * v->nq);
    v->hit_off = read_block(f, 8 * (v->nq + 1));
    v->hit_text = read_block(f, 8 * v->total_hits);
    v->hit_pos = read_block(f, 8 * v->total_hits);
    fclose(f);
}

#define CHECK(call)                                                                         \
    do {                                                                                    \
        const int rc_ = (call);                                                             \
        if (rc_ != GDX_OK) {                                                                \
            printf("FAIL %s: %s -> %d (%s)\n", name, #call, rc_, gdx_last_error());       \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

static int run_variant(const char *name, const vectors_t *v, const gdx_build_options_t *opts)
{
    gdx_index_t *ix = NULL;
    CHECK(gdx_index_build_ex(v->texts, v->text_offsets, v->n_texts, v->io_to_dense, (int)v->sigma, (int)v->n_searchable,
                             v->sa_rate, (int)v->lookup_depth, 32, 0, opts, &ix));
    gdx_index_info_t info;
    CHECK(gdx_index_info(ix, &info));
    uint64_t seed[8] = {0};
    CHECK(gdx_index_seed_info(ix, seed));
    const uint64_t nq = v->nq;
    uint64_t *counts = calloc(nq + 1, 8), *s = calloc(nq + 1, 8), *e = calloc(nq + 1, 8), *off = calloc(nq + 2, 8);
    uint8_t *status = calloc(nq + 1, 1);
    int bad = 0;

    /* FmIndex::count_many lib.rs:155 */
    CHECK(gdx_count_many(ix, v->qbuf, v->qoff, nq, counts, status));
    for (uint64_t i = 0; i < nq && !bad; i++)
        if (counts[i] != v->end[i] - v->start[i] || status[i]) {
            printf("FAIL %s: count of query %" PRIu64 " is %" PRIu64 ", the oracle's %" PRIu64 "\n", name, i, counts[i],
                   v->end[i] - v->start[i]);
            bad = 1;
        }
    /* FmIndex::cursors_for_many_queries lib.rs:241: the half-open intervals, empty ones where they froze */
    CHECK(gdx_cursors_for_many_queries(ix, v->qbuf, v->qoff, nq, s, e, status));
    for (uint64_t i = 0; i < nq && !bad; i++)
        if (s[i] != v->start[i] || e[i] != v->end[i]) {
            printf("FAIL %s: interval of query %" PRIu64 " is [%" PRIu64 ", %" PRIu64 "), the oracle's [%" PRIu64 ", %" PRIu64 ")\n",
                   name, i, s[i], e[i], v->start[i], v->end[i]);
            bad = 1;
        }
    /* FmIndex::locate_many lib.rs:179: offsets + hits in suffix-array order, one pass, library-allocated */
    gdx_hit_t *hits = NULL;
    uint64_t total = 0;
    CHECK(gdx_locate_many_alloc(ix, v->qbuf, v->qoff, nq, off, &hits, &total, status));
    if (!bad && total != v->total_hits) {
        printf("FAIL %s: %" PRIu64 " hits, the oracle's %" PRIu64 "\n", name, total, v->total_hits);
        bad = 1;
    }
    for (uint64_t i = 0; i <= nq && !bad; i++)
        if (off[i] != v->hit_off[i]) {
            printf("FAIL %s: hit offset %" PRIu64 "\n", name, i);
            bad = 1;
        }
    for (uint64_t h = 0; h < total && !bad; h++)
        if (hits[h].text_id != v->hit_text[h] || hits[h].position != v->hit_pos[h]) {
            printf("FAIL %s: hit %" PRIu64 " is (%" PRIu64 ", %" PRIu64 "), the oracle's (%" PRIu64 ", %" PRIu64 ")\n", name, h,
                   hits[h].text_id, hits[h].position, v->hit_text[h], v->hit_pos[h]);
            bad = 1;
        }
    /* a released array may come back from the library's one-array cache: the same call again gives the same hits */
    if (!bad) {
        gdx_hit_t *keep = calloc(total + 1, sizeof(gdx_hit_t));
        memcpy(keep, hits, total * sizeof(gdx_hit_t));
        gdx_free_hits(hits);
        hits = NULL;
        uint64_t total2 = 0;
        CHECK(gdx_locate_many_alloc(ix, v->qbuf, v->qoff, nq, off, &hits, &total2, status));
        if (total2 != total || memcmp(keep, hits, total * sizeof(gdx_hit_t)) != 0) {
            printf("FAIL %s: the second gdx_locate_many_alloc differs from the first\n", name);
            bad = 1;
        }
        free(keep);
    }
    /* the two-call form with a caller-owned buffer: sizing call, then the fill */
    if (!bad) {
        uint64_t need = 0;
        const int rc = gdx_locate_many(ix, v->qbuf, v->qoff, nq, off, NULL, 0, &need, status);
        if ((total && rc != GDX_ERR_CAPACITY) || need != total) {
            printf("FAIL %s: sizing call -> %d, %" PRIu64 " hits\n", name, rc, need);
            bad = 1;
        } else {
            gdx_hit_t *own = calloc(total + 1, sizeof(gdx_hit_t));
            CHECK(gdx_locate_many(ix, v->qbuf, v->qoff, nq, off, own, total, &need, status));
            if (memcmp(own, hits, total * sizeof(gdx_hit_t)) != 0) {
                printf("FAIL %s: gdx_locate_many differs from gdx_locate_many_alloc\n", name);
                bad = 1;
            }
            free(own);
        }
    }
    /* the batch as 2-bit codes (gdx_query_layout_t.packed; reads with a symbol outside A C G T are its exceptions and keep the
       answers of the plain calls): counts and every read's hits are those of the oracle */
    if (!bad && nq) {
        const uint64_t n_sym = v->qoff[nq];
        uint8_t *packed = calloc(gdx_packed_bytes(n_sym), 1);
        uint64_t *exc = calloc(nq, 8), n_exc = 0;
        CHECK(gdx_pack_queries_table(v->io_to_dense, v->qbuf, v->qoff, nq, packed, exc, nq, &n_exc));
        uint8_t *is_exc = calloc(nq, 1);
        for (uint64_t i = 0; i < n_exc; i++) is_exc[exc[i]] = 1;
        gdx_query_layout_t lay;
        gdx_query_layout_init(&lay);
        lay.packed = 1;
        uint64_t *pc = calloc(nq, 8), *poff = calloc(nq + 1, 8), ptotal = 0;
        CHECK(gdx_count_many_layout(ix, packed, v->qoff, nq, &lay, pc, status));
        gdx_hit_t *ph = NULL;
        CHECK(gdx_locate_many_alloc_layout(ix, packed, v->qoff, nq, &lay, poff, &ph, &ptotal, status));
        for (uint64_t i = 0; i < nq && !bad; i++) {
            if (is_exc[i]) continue;
            const uint64_t want = v->hit_off[i + 1] - v->hit_off[i];
            if (pc[i] != want || poff[i + 1] - poff[i] != want ||
                memcmp(ph + poff[i], hits + v->hit_off[i], want * sizeof(gdx_hit_t)) != 0) {
                printf("FAIL %s: packed form, query %" PRIu64 "\n", name, i);
                bad = 1;
            }
        }
        /* the narrow form of the same call (gdx_locate_many_alloc_layout32): u32 offsets and 8-byte hits in pinned memory of
           the library's, given back with gdx_free_hits32; then once more after gdx_release_cached_hits */
        for (int round = 0; round < 2 && !bad; round++) {
            gdx_hits32_t r32;
            CHECK(gdx_locate_many_alloc_layout32(ix, packed, v->qoff, nq, &lay, &r32, status));
            if (r32.nq != nq || r32.total_hits != ptotal || r32.hit_offsets[0] != 0) {
                printf("FAIL %s: narrow packed form, %" PRIu64 " hits of %" PRIu64 "\n", name, r32.total_hits, ptotal);
                bad = 1;
            }
            for (uint64_t i = 0; i < nq && !bad; i++) {
                if (r32.hit_offsets[i + 1] - r32.hit_offsets[i] != poff[i + 1] - poff[i]) bad = 1;
                for (uint64_t h = 0; h < poff[i + 1] - poff[i] && !bad; h++)
                    if (r32.hits[r32.hit_offsets[i] + h].text_id != ph[poff[i] + h].text_id ||
                        r32.hits[r32.hit_offsets[i] + h].position != ph[poff[i] + h].position)
                        bad = 1;
                if (bad) printf("FAIL %s: narrow packed form, query %" PRIu64 "\n", name, i);
            }
            gdx_free_hits32(&r32);
            if (r32.hits != NULL || r32.hit_offsets != NULL) bad = 1; /* (the struct is cleared: nothing dangles) */
            gdx_release_cached_hits();
        }
        gdx_free_hits(ph);
        free(packed), free(exc), free(is_exc), free(pc), free(poff);
    }
    /* Cursor::extend_query_front cursor.rs:34, batched: the last symbol of every query from the empty cursor == its
       one-symbol search (compared with the fused call on those one-symbol queries) */
    if (!bad && nq) {
        uint64_t cs = 0, ce = 0;
        CHECK(gdx_cursor_empty(ix, &cs, &ce));
        if (cs != 0 || ce != info.total_text_len) {
            printf("FAIL %s: empty cursor [%" PRIu64 ", %" PRIu64 ")\n", name, cs, ce);
            bad = 1;
        }
        uint8_t *last = calloc(nq, 1);
        uint64_t *one_off = calloc(nq + 1, 8), *s1 = calloc(nq, 8), *e1 = calloc(nq, 8);
        uint64_t m = 0;
        for (uint64_t i = 0; i < nq; i++)
            if (v->qoff[i + 1] > v->qoff[i]) {
                last[m] = v->qbuf[v->qoff[i + 1] - 1];
                s[m] = cs;
                e[m] = ce;
                m++;
                one_off[m] = m;
            }
        CHECK(gdx_cursor_extend_front_many(ix, s, e, last, m, status));
        CHECK(gdx_cursors_for_many_queries(ix, last, one_off, m, s1, e1, status));
        for (uint64_t i = 0; i < m && !bad; i++)
            if (s[i] != s1[i] || e[i] != e1[i]) {
                printf("FAIL %s: cursor step %" PRIu64 "\n", name, i);
                bad = 1;
            }
        free(last), free(one_off), free(s1), free(e1);
    }
    if (!bad)
        printf("ok %s: n = %" PRIu64 ", %" PRIu64 " queries, %" PRIu64 " hits, seed k = %" PRIu64 ", %.1f MB on the device\n", name,
               info.total_text_len, nq, total, seed[0], info.device_bytes / 1e6);
    gdx_free_hits(hits);
    free(counts), free(s), free(e), free(off), free(status);
    gdx_index_free(ix);
    return bad;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: smoke <vectors.bin>\n");
        return 2;
    }
    vectors_t v;
    load_vectors(argv[1], &v);
    if (gdx_device_count() < 1) {
        printf("FAIL no GPU: %s\n", gdx_last_error());
        return 3;
    }
    int bad = 0;
    /* bench.py's headline index: the reference's arrays + seed table + text units + full suffix array, nothing else */
    gdx_build_options_t headline;
    gdx_build_options_init(&headline);
    headline.pair_lines = 0;
    headline.jump_entry_bytes = 0;
    headline.top_table_depth = 0;
    headline.full_suffix_array = 1;
    headline.seed_symbols = 1;
    bad |= run_variant("headline (seed table + text units + full suffix array)", &v, &headline);
    /* the library's defaults: pair lines + jump table + top table */
    bad |= run_variant("defaults (pair lines + jump table + top table)", &v, NULL);
    /* the reference's arrays alone */
    gdx_build_options_t lean;
    gdx_build_options_init(&lean);
    lean.pair_lines = 0;
    lean.jump_entry_bytes = 0;
    lean.top_table_depth = 0;
    bad |= run_variant("reference arrays only", &v, &lean);
    puts(bad ? "FAILED" : "PASS");
    return bad ? 1 : 0;
}
