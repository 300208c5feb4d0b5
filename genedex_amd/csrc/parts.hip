// parts.hip -- collections of texts beyond 2^32 - 1 symbols: a PARTITIONED index (gdx_parts_*).
//
// The reference indexes such collections with `IndexStorage = i64` (construction/mod.rs:225-252).  Every table and
// kernel here keeps rows and text positions in 32 bits, so a collection that does not fit one index is cut at text
// borders into parts of at most 2^32 - 1 symbols (sentinels included), every part gets its own index on the device,
// and a query runs against each of them:
//     count(q)  = sum of the parts' counts           (an occurrence lies inside one text, hence inside one part)
//     locate(q) = the parts' hits, part after part, text ids shifted by the number of texts before the part
// Counts and hit SETS are the reference's.  What a partitioned index cannot give is the single suffix-array interval of
// the whole collection (cursors_for_many_queries, Cursor): those calls exist on one-part indexes only, and the order
// of a query's hits is suffix-array order within a part, parts in text order (the reference documents the order of
// locate() as unspecified, lib.rs:163).  A single text longer than 2^32 - 2 symbols is refused.
#include <algorithm>
#include <cstring>
#include <thread>

#include <sys/mman.h>

#include "fm_index.hpp"

namespace gdx {

Parts::~Parts() = default;

std::unique_ptr<Parts> Parts::build(const uint8_t *texts_buf, bool texts_on_device, const uint64_t *text_offsets,
                                    uint64_t n_texts, const IndexConfig &cfg, uint64_t max_part_symbols)
{
    if (n_texts == 0) fail(GDX_ERR_INVALID_ARGUMENT, "There should be at least one text (construction/mod.rs:303)");
    if (!text_offsets) fail(GDX_ERR_INVALID_ARGUMENT, "text_offsets is null");
    const uint64_t limit = max_part_symbols == 0 || max_part_symbols > 0xffffffffull ? 0xffffffffull : max_part_symbols;
    auto p = std::make_unique<Parts>();
    p->cfg = cfg;
    p->cfg.index_width = 32;  // every part is a 32-bit index
    // greedy cut at text borders: consecutive texts while the part's symbols (one sentinel per text) fit
    uint64_t first = 0;
    while (first < n_texts) {
        uint64_t last = first, symbols = 0;
        while (last < n_texts) {
            if (text_offsets[last + 1] < text_offsets[last]) fail(GDX_ERR_INVALID_ARGUMENT, "text_offsets must be non-decreasing");
            const uint64_t len = text_offsets[last + 1] - text_offsets[last] + 1;
            if (len > limit)
                fail(GDX_ERR_TEXT_TOO_LONG, "text %llu has %llu symbols: a single text must fit one part (%llu symbols)",
                     static_cast<unsigned long long>(last), static_cast<unsigned long long>(len - 1),
                     static_cast<unsigned long long>(limit));
            if (symbols + len > limit) break;
            symbols += len;
            last++;
        }
        p->first_text.push_back(first);
        p->parts.push_back(FmIndex::construct_index(texts_buf, texts_on_device, text_offsets + first, last - first, p->cfg));
        p->total_len += symbols;
        first = last;
    }
    p->first_text.push_back(n_texts);
    return p;
}

int Parts::count_many(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_counts, uint8_t *out_status) const
{
    if (nq != 0 && !out_counts) fail(GDX_ERR_INVALID_ARGUMENT, "out_counts is null");
    std::vector<uint64_t> tmp(parts.size() > 1 ? nq : 0);
    std::vector<uint8_t> st(nq), st_part(parts.size() > 1 ? nq : 0);
    int rc = GDX_OK;
    for (size_t k = 0; k < parts.size(); k++) {
        uint64_t *dst = k == 0 ? out_counts : tmp.data();
        uint8_t *sdst = k == 0 ? st.data() : st_part.data();
        const int r = parts[k]->cursors_for_many_queries(qbuf, qoff, nq, nullptr, nullptr, dst, sdst);
        if (r == GDX_ERR_QUERY_STATUS) rc = r;
        if (k != 0)
            for (uint64_t i = 0; i < nq; i++) {
                out_counts[i] += tmp[i];
                // a symbol is reached in the whole collection iff some part reaches it
                if (st[i] == 0) st[i] = st_part[i];
            }
    }
    for (uint64_t i = 0; i < nq; i++)
        if (st[i] != 0) out_counts[i] = 0;  // the reference panics on such a query: no count
    if (out_status) std::memcpy(out_status, st.data(), nq);
    return rc;
}

int Parts::locate_many_alloc(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets,
                             gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status) const
{
    if (!out_hits || !out_hit_offsets) fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
    *out_hits = nullptr;
    if (out_total) *out_total = 0;
    const size_t g = parts.size();
    if (g == 1) return parts[0]->locate_many_alloc(qbuf, qoff, nq, out_hit_offsets, out_hits, out_total, out_status);
    std::vector<std::vector<uint64_t>> off(g, std::vector<uint64_t>(nq + 1));
    std::vector<gdx_hit_t *> hits(g, nullptr);
    std::vector<uint64_t> total(g, 0);
    std::vector<uint8_t> st(nq, 0), st_part(nq);
    int rc = GDX_OK;
    gdx_hit_t *all = nullptr;
    try {
        for (size_t k = 0; k < g; k++) {
            const int r = parts[k]->locate_many_alloc(qbuf, qoff, nq, off[k].data(), &hits[k], &total[k], st_part.data());
            if (r == GDX_ERR_QUERY_STATUS) rc = r;
            for (uint64_t i = 0; i < nq; i++)
                if (st[i] == 0) st[i] = st_part[i];
        }
        // global offsets: query i holds the hits of part 0, then part 1, ... (a query with a status has none); with
        // gdx_query_options_t.max_hits_per_query = k every part has located at most k hits of a query, and the merged
        // query keeps the first k of them -- locate(q).take(k) of the whole collection, not k per part
        const uint64_t cap = parts[0]->query_options().max_hits_per_query;
        out_hit_offsets[0] = 0;
        for (uint64_t i = 0; i < nq; i++) {
            uint64_t c = 0;
            if (st[i] == 0)
                for (size_t k = 0; k < g; k++) c += off[k][i + 1] - off[k][i];
            if (cap != 0 && c > cap) c = cap;
            out_hit_offsets[i + 1] = out_hit_offsets[i] + c;
        }
        const uint64_t n_hits = out_hit_offsets[nq];
        if (n_hits) {
            const size_t bytes = (n_hits * sizeof(gdx_hit_t) + (2u << 20) - 1) / (2u << 20) * (2u << 20);
            void *mem = nullptr;
            if (posix_memalign(&mem, 2u << 20, bytes) != 0 || !mem) fail(GDX_ERR_DEVICE, "out of host memory for %llu hits", static_cast<unsigned long long>(n_hits));
            (void)madvise(mem, bytes, MADV_HUGEPAGE);
            all = static_cast<gdx_hit_t *>(mem);
            const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency() / 16u + 1u));
            std::vector<std::thread> threads;
            for (unsigned t = 0; t < nt; t++)
                threads.emplace_back([&, t] {
                    for (uint64_t i = nq * t / nt; i < nq * (t + 1) / nt; i++) {
                        if (st[i] != 0) continue;
                        gdx_hit_t *dst = all + out_hit_offsets[i];
                        gdx_hit_t *const end = all + out_hit_offsets[i + 1];
                        for (size_t k = 0; k < g && dst < end; k++) {
                            const uint64_t a = off[k][i], b = off[k][i + 1];
                            for (uint64_t h = a; h < b && dst < end; h++) {
                                dst->text_id = hits[k][h].text_id + first_text[k];
                                dst->position = hits[k][h].position;
                                dst++;
                            }
                        }
                    }
                });
            for (auto &th : threads) th.join();
        }
        *out_hits = all;
        if (out_total) *out_total = n_hits;
        if (out_status) std::memcpy(out_status, st.data(), nq);
    } catch (...) {
        for (gdx_hit_t *h : hits) std::free(h);
        std::free(all);
        throw;
    }
    for (gdx_hit_t *h : hits) std::free(h);
    return rc;
}

}  // namespace gdx
