// locate.hip -- locate walk over the sampled suffix array (sampled_suffix_array.rs:110-138)
// followed by the text-id resolution (text_id_search_tree.rs:35-64), one lane per hit.
//
// Two phases (hits per query are unbounded, e.g. poly-A): (1) exclusive scan of the interval
// sizes -> hit_offsets, (2) every hit slot learns its query through a scattered head marker +
// inclusive max-scan, then walks independently, so load balance does not depend on how the hits
// are distributed over the queries.
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"
#include "kernels.hpp"

namespace gdx {

namespace {

constexpr int kBlock = 256;

struct IntervalSize {
    const uint32_t *start;
    const uint32_t *end;
    uint64_t m;
    __host__ __device__ uint64_t operator()(uint64_t q) const
    {
        return q < m ? static_cast<uint64_t>(end[q] - start[q]) : 0ull;
    }
};

using SizeIterator =
    rocprim::transform_iterator<rocprim::counting_iterator<uint64_t>, IntervalSize, uint64_t>;

// the same from the 16-byte search records {start, end, hint row, hint symbols | status << 24}
struct RecordSize {
    const uint4 *rec;
    const uint32_t *compact;  // null, or the compact results beside the records (kernels.hpp)
    uint64_t m;
    uint32_t max_hits;  // 0 = no limit; a query with more occurrences than this gets no hit slots (it is counted,
                        // not located: what read mappers do with reads from repeats) -- or, with `take`, slots for its
                        // first max_hits rows (lib.rs:187-197: the reference's locate is lazy, callers take(k))
    bool take;
    __host__ __device__ uint64_t operator()(uint64_t q) const
    {
        if (q >= m) return 0ull;
        if (compact != nullptr) {
            const uint32_t c4 = compact[q];
            if (c4 != kCompactSee) return c4 == kCompactNone ? 0ull : 1ull;
        }
        const uint2 v = *reinterpret_cast<const uint2 *>(rec + q);
        const uint32_t c = v.y - v.x;
        return (max_hits != 0u && c > max_hits) ? (take ? static_cast<uint64_t>(max_hits) : 0ull) : static_cast<uint64_t>(c);
    }
};

using RecordSizeIterator =
    rocprim::transform_iterator<rocprim::counting_iterator<uint64_t>, RecordSize, uint64_t>;

// text_id_search_tree.rs:35-64: smallest t with pos <= sentinel_indices[t], position inside that text
// `sentinels` is either the global array or the block's copy of it in LDS (locate_queue_kernel, few texts)
template <bool kWide>
__device__ __forceinline__ void store_hit(const IndexView &ix, uint32_t pos, void *hits_out, uint64_t at,
                                          const uint32_t *sentinels = nullptr)
{
    if (!sentinels) sentinels = ix.sentinels;
    const uint32_t t = lower_bound_u32(sentinels, ix.n_texts, pos);
    const uint32_t in_text = t == 0 ? pos : pos - sentinels[t - 1] - 1u;
    if (kWide) {
        gdx_hit_t out;
        out.text_id = t;
        out.position = in_text;
        static_cast<gdx_hit_t *>(hits_out)[at] = out;
    } else {
        gdx_hit32_t out;
        out.text_id = t;
        out.position = in_text;
        static_cast<gdx_hit32_t *>(hits_out)[at] = out;
    }
}

// hit offsets as the queue kernel reads them: u64[m + 1], or u32[m + 1] when the caller asked for narrow offsets (fewer
// than 2^32 hits: gdx_locate_many_offsets32_hits_compact_dev)
struct HitOffsets {
    const void *p;
    uint32_t narrow;
    __device__ __forceinline__ uint64_t operator[](uint64_t i) const
    {
        return narrow ? static_cast<uint64_t>(static_cast<const uint32_t *>(p)[i]) : static_cast<const uint64_t *>(p)[i];
    }
};

// first[c] = the query that owns hit slot c * chunk (the largest q with hit_offsets[q] <= c * chunk; empty
// queries in between share the offset and are skipped by taking the largest).  One lane per chunk.
// d_total != null (a step without host round trip, gdx_locate_many_step_compact_layout_dev): the number of hit slots is
// read on the device -- `total` is then the capacity of the hit buffer, and what lies beyond it is not located; chunk_flags
// != null: only the entries a flagged chunk reads (its own and the next one's) are computed
__global__ __launch_bounds__(kBlock) void chunk_first_query_kernel(HitOffsets hit_offsets, uint64_t m,
                                                                   uint64_t n_chunks, uint32_t chunk, uint64_t total,
                                                                   uint32_t *__restrict__ first,
                                                                   const unsigned long long *__restrict__ d_total,
                                                                   const uint8_t *__restrict__ chunk_flags,
                                                                   uint32_t *__restrict__ ticket)  // the locate kernel's chunk counter
{
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (c == 0 && ticket != nullptr) *ticket = 0u;
    if (d_total != nullptr) {
        const uint64_t t = *d_total;
        total = t < total ? t : total;
        n_chunks = (total + chunk - 1) / chunk;
        if (total == 0) return;
    }
    if (c > n_chunks) return;
    if (chunk_flags != nullptr && !((c < n_chunks && chunk_flags[c] != 0) || (c > 0 && chunk_flags[c - 1] != 0))) return;
    // entry n_chunks = the query that owns the LAST hit slot, so that the last chunk's block stops there instead of
    // scanning every trailing query without hits
    const uint64_t h0 = c < n_chunks ? c * chunk : total - 1;
    uint64_t lo = 0, hi = m;  // upper_bound over hit_offsets[0 .. m): first q with hit_offsets[q] > h0
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (hit_offsets[mid] <= h0) lo = mid + 1;
        else hi = mid;
    }
    first[c] = static_cast<uint32_t>(lo - 1);  // hit_offsets[0] = 0 <= h0, so lo >= 1
}

constexpr uint32_t kLocateChunk = 2048;  // hit slots a block takes at a time (locate_queue_kernel, locate_stream_kernel)
constexpr size_t kFlagsHead = 16;         // bytes in front of a workspace's chunk flags: the "any chunk flagged" word (below)

// Text ids through a coarse table (text_id_search_tree.rs:35-64 computes the same lower bound): s_tab[b] = the text that holds
// position b << shift, 513 entries over the whole collection, so that the search for a position runs between two neighbouring
// entries -- no step at all unless a text border falls into the position's block (one of 24 texts in 3.1 G symbols: one block
// in twenty) -- instead of a five-step binary search of dependent LDS loads for every hit.
constexpr uint32_t kTextTab = 512;
__device__ __forceinline__ void build_text_table(uint32_t *s_tab, const uint32_t *sentinels, uint32_t n_texts, uint32_t shift)
{
    for (uint32_t b = threadIdx.x; b <= kTextTab; b += kBlock) {
        const uint64_t x = static_cast<uint64_t>(b) << shift;
        s_tab[b] = x > 0xffffffffull ? n_texts : lower_bound_u32(sentinels, n_texts, static_cast<uint32_t>(x));
    }
}
template <bool kWide>
__device__ __forceinline__ void store_hit_tab(const uint32_t *s_tab, uint32_t shift, const uint32_t *sentinels, uint32_t pos,
                                              void *hits_out, uint64_t at)
{
    const uint32_t b = pos >> shift;
    uint32_t lo = s_tab[b], hi = s_tab[b + 1];
    while (lo < hi) {  // smallest t in [lo, hi] with pos <= sentinels[t]
        const uint32_t mid = (lo + hi) >> 1;
        if (sentinels[mid] < pos) lo = mid + 1u;
        else hi = mid;
    }
    const uint32_t in_text = lo == 0u ? pos : pos - sentinels[lo - 1u] - 1u;
    if (kWide) {
        gdx_hit_t out;
        out.text_id = lo;
        out.position = in_text;
        static_cast<gdx_hit_t *>(hits_out)[at] = out;
    } else {
        gdx_hit32_t out;
        out.text_id = lo;
        out.position = in_text;
        static_cast<gdx_hit32_t *>(hits_out)[at] = out;
    }
}

// The locate kernel of an index on which SA[row] is ONE fetch (full suffix array, or 32-byte jump entries): nothing walks, so
// a hit is a record decode, one SA load, a text-id lookup and an 8-byte store -- and what the general queue kernel below
// spends around that (a 256-thread max-scan of 16 barriers per chunk, three dependent loads per slot, a five-step LDS search)
// was most of its time: 573 M hits of a text of repeats took 3.3 ms at 0.26 of the HBM peak.  Same chunks, same slot -> query
// map by head marks; here the marks carry the head's slot, so that a slot knows its number inside its query without
// reading the offsets again, the scan is a wavefront scan with four barriers per chunk, a thread's eight loads of a kind are
// issued together, and text ids come from the coarse table.  Consecutive slots of one query are consecutive rows: their SA loads
// and hit stores are coalesced, their record loads one broadcast.
struct StreamView {
    const uint32_t *sa_full, *jump32;  // SA[row], or word 6 of the 32-byte jump entry of the row
    const uint32_t *sentinels;
    uint32_t n_texts, tab_shift;
    uint32_t skip_single;  // 2: the hits the compact results answer are in place already (launch_scan_offsets_store)
};

// what a slot needs to know of its query, made once per query and chunk (from its offsets, compact result and record) and
// kept in LDS for the chunk's first kStreamDesc queries -- a query of many slots costs its record ONE load, and the slots
// have nothing to wait for but their SA value
constexpr uint32_t kStreamDesc = 512;
constexpr uint32_t kDescSkip = 0, kDescPos = 1, kDescRows = 2, kDescMask = 3, kDescRow = 4;

template <bool kWide>
__global__ __launch_bounds__(kBlock) void locate_stream_kernel(
    StreamView sv, const uint32_t *__restrict__ start, HitOffsets hit_offsets, uint64_t m, const uint32_t *__restrict__ first_query,
    const uint2 *__restrict__ hint, const uint4 *__restrict__ rec, uint64_t total, void *__restrict__ hits_out,
    const uint32_t *__restrict__ compact, const uint8_t *__restrict__ chunk_flags, const unsigned long long *__restrict__ d_total,
    uint32_t *__restrict__ ticket)  // zeroed by chunk_first_query_kernel: blocks take chunks in order as they get done
{
    if (d_total != nullptr) {
        const uint64_t t = *d_total;
        total = t < total ? t : total;
    }
    const uint64_t n_chunks = (total + kLocateChunk - 1) / kLocateChunk;
    // nothing flagged at all (the usual case on a text without repeats: the store pass located the few exceptions itself): done
    if (chunk_flags != nullptr && *reinterpret_cast<const uint32_t *>(chunk_flags - kFlagsHead) == 0u) return;
    constexpr uint32_t kPer = kLocateChunk / kBlock;  // slots per thread
    constexpr uint32_t kLdsTexts = 256;
    __shared__ uint32_t s_map[kLocateChunk];   // the slot of the head of the slot's query + 1 (0 before the scan: no head here)
    __shared__ uint32_t s_qrel[kLocateChunk];  // at a head's slot: its query, relative to the chunk's first -- a full word: a chunk of
                                               // a hit-sparse batch spans millions of queries (21 bits beside the slot were not enough)
    __shared__ uint4 s_desc[kStreamDesc];
    __shared__ uint32_t s_tab[kTextTab + 1];
    __shared__ uint32_t s_sentinels[kLdsTexts];
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_carry;  // slots of the chunk's first query that lie in earlier chunks
    __shared__ uint32_t s_ticket[2];
    const uint32_t *sentinels = sv.sentinels;
    if (threadIdx.x == 0) s_ticket[0] = atomicAdd(ticket, 1u);
    if (sv.n_texts <= kLdsTexts)
        for (uint32_t i = threadIdx.x; i < sv.n_texts; i += kBlock) s_sentinels[i] = sv.sentinels[i];
    __syncthreads();
    if (sv.n_texts <= kLdsTexts) sentinels = s_sentinels;
    build_text_table(s_tab, sentinels, sv.n_texts, sv.tab_shift);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // {a, mask, symbols to subtract, kind} of query q with `n_slots` hit slots (a: the position, the first row, or the row)
    auto describe = [&](uint64_t q, uint64_t n_slots) __attribute__((always_inline)) -> uint4 {
        const uint32_t c4 = compact != nullptr ? compact[q] : kCompactSee;
        if (c4 < kCompactSee)  // the position itself (skip_single 2: launch_scan_offsets_store has stored it)
            return make_uint4(c4, 0u, 0u, sv.skip_single == 2u ? kDescSkip : kDescPos);
        if (rec != nullptr) {
            const uint4 r = rec[q];
            if (r.w & kRecResolved) return make_uint4(r.z, r.x, 0u, kDescPos);  // the search already knows the text position (of two: the second)
            if (r.w & kRecMasked) return make_uint4(r.x, r.z, r.w & 0x1fffffu, kDescMask);  // the rows of the mask, `symbols` steps before the hits
            if (r.z != 0xffffffffu && r.y - r.x == 1u) return make_uint4(r.z, 0u, r.w & 0xffffffu, kDescRow);  // a hinted row
            return make_uint4(r.x, 0u, 0u, kDescRows);
        }
        if (hint != nullptr && n_slots == 1u) {
            const uint2 hv = hint[q];
            if (hv.x != 0xffffffffu && hv.y < (1u << 21)) return make_uint4(hv.x, 0u, hv.y, kDescRow);
        }
        return make_uint4(start[q], 0u, 0u, kDescRows);
    };
    // a block's time is a chain of memory latencies with barriers in between -- the chunk's queries, their offsets / compact
    // results / records, the SA values -- so the chain is kept short: the next chunk's flag and queries are on their way
    // while this one is worked on, and a query's offsets, compact result and record are loaded together, not one after the
    // answer of the other (describe() reads them in turn: the rare slot beyond the descriptors takes that way)
    uint32_t n_flag = 1, n_qa = 0, n_qb = 0;
    auto prefetch_chunk = [&](uint64_t ch) __attribute__((always_inline)) {
        if (ch < n_chunks) {
            n_flag = chunk_flags != nullptr ? chunk_flags[ch] : 1u;
            n_qa = first_query[ch];
            n_qb = first_query[ch + 1];
        }
    };
    // Chunks are handed out by ticket, in order, as blocks get done (a chunk of 2048 single hits and a chunk of one query's
    // 2048 rows take different times: with every block striding over its own share the kernel waited for the unluckiest
    // block -- 3.3 ms for 573 M hits on any grid the chip held at once, 2.7 ms with twice as many blocks, i.e. with the
    // hardware handing out the second half as blocks ended).
    uint64_t chunk = s_ticket[0];
    prefetch_chunk(chunk);
    for (uint32_t it = 0; chunk < n_chunks; it++) {
        const uint32_t flag = n_flag, qa = n_qa, qb = n_qb;
        if (threadIdx.x == 0) s_ticket[(it + 1u) & 1u] = atomicAdd(ticket, 1u);
        __syncthreads();  // the next ticket is there; the previous chunk's map and descriptors are read, the table is built
        const uint64_t next = s_ticket[(it + 1u) & 1u];
        prefetch_chunk(next);
        const uint64_t base = chunk * kLocateChunk;
        const uint32_t cnt = total - base < kLocateChunk ? static_cast<uint32_t>(total - base) : kLocateChunk;
        chunk = next;
        if (flag == 0u) continue;  // (block-uniform; an unflagged chunk's queries were never computed)
#pragma unroll
        for (uint32_t j = 0; j < kPer; j += 4)
            *reinterpret_cast<uint4 *>(&s_map[threadIdx.x * kPer + j]) = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (uint64_t q = static_cast<uint64_t>(qa) + threadIdx.x; q <= qb; q += kBlock) {
            const uint64_t a = hit_offsets[q], b = hit_offsets[q + 1];
            const bool wants = q - qa < kStreamDesc;
            const uint32_t c4 = (wants && compact != nullptr) ? compact[q] : kCompactSee;
            uint4 r = make_uint4(0u, 0u, 0xffffffffu, 0u);
            uint2 hv = make_uint2(0xffffffffu, 0u);
            if (wants) {
                if (rec != nullptr) r = rec[q];
                else r.x = start[q];
                if (rec == nullptr && hint != nullptr) hv = hint[q];
            }
            const uint64_t from = a > base ? a : base;
            if (b > from && from < base + cnt) {
                s_map[from - base] = static_cast<uint32_t>(from - base) + 1u;
                s_qrel[from - base] = static_cast<uint32_t>(q - qa);
                if (q == qa) s_carry = static_cast<uint32_t>(from - a);  // (the owner of slot `base`: always has slots here)
                if (wants) {  // describe(q, b - a) on what has been loaded
                    uint4 d;
                    if (c4 < kCompactSee) d = make_uint4(c4, 0u, 0u, sv.skip_single == 2u ? kDescSkip : kDescPos);
                    else if (rec != nullptr) {
                        if (r.w & kRecResolved) d = make_uint4(r.z, r.x, 0u, kDescPos);
                        else if (r.w & kRecMasked) d = make_uint4(r.x, r.z, r.w & 0x1fffffu, kDescMask);
                        else if (r.z != 0xffffffffu && r.y - r.x == 1u) d = make_uint4(r.z, 0u, r.w & 0xffffffu, kDescRow);
                        else d = make_uint4(r.x, 0u, 0u, kDescRows);
                    } else if (hint != nullptr && b - a == 1u && hv.x != 0xffffffffu && hv.y < (1u << 21)) {
                        d = make_uint4(hv.x, 0u, hv.y, kDescRow);
                    } else {
                        d = make_uint4(r.x, 0u, 0u, kDescRows);
                    }
                    s_desc[q - qa] = d;
                }
            }
        }
        __syncthreads();
        {   // inclusive max-scan of the marks over the chunk's slots: eight consecutive slots per thread, the lanes of a
            // wavefront by shuffles, the four wavefronts through LDS
            uint32_t v[kPer];
#pragma unroll
            for (uint32_t j = 0; j < kPer; j += 4) {
                const uint4 t = *reinterpret_cast<const uint4 *>(&s_map[threadIdx.x * kPer + j]);
                v[j] = t.x, v[j + 1] = t.y, v[j + 2] = t.z, v[j + 3] = t.w;
            }
#pragma unroll
            for (uint32_t j = 1; j < kPer; j++) v[j] = v[j] > v[j - 1] ? v[j] : v[j - 1];
            uint32_t x = v[kPer - 1];
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(x, off);
                if (static_cast<int>(lane) >= off) x = o > x ? o : x;
            }
            if (lane == 63u) s_wave[wave] = x;
            uint32_t before = __shfl_up(x, 1);
            if (lane == 0u) before = 0u;
            __syncthreads();
            for (uint32_t w = 0; w < wave; w++) before = s_wave[w] > before ? s_wave[w] : before;
#pragma unroll
            for (uint32_t j = 0; j < kPer; j++) v[j] = v[j] > before ? v[j] : before;
#pragma unroll
            for (uint32_t j = 0; j < kPer; j += 4)
                *reinterpret_cast<uint4 *>(&s_map[threadIdx.x * kPer + j]) = make_uint4(v[j], v[j + 1], v[j + 2], v[j + 3]);
        }
        __syncthreads();
        const uint32_t carry = s_carry;
        // the thread's eight slots (strided: the slots of a wavefront's instruction are neighbours): descriptor -> row, then the
        // eight SA loads together, then the stores
        uint32_t row[kPer], val[kPer];  // val: the position, or the symbols to subtract from SA[row]
        uint32_t live = 0, need_sa = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint32_t i = threadIdx.x + j * kBlock;
            row[j] = 0;
            val[j] = 0;
            if (i >= cnt) continue;
            const uint32_t head = s_map[i] - 1u;  // (every slot of a chunk lies at or behind a head: slot `base` has an owner)
            const uint32_t qrel = s_qrel[head];
            const uint32_t within = i - head + (qrel == 0u ? carry : 0u);
            uint4 d;
            if (qrel < kStreamDesc) {
                d = s_desc[qrel];
            } else {  // (a chunk of more queries than the block keeps descriptors for: single hits mostly)
                const uint64_t q = static_cast<uint64_t>(qa) + qrel;
                d = describe(q, hit_offsets[q + 1] - hit_offsets[q]);
            }
            if (d.w == kDescSkip) continue;
            live |= 1u << j;
            if (d.w == kDescPos) {
                val[j] = within == 0u ? d.x : d.y;  // (a resolved record of two: its second slot)
                continue;
            }
            need_sa |= 1u << j;
            val[j] = d.z;
            row[j] = d.x;
            if (d.w == kDescRows) {
                row[j] = d.x + within;
            } else if (d.w == kDescMask) {  // the within-th surviving row of the mask
                uint32_t mk = d.y;
                for (uint32_t t = within; t > 0u; t--) mk &= mk - 1u;
                row[j] = d.x + static_cast<uint32_t>(__builtin_ctz(mk | 0x80000000u));
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++)
            if (need_sa & (1u << j))
                row[j] = sv.sa_full != nullptr ? sv.sa_full[row[j]] : sv.jump32[static_cast<uint64_t>(row[j]) * 8u + 6u];
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++)
            if (live & (1u << j))
                store_hit_tab<kWide>(s_tab, sv.tab_shift, sentinels, (need_sa & (1u << j)) ? row[j] - val[j] : val[j], hits_out,
                                     base + threadIdx.x + j * kBlock);
    }
}

// The same index (SA[row] is one fetch), every query's slots open (no chunk flags): BY QUERY instead of by slot.  A wavefront
// takes 64 consecutive queries: a lane reads its query's offsets and record once and writes the hits of a query with up to
// kLaneHits of them itself (resolved positions need nothing else; masked rows one SA line); the queries with more are taken
// one after the other by the whole wavefront, lanes striding over consecutive rows -- coalesced SA loads, coalesced stores.
// No slot -> query map, no ticket, no barrier after the text table: on the text of repeats the stream kernel's chain of
// dependent loads per chunk (offsets -> records -> marks -> scan -> SA) was what its 3.4 ms were made of, not the bytes.
constexpr uint32_t kLaneHits = 4;
template <bool kWide>
__global__ __launch_bounds__(kBlock) void locate_by_query_kernel(
    StreamView sv, const uint32_t *__restrict__ start, HitOffsets hit_offsets, uint64_t m, const uint2 *__restrict__ hint,
    const uint4 *__restrict__ rec, uint64_t total, void *__restrict__ hits_out, const uint32_t *__restrict__ compact,
    const unsigned long long *__restrict__ d_total)
{
    if (d_total != nullptr) {
        const uint64_t t = *d_total;
        total = t < total ? t : total;
    }
    constexpr uint32_t kLdsTexts = 256;
    __shared__ uint32_t s_tab[kTextTab + 1];
    __shared__ uint32_t s_sentinels[kLdsTexts];
    const uint32_t *sentinels = sv.sentinels;
    if (sv.n_texts <= kLdsTexts)
        for (uint32_t i = threadIdx.x; i < sv.n_texts; i += kBlock) s_sentinels[i] = sv.sentinels[i];
    __syncthreads();
    if (sv.n_texts <= kLdsTexts) sentinels = s_sentinels;
    build_text_table(s_tab, sentinels, sv.n_texts, sv.tab_shift);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    auto sa_of = [&](uint32_t row) __attribute__((always_inline)) -> uint32_t {
        return sv.sa_full != nullptr ? sv.sa_full[row] : sv.jump32[static_cast<uint64_t>(row) * 8u + 6u];
    };
    // the text position of hit slot i of a query described by d (locate_stream_kernel's descriptors)
    auto position = [&](const uint4 &d, uint32_t i) __attribute__((always_inline)) -> uint32_t {
        if (d.w == kDescPos) return i == 0u ? d.x : d.y;
        if (d.w == kDescRows) return sa_of(d.x + i);
        if (d.w == kDescMask) {
            uint32_t mk = d.y;
            for (uint32_t t = i; t > 0u; t--) mk &= mk - 1u;
            return sa_of(d.x + static_cast<uint32_t>(__builtin_ctz(mk | 0x80000000u))) - d.z;
        }
        return sa_of(d.x) - d.z;  // kDescRow
    };
    const uint64_t n_groups = (m + 63u) / 64u, g_stride = static_cast<uint64_t>(gridDim.x) * (kBlock / 64);
    // what a lane reads of its query: offsets, compact result, record (or start row and hint).  (Loading the next round's while
    // this round's hits are made changed nothing measurable.)
    struct Loaded {
        uint64_t a, b;
        uint32_t c4;
        uint4 r;
        uint2 hv;
    };
    auto load = [&](uint64_t g) __attribute__((always_inline)) -> Loaded {
        Loaded x;
        x.a = x.b = 0;
        x.c4 = kCompactSee;
        x.r = make_uint4(0u, 0u, 0xffffffffu, 0u);
        x.hv = make_uint2(0xffffffffu, 0u);
        const uint64_t q = g * 64u + lane;
        if (g < n_groups && q < m) {
            x.a = hit_offsets[q];
            x.b = hit_offsets[q + 1];
            if (compact != nullptr) x.c4 = compact[q];
            if (rec != nullptr) x.r = rec[q];
            else x.r.x = start[q];
            if (rec == nullptr && hint != nullptr) x.hv = hint[q];
        }
        return x;
    };
    // kGroups groups of 64 queries per round: a lane's loads of a kind -- offsets and records, then SA values -- go out together
    constexpr uint32_t kGroups = 4;
    for (uint64_t g0 = (static_cast<uint64_t>(blockIdx.x) * (kBlock / 64) + wave) * kGroups; g0 < n_groups; g0 += g_stride * kGroups) {
        Loaded cur[kGroups];
#pragma unroll
        for (uint32_t u = 0; u < kGroups; u++) cur[u] = load(g0 + u);
        uint64_t a[kGroups], n[kGroups];
        uint4 d[kGroups];
#pragma unroll
        for (uint32_t u = 0; u < kGroups; u++) {
            a[u] = cur[u].a;
            n[u] = a[u] >= total ? 0u : (cur[u].b > total ? total : cur[u].b) - a[u];  // (a step into too small a buffer: the rest is not located)
            d[u] = make_uint4(0u, 0u, 0u, kDescSkip);
            if (n[u] != 0u) {
                const uint4 r = cur[u].r;
                if (cur[u].c4 < kCompactSee) d[u] = make_uint4(cur[u].c4, 0u, 0u, sv.skip_single == 2u ? kDescSkip : kDescPos);
                else if (rec != nullptr) {
                    if (r.w & kRecResolved) d[u] = make_uint4(r.z, r.x, 0u, kDescPos);
                    else if (r.w & kRecMasked) d[u] = make_uint4(r.x, r.z, r.w & 0x1fffffu, kDescMask);
                    else if (r.z != 0xffffffffu && r.y - r.x == 1u) d[u] = make_uint4(r.z, 0u, r.w & 0xffffffu, kDescRow);
                    else d[u] = make_uint4(r.x, 0u, 0u, kDescRows);
                } else if (hint != nullptr && n[u] == 1u && cur[u].hv.x != 0xffffffffu && cur[u].hv.y < (1u << 21)) {
                    d[u] = make_uint4(cur[u].hv.x, 0u, cur[u].hv.y, kDescRow);
                } else {
                    d[u] = make_uint4(r.x, 0u, 0u, kDescRows);
                }
                if (d[u].w == kDescSkip) n[u] = 0;
            }
        }
        uint32_t pos[kGroups][kLaneHits];
#pragma unroll
        for (uint32_t u = 0; u < kGroups; u++)
#pragma unroll
            for (uint32_t i = 0; i < kLaneHits; i++) pos[u][i] = (i < n[u] && n[u] <= kLaneHits) ? position(d[u], i) : 0u;
#pragma unroll
        for (uint32_t u = 0; u < kGroups; u++)
#pragma unroll
            for (uint32_t i = 0; i < kLaneHits; i++)
                if (i < n[u] && n[u] <= kLaneHits) store_hit_tab<kWide>(s_tab, sv.tab_shift, sentinels, pos[u][i], hits_out, a[u] + i);
#pragma unroll
        for (uint32_t u = 0; u < kGroups; u++) {
            uint64_t big = __ballot(n[u] > kLaneHits);
            while (big != 0ull) {
                const int l = __builtin_ctzll(big);
                big &= big - 1ull;
                const uint4 dl = make_uint4(__shfl(d[u].x, l), __shfl(d[u].y, l), __shfl(d[u].z, l), __shfl(d[u].w, l));
                const uint64_t al = (static_cast<uint64_t>(__shfl(static_cast<uint32_t>(a[u] >> 32), l)) << 32) | __shfl(static_cast<uint32_t>(a[u]), l);
                const uint32_t nl = __shfl(static_cast<uint32_t>(n[u]), l);  // (a query's hit slots fit 32 bits: its rows do)
                // (four loads of a lane in flight: a query of 700 rows is three round trips to memory, not eleven)
                for (uint32_t i0 = lane; i0 < nl; i0 += 256u) {
                    uint32_t pb[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) pb[k] = i0 + 64u * k < nl ? position(dl, i0 + 64u * k) : 0u;
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++)
                        if (i0 + 64u * k < nl) store_hit_tab<kWide>(s_tab, sv.tab_shift, sentinels, pb[k], hits_out, al + i0 + 64u * k);
                }
            }
        }
    }
}

// The default locate kernel.  A block takes chunks of kLocateChunk consecutive hit slots.  Phase 0, one lane per
// hit, coalesced: the hits that need no walk are finished at once with their single sample read -- the row is
// sampled itself, or the search left a hint for this one-row interval (launch_search: a sampled row the query's
// one-row interval passed through, and how many symbols were still to be consumed there, so that
// SA[hit row] = SA[hint row] - symbols) -- and the others are queued in LDS.  Phase 1: every lane walks a queued
// hit and takes the next one from the queue the moment it is done, so the geometric tail of the walk lengths
// (mean 3 steps at rate 4, maximum over a wavefront ~15) does not idle the other lanes.

// Phase 1 walks through the JUMP TABLE when the index has 16-byte entries (kJumpWalk): the entry of row r names the
// rows after 8 and 16 LF steps, so one fetch offers two candidates for a sampled row (SA[r] = SA[t_j] + 8 j) where a
// rank-line step offers one.  A level that is invalid (a sentinel or a symbol outside 1..4 within its eight steps) is
// crossed with rank-line steps.  (Round 2 walked 32-byte entries the same way over five targets; they now carry SA[r]
// itself: locate_stream_kernel.)
// (the kernel gets the few fields of the IndexView it reads -- the whole view costs SGPRs -- and its queue shares LDS
// with the slot -> query map, so that eight blocks fit a CU: it ran at 5 waves per SIMD before)
struct LocateView {
    const u32x4 *lines;
    const uint32_t *sb_offsets;
    const uint64_t *g_planes;
    const uint16_t *g_block_off;
    const void *jump;
    const uint32_t *sa_full;  // (unused here since round 5: an index with SA[row] at hand runs locate_stream_kernel)
    const uint32_t *count, *sa_samples, *border_keys, *border_vals, *sentinels;
    uint32_t sb_stride, jump_bytes, n_texts, sa_inv, sa_rot, sa_limit;
    int32_t sigma, nbits;
    uint32_t skip_single;  // 2: the hits the compact results answer are in place already (launch_scan_offsets_store)
    uint32_t g_kind, g_wpb, g_used, g_sb;  // IndexView: which of the reference's table variants layout 1 is
};

// (An index on which SA[row] is one fetch -- 32-byte jump entries, a full suffix array -- never comes here: locate_stream_kernel.)
template <class Table, bool kWide, bool kJumpWalk>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void locate_queue_kernel(LocateView lv, const uint32_t *__restrict__ start,
                                                              HitOffsets hit_offsets, uint64_t m,
                                                              const uint32_t *__restrict__ first_query,
                                                              const uint2 *__restrict__ hint,
                                                              const uint4 *__restrict__ rec, uint64_t total,
                                                              void *__restrict__ hits_out,
                                                              unsigned long long *__restrict__ step_stats,
                                                              const uint32_t *__restrict__ compact,
                                                              const uint8_t *__restrict__ chunk_flags,  // != null: only the chunks flagged
                                                              const unsigned long long *__restrict__ d_total)  // != null: see chunk_first_query_kernel
{
    if (d_total != nullptr) {
        const uint64_t t = *d_total;
        total = t < total ? t : total;
    }
    if (chunk_flags != nullptr && *reinterpret_cast<const uint32_t *>(chunk_flags - kFlagsHead) == 0u) return;  // nothing flagged at all
    if (chunk_flags != nullptr) {  // nothing flagged among this block's chunks: done
        const uint64_t n_ch = (total + kLocateChunk - 1) / kLocateChunk;
        int any = 0;
        for (uint64_t ch = blockIdx.x + static_cast<uint64_t>(threadIdx.x) * gridDim.x; ch < n_ch; ch += static_cast<uint64_t>(kBlock) * gridDim.x)
            any |= chunk_flags[ch] != 0;
        if (!__syncthreads_or(any)) return;
    }
    IndexView ix{};
    ix.lines = lv.lines;
    ix.sb_offsets = lv.sb_offsets;
    ix.g_planes = lv.g_planes;
    ix.g_block_off = lv.g_block_off;
    ix.jump = lv.jump;
    ix.count = lv.count;
    ix.sa_samples = lv.sa_samples;
    ix.border_keys = lv.border_keys;
    ix.border_vals = lv.border_vals;
    ix.sentinels = lv.sentinels;
    ix.sb_stride = lv.sb_stride;
    ix.g_kind = lv.g_kind;
    ix.g_wpb = lv.g_wpb;
    ix.g_used = lv.g_used;
    ix.g_sb = lv.g_sb;
    ix.jump_bytes = lv.jump_bytes;
    ix.n_texts = lv.n_texts;
    ix.sa_inv = lv.sa_inv;
    ix.sa_rot = lv.sa_rot;
    ix.sa_limit = lv.sa_limit;
    ix.sigma = lv.sigma;
    ix.nbits = lv.nbits;
    __shared__ uint32_t s_count[257];
    __shared__ uint32_t s_idx[kLocateChunk];  // slot in the chunk (low 11 bits) | symbols to subtract << 11
    // query of every hit slot of the chunk, relative to the chunk's first (+ 1); once every thread holds its slots'
    // queries in registers the same memory is the queue's row array
    __shared__ uint32_t s_query[kLocateChunk];
    uint32_t *const s_row = s_query;
    __shared__ uint32_t s_part[kBlock];
    __shared__ uint32_t s_n, s_head;
    // the text-id search of every hit is a chain of dependent loads: from LDS when the sentinel array is small
    constexpr uint32_t kLdsTexts = 256;
    __shared__ uint32_t s_sentinels[kLdsTexts];
    const uint32_t *sentinels = ix.n_texts <= kLdsTexts ? s_sentinels : ix.sentinels;
    if (ix.n_texts <= kLdsTexts)
        for (uint32_t i = threadIdx.x; i < ix.n_texts; i += kBlock) s_sentinels[i] = ix.sentinels[i];
    for (int i = threadIdx.x; i <= ix.sigma; i += kBlock) s_count[i] = ix.count[i];
    uint32_t walk_steps = 0;  // only reported through step_stats (bench accounting)
    const uint64_t n_chunks = (total + kLocateChunk - 1) / kLocateChunk;
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        if (chunk_flags != nullptr && chunk_flags[chunk] == 0) continue;  // (block-uniform) nothing left to do in this chunk
        const uint64_t base = chunk * kLocateChunk;
        const uint32_t cnt = total - base < kLocateChunk ? static_cast<uint32_t>(total - base) : kLocateChunk;
        __syncthreads();  // the previous chunk's queue is drained, s_count is loaded
        if (threadIdx.x == 0) {
            s_n = 0;
            s_head = 0;
        }
        // hit slot -> query: the chunk's slots belong to the queries qa .. qb (chunk_first_query_kernel).  Every
        // query with hits marks its FIRST slot inside the chunk with its number (+ 1), then an inclusive max-scan over
        // the chunk's slots (8 per thread, partial maxima through LDS) carries the numbers to the other slots: the cost
        // per slot does not depend on how the hits are distributed over the queries (one query with all 2048 slots of
        // the chunk costs what 2048 queries with one hit each cost), and no batch-wide pass over the hits is needed.
        const uint32_t qa = first_query[chunk];
        const uint32_t qb = first_query[chunk + 1];
        for (uint32_t i = threadIdx.x; i < kLocateChunk; i += kBlock) s_query[i] = 0;
        __syncthreads();
        for (uint64_t q = static_cast<uint64_t>(qa) + threadIdx.x; q <= qb; q += kBlock) {
            const uint64_t a = hit_offsets[q], b = hit_offsets[q + 1];
            const uint64_t from = a > base ? a : base;
            if (b > from && from < base + cnt) s_query[from - base] = static_cast<uint32_t>(q - qa) + 1u;
        }
        __syncthreads();
        {
            constexpr uint32_t kPer = kLocateChunk / kBlock;  // consecutive slots per thread
            uint32_t run = 0;
#pragma unroll
            for (uint32_t j = 0; j < kPer; j++) {
                const uint32_t v = s_query[threadIdx.x * kPer + j];
                run = v > run ? v : run;
            }
            s_part[threadIdx.x] = run;
            __syncthreads();
            for (int off = 1; off < kBlock; off <<= 1) {  // inclusive max-scan of the partial maxima
                const uint32_t o = static_cast<int>(threadIdx.x) >= off ? s_part[threadIdx.x - off] : 0u;
                __syncthreads();
                if (o > s_part[threadIdx.x]) s_part[threadIdx.x] = o;
                __syncthreads();
            }
            run = threadIdx.x > 0 ? s_part[threadIdx.x - 1] : 0u;
#pragma unroll
            for (uint32_t j = 0; j < kPer; j++) {
                const uint32_t v = s_query[threadIdx.x * kPer + j];
                run = v > run ? v : run;
                s_query[threadIdx.x * kPer + j] = run;
            }
        }
        __syncthreads();
        constexpr uint32_t kSlotsPerThread = kLocateChunk / kBlock;
        uint32_t qrel[kSlotsPerThread];
#pragma unroll
        for (uint32_t j = 0; j < kSlotsPerThread; j++) qrel[j] = s_query[threadIdx.x + j * kBlock];
        __syncthreads();  // s_query may be overwritten by the queue (s_row) from here on
#pragma unroll
        for (uint32_t j = 0; j < kSlotsPerThread; j++) {
            const uint32_t i = threadIdx.x + j * kBlock;
            if (i >= cnt) break;
            const uint64_t h = base + i;
            const uint32_t q = qa + qrel[j] - 1u;
            const uint64_t first = hit_offsets[q];
            uint32_t row;       // SA index of this hit, or the hinted row
            uint32_t back = 0;  // SA[hit row] = SA[row] - back
            if (rec != nullptr) {
                if (compact != nullptr) {  // (kernels.hpp: the position itself, or "see the record")
                    const uint32_t c4 = compact[q];
                    if (c4 < kCompactSee) {
                        // (skip_single 2: launch_scan_offsets_store has stored these hits already)
                        if (lv.skip_single != 2u) store_hit<kWide>(ix, c4, hits_out, h, sentinels);
                        continue;
                    }
                }
                const uint4 r = rec[q];
                row = r.x + static_cast<uint32_t>(h - first);
                if (r.w & kRecResolved) {  // the search already knows the text position (of two: slot 1 is the first word)
                    store_hit<kWide>(ix, h == first ? r.z : r.x, hits_out, h, sentinels);
                    continue;
                }
                if (r.w & kRecMasked) {  // the (h - first)-th surviving row of the mask, `symbols` steps before the hit
                    uint32_t m = r.z;
                    for (uint32_t t = static_cast<uint32_t>(h - first); t > 0u; t--) m &= m - 1u;
                    row = r.x + static_cast<uint32_t>(__builtin_ctz(m | 0x80000000u));
                    back = r.w & 0x1fffffu;
                } else if (r.z != 0xffffffffu && r.y - r.x == 1u) {
                    row = r.z;
                    back = r.w & 0xffffffu;
                }
            } else {
                row = start[q] + static_cast<uint32_t>(h - first);
                if (hint != nullptr && hit_offsets[q + 1] - first == 1u) {
                    const uint2 hv = hint[q];
                    if (hv.x != 0xffffffffu && hv.y < (1u << 21)) {
                        row = hv.x;
                        back = hv.y;
                    }
                }
            }
            uint32_t slot;
            if (sampled_slot(ix, row, slot)) {  // sampled_suffix_array.rs:133-136 with zero steps
                store_hit<kWide>(ix, ix.sa_samples[slot] - back, hits_out, h, sentinels);
            } else {
                // (a hinted row that is not sampled comes from the search's lazy tail: the walk starts there)
                const uint32_t k = atomicAdd(&s_n, 1u);
                s_row[k] = row;
                s_idx[k] = i | (back << 11);
            }
        }
        __syncthreads();
        const uint32_t queued = s_n;
        if (step_stats && threadIdx.x == 0) atomicAdd(step_stats + 1, static_cast<unsigned long long>(queued));
        bool have = false;
        uint32_t row = 0, steps = 0, idx = 0, back = 0;
        for (;;) {
            if (!have) {
                const uint32_t k = atomicAdd(&s_head, 1u);
                if (k < queued) {
                    row = s_row[k];
                    idx = s_idx[k] & 2047u;
                    back = s_idx[k] >> 11;
                    steps = 0;
                    have = true;
                }
            }
            if (!__any(have)) break;
            bool jumped = false;
            if (kJumpWalk && have) {
                // levels of the entry of `row` (layout.hpp; 16-byte entries here, 32-byte ones run locate_stream_kernel):
                // first the sampled target nearest to row, else as far as the valid levels reach
                const u32x4 e0 = *reinterpret_cast<const u32x4 *>(static_cast<const uint32_t *>(ix.jump) +
                                                                 static_cast<uint64_t>(row) * 4u);
                const uint32_t valid = e0.w >> 16;
                const uint32_t t[2] = {e0.x, e0.y};
                const uint32_t reach = (valid & 2u) ? 2u : (valid & 1u);  // valid levels (cumulative bits)
                uint32_t got = 0;  // first level whose target is a sampled row
#pragma unroll
                for (uint32_t j = 2; j >= 1; j--)
                    if (j <= reach && is_sampled(ix, t[j - 1])) got = j;
                if (got != 0u) {
                    uint32_t slot;
                    (void)sampled_slot(ix, t[got - 1u], slot);
                    steps += got * kJumpSymbols;
                    store_hit<kWide>(ix, ix.sa_samples[slot] + steps - back, hits_out, base + idx, sentinels);
                    walk_steps += steps;
                    have = false;
                    jumped = true;
                } else if (reach != 0u) {
                    row = t[reach - 1u];
                    steps += reach * kJumpSymbols;
                    jumped = true;
                }
            }
            if (have && !jumped) {  // one step of sampled_suffix_array.rs:118-131 (the row is known not to be sampled)
                uint32_t r;
                const uint32_t c = Table::symbol_and_rank(ix, row, r);
                if (c == 0) {  // :121-126 BWT sentinel: the walk reached the start of a text
                    const uint32_t b = lower_bound_u32(ix.border_keys, ix.n_texts, row);
                    store_hit<kWide>(ix, ix.border_vals[b] + steps - back, hits_out, base + idx, sentinels);
                    walk_steps += steps;
                    have = false;
                } else {
                    row = s_count[c] + r;  // lf_mapping_step lib.rs:273-275
                    steps++;
                    uint32_t slot;
                    if (sampled_slot(ix, row, slot)) {
                        store_hit<kWide>(ix, ix.sa_samples[slot] + steps - back, hits_out, base + idx, sentinels);
                        walk_steps += steps;
                        have = false;
                    }
                }
            }
        }
    }
    if (step_stats) atomicAdd(step_stats, static_cast<unsigned long long>(walk_steps));
}

unsigned grid_for_items(uint64_t items)
{
    const uint64_t blocks = (items + kBlock - 1) / kBlock;
    const uint64_t cap = 256u * 8u;
    return static_cast<unsigned>(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

size_t hit_offsets_temp_bytes(uint64_t m)
{
    size_t bytes = 0;
    SizeIterator in(rocprim::counting_iterator<uint64_t>(0), IntervalSize{nullptr, nullptr, m});
    uint64_t *out = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, in, out, uint64_t(0), static_cast<size_t>(m + 1),
                                  rocprim::plus<uint64_t>());
    return bytes;
}

void launch_hit_offsets(const uint32_t *d_start, const uint32_t *d_end, uint64_t m, uint64_t *d_hit_offsets,
                        void *d_temp, size_t temp_bytes, hipStream_t stream)
{
    SizeIterator in(rocprim::counting_iterator<uint64_t>(0), IntervalSize{d_start, d_end, m});
    GDX_HIP(rocprim::exclusive_scan(d_temp, temp_bytes, in, d_hit_offsets, uint64_t(0),
                                    static_cast<size_t>(m + 1), rocprim::plus<uint64_t>(), stream));
}

// Offsets scan in two passes over the counts: tile sums, a scan of the sums by one block, then every tile scans itself
// from its base.  rocPRIM's single-pass scan reads the counts once but its look-back crosses the XCDs (as in
// scan_locate_kernel, profiles/r03/experiments.md section 8): 0.63-0.66 ms for 100 M counts, 2.5 times what the bytes
// take.  Here every wavefront owns 512 consecutive queries and reads / writes them as eight coalesced rows of 64.
constexpr uint32_t kScan2Rows = 8;
constexpr uint32_t kScan2Wave = 64 * kScan2Rows;              // queries per wavefront
constexpr uint32_t kScan2Tile = (kBlock / 64) * kScan2Wave;   // queries per block and tile (2048)
constexpr uint32_t kScanInlineMax = 2048;  // slots of a "see the record" query the store pass locates itself (ScanStore)
// blocks of `kernel` the device holds at once (occupancy x compute units): the grid of the kernels whose blocks stride over
// their work -- one block more per CU than fit would run as a second round at a fraction of the occupancy
template <class Kernel>
static unsigned resident_blocks(Kernel kernel)
{
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1)
        return 1024u;
    return static_cast<unsigned>(per_cu) * static_cast<unsigned>(prop.multiProcessorCount);
}

// the counts of queries q0, q0 + 64, ..: all loads issued before any is looked at (RecordSize::operator() asks for the record
// only after it has seen the compact result -- eight dependent round trips per thread when called in a loop)
__device__ __forceinline__ void scan2_load_counts(const RecordSize &f, uint64_t q0, uint64_t m, unsigned long long (&c)[kScan2Rows],
                                                  uint32_t (&c4)[kScan2Rows])
{
    if (f.compact != nullptr) {
#pragma unroll
        for (uint32_t j = 0; j < kScan2Rows; j++) {
            const uint64_t q = q0 + j * 64u;
            c4[j] = q < m ? f.compact[q] : kCompactNone;
        }
#pragma unroll
        for (uint32_t j = 0; j < kScan2Rows; j++)
            c[j] = c4[j] == kCompactSee ? f(q0 + j * 64u) : (c4[j] == kCompactNone ? 0ull : 1ull);
    } else {
        uint2 v[kScan2Rows];
#pragma unroll
        for (uint32_t j = 0; j < kScan2Rows; j++) {
            const uint64_t q = q0 + j * 64u;
            v[j] = q < m ? *reinterpret_cast<const uint2 *>(f.rec + q) : make_uint2(0u, 0u);
            c4[j] = kCompactSee;
        }
#pragma unroll
        for (uint32_t j = 0; j < kScan2Rows; j++) {
            const uint32_t n = v[j].y - v[j].x;
            c[j] = (f.max_hits != 0u && n > f.max_hits) ? (f.take ? static_cast<unsigned long long>(f.max_hits) : 0ull)
                                                        : static_cast<unsigned long long>(n);
        }
    }
}

// rest (optional, pre-zeroed): += the hit slots of the queries whose compact result says "see the record" (all of them
// when there are no compact results) -- what a locate pass still has to fill after scan2_tile_scan_kernel stored the hits
// of the others
__global__ __launch_bounds__(kBlock) void scan2_tile_sums_kernel(RecordSize f, uint64_t m, unsigned long long *__restrict__ sums,
                                                                 unsigned long long *__restrict__ rest)
{
    __shared__ unsigned long long s_part[kBlock / 64];
    const uint64_t n_tiles = (m + kScan2Tile - 1) / kScan2Tile;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    unsigned long long open_slots = 0;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t q0 = tile * kScan2Tile + wave * kScan2Wave + lane;
        unsigned long long mine = 0, c[kScan2Rows];
        uint32_t c4[kScan2Rows];
        scan2_load_counts(f, q0, m, c, c4);
#pragma unroll
        for (uint32_t j = 0; j < kScan2Rows; j++) {
            mine += c[j];
            if (rest != nullptr && c4[j] == kCompactSee) open_slots += c[j];
        }
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
        if (lane == 0) s_part[wave] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (uint32_t w = 0; w < kBlock / 64; w++) t += s_part[w];
            sums[tile] = t;
        }
        __syncthreads();
    }
    if (rest != nullptr) {  // one atomic per block (a resident grid: launch_scan_totals)
        for (int off = 32; off > 0; off >>= 1) open_slots += __shfl_xor(open_slots, off);
        __syncthreads();
        if (lane == 0) s_part[wave] = open_slots;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (uint32_t w = 0; w < kBlock / 64; w++) t += s_part[w];
            if (t != 0) atomicAdd(rest, t);
        }
    }
}

// exclusive scan of the tile sums in place, by one block; sums[n_tiles] = the total.  Every thread owns 16 consecutive
// sums, a wavefront scans its 1024 with DPP-free shuffles and only the 16 wavefront totals go through LDS: two barriers
// per 16 K sums (the first version scanned in LDS, ten doubling steps with two barriers each: 60 us for the 48.8 K tiles
// of 100 M queries, a tenth of the pass it prepares)
__global__ __launch_bounds__(1024) void scan2_sums_kernel(unsigned long long *__restrict__ sums, uint64_t n_tiles,
                                                          unsigned long long *__restrict__ total_out,  // optional: = sums[n_tiles]
                                                          uint64_t array_stride = 0)  // block b scans the array at sums + b * array_stride
{
    sums += static_cast<uint64_t>(blockIdx.x) * array_stride;
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    constexpr uint32_t kPer = 16;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint64_t base = 0; base < n_tiles; base += 1024ull * kPer) {
        unsigned long long v[kPer], run = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint64_t i = base + threadIdx.x * kPer + j;
            v[j] = i < n_tiles ? sums[i] : 0ull;
            run += v[j];
        }
        unsigned long long x = run;  // inclusive scan of the threads' sums inside the wavefront
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(x, off);
            if (static_cast<int>(lane) >= off) x += o;
        }
        if (lane == 63u) s_wave[wave] = x;
        __syncthreads();
        unsigned long long before = s_carry + x - run;
        for (uint32_t w = 0; w < wave; w++) before += s_wave[w];
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint64_t i = base + threadIdx.x * kPer + j;
            if (i < n_tiles) sums[i] = before;
            before += v[j];
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[n_tiles] = s_carry;
        if (total_out != nullptr) *total_out = s_carry;
    }
}

// what the store pass needs beside the counts (scan2_tile_scan_kernel<true, .>)
struct ScanStore {
    const uint32_t *sentinels;
    uint32_t n_texts, tab_shift;
    void *hits_out;
    uint64_t hits_capacity;
    uint8_t *chunk_flags;  // != null (pre-zeroed): marks the locate chunks that hold slots this pass leaves open
    // SA[row] in one fetch (full suffix array, or 32-byte jump entries: word 6), or both null.  With it the pass locates the
    // queries whose compact result says "see the record" ITSELF when they are few and small (`inline_max` slots at most;
    // d_totals != null: only when the totals say that at most a sixteenth of the slots is theirs): on a text without repeats
    // a few reads in a million, whose chunks the queue kernel then need not visit -- it finds no flag and leaves.
    const uint32_t *sa_full, *jump32;
    const unsigned long long *d_totals;
    uint32_t inline_max;
};

// The second pass of the offsets scan: every tile scans itself from its base.  kStore: the pass also stores the hit of every
// query whose compact result IS its position -- offsets and most hits in one pass over 4 bytes per query; hits at or beyond
// hits_capacity are not stored.  A resident grid: a block takes tiles blockIdx.x, + gridDim.x, ... and has the next tile's
// compact words and base on their way while it scans and stores the current one (one short-lived block per tile paid the
// whole chain load -> scan -> barrier -> base -> stores with nothing else of its own in flight: 0.40 of the HBM peak).
template <bool kStore, bool kWide>
__global__ __launch_bounds__(kBlock) void scan2_tile_scan_kernel(RecordSize f, uint64_t m, const unsigned long long *__restrict__ sums,
                                                                 uint64_t *__restrict__ offsets, ScanStore ss,
                                                                 uint32_t narrow)  // offsets is u32[m + 1] (the total fits)
{
    __shared__ unsigned long long s_part[kBlock / 64];
    constexpr uint32_t kLdsTexts = 256, kExc = 64;
    __shared__ uint32_t s_sentinels[kStore ? kLdsTexts : 1];
    __shared__ uint32_t s_tab[kStore ? kTextTab + 1 : 1];
    __shared__ uint32_t s_exc_q[kStore ? kExc : 1], s_exc_lo[kStore ? kExc : 1], s_exc_hi[kStore ? kExc : 1];
    __shared__ uint32_t s_nexc;
    const uint32_t *sentinels = ss.sentinels;
    bool inline_on = false;
    if (kStore) {
        if (ss.n_texts <= kLdsTexts) {
            for (uint32_t i = threadIdx.x; i < ss.n_texts; i += kBlock) s_sentinels[i] = ss.sentinels[i];
            sentinels = s_sentinels;
            __syncthreads();
        }
        build_text_table(s_tab, sentinels, ss.n_texts, ss.tab_shift);
        if (threadIdx.x == 0) s_nexc = 0;
        inline_on = ss.inline_max != 0u && (ss.sa_full != nullptr || ss.jump32 != nullptr);
        if (inline_on && ss.d_totals != nullptr) inline_on = ss.d_totals[1] * 16ull <= ss.d_totals[0];
        __syncthreads();
    }
    const uint64_t n_tiles = (m + kScan2Tile - 1) / kScan2Tile;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const bool have_compact = f.compact != nullptr;
    // the next tile's compact words (records-only calls read their counts from the records when the tile's turn comes)
    uint32_t c4n[kScan2Rows];
    unsigned long long base_n = 0;
    auto prefetch = [&](uint64_t tile) __attribute__((always_inline)) {
        const uint64_t q0 = tile * kScan2Tile + wave * kScan2Wave + lane;
#pragma unroll
        for (uint32_t j = 0; j < kScan2Rows; j++) {
            const uint64_t q = q0 + j * 64u;
            c4n[j] = have_compact ? (q < m ? f.compact[q] : kCompactNone) : kCompactSee;
        }
        base_n = sums[tile];
    };
    if (blockIdx.x < n_tiles) prefetch(blockIdx.x);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t q0 = tile * kScan2Tile + wave * kScan2Wave + lane;
        uint32_t c4[kScan2Rows], c[kScan2Rows];
        unsigned long long incl[kScan2Rows], carry = 0;
        const unsigned long long tile_base = base_n;
#pragma unroll
        for (uint32_t j = 0; j < kScan2Rows; j++) c4[j] = c4n[j];
        if (tile + gridDim.x < n_tiles) prefetch(tile + gridDim.x);
        if (have_compact) {
#pragma unroll
            for (uint32_t j = 0; j < kScan2Rows; j++)
                c[j] = c4[j] == kCompactSee ? static_cast<uint32_t>(f(q0 + j * 64u)) : (c4[j] == kCompactNone ? 0u : 1u);
        } else {
            uint2 v[kScan2Rows];
#pragma unroll
            for (uint32_t j = 0; j < kScan2Rows; j++) {
                const uint64_t q = q0 + j * 64u;
                v[j] = q < m ? *reinterpret_cast<const uint2 *>(f.rec + q) : make_uint2(0u, 0u);
            }
#pragma unroll
            for (uint32_t j = 0; j < kScan2Rows; j++) {
                const uint32_t n = v[j].y - v[j].x;
                c[j] = (f.max_hits != 0u && n > f.max_hits) ? (f.take ? f.max_hits : 0u) : n;
            }
        }
        bool big = false;
#pragma unroll
        for (uint32_t j = 0; j < kScan2Rows; j++) big = big || c[j] >= (1u << 25);
        // row j = queries q0 - lane + 64 j ..: an inclusive scan across the lanes, rows chained by their totals -- in 32
        // bits when no count of the wavefront could make a row's sum overflow (64 x 2^25), which is practically always
        if (__ballot(big) == 0ull) {
#pragma unroll
            for (uint32_t j = 0; j < kScan2Rows; j++) {
                // inclusive scan over the 64 lanes by DPP: inside the rows of 16 (row_shr 1, 2, 4, 8), then lane 15 of row 0 / 2
                // onto row 1 / 3 (row_bcast15) and lane 31 onto rows 2 and 3 (row_bcast31)
                int x = static_cast<int>(c[j]);
                x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
                x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
                x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
                x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
                x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
                x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
                incl[j] = static_cast<uint32_t>(x) + carry;
                carry += static_cast<uint32_t>(__builtin_amdgcn_readlane(x, 63));
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < kScan2Rows; j++) {
                unsigned long long x = c[j];
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned long long o = __shfl_up(x, off);
                    if (static_cast<int>(lane) >= off) x += o;
                }
                incl[j] = x + carry;
                carry += __shfl(x, 63);
            }
        }
        if (lane == 0) s_part[wave] = carry;  // the wavefront's total
        __syncthreads();
        unsigned long long before = tile_base;
        for (uint32_t w = 0; w < wave; w++) before += s_part[w];
#pragma unroll
        for (uint32_t j = 0; j < kScan2Rows; j++) {
            const uint64_t q = q0 + j * 64u;
            if (q < m) {
                const uint64_t at = before + incl[j] - c[j];
                if (narrow) reinterpret_cast<uint32_t *>(offsets)[q] = static_cast<uint32_t>(at);
                else offsets[q] = at;
                if (kStore && c[j] != 0u) {
                    if (c4[j] < kCompactSee) {
                        if (at < ss.hits_capacity) store_hit_tab<kWide>(s_tab, ss.tab_shift, sentinels, c4[j], ss.hits_out, at);
                    } else {
                        bool open = true;  // the query's slots are left to the queue kernel
                        if (inline_on && c[j] <= ss.inline_max && at + c[j] <= ss.hits_capacity) {
                            const uint32_t k = atomicAdd(&s_nexc, 1u);
                            if (k < kExc) {
                                s_exc_q[k] = static_cast<uint32_t>(q);
                                s_exc_lo[k] = static_cast<uint32_t>(at);
                                s_exc_hi[k] = static_cast<uint32_t>(at >> 32);
                                open = false;
                            }
                        }
                        if (open && ss.chunk_flags != nullptr) {
                            // (chunks that begin at or beyond the capacity have no flag, and nothing of them is located)
                            for (uint64_t ch = at / kLocateChunk; ch <= (at + c[j] - 1u) / kLocateChunk && ch * kLocateChunk < ss.hits_capacity; ch++)
                                ss.chunk_flags[ch] = 1;
                            *reinterpret_cast<uint32_t *>(ss.chunk_flags - kFlagsHead) = 1u;  // (the "any chunk flagged" word)
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (kStore && s_nexc != 0u) {  // (block-uniform) the tile's few "see the record" queries, a wavefront each
            const uint32_t n_exc = s_nexc < kExc ? s_nexc : kExc;
            for (uint32_t e = wave; e < n_exc; e += kBlock / 64) {
                const uint32_t q = s_exc_q[e];
                const uint64_t at = (static_cast<uint64_t>(s_exc_hi[e]) << 32) | s_exc_lo[e];
                const uint4 r = f.rec[q];
                uint32_t cnt = r.y - r.x;
                if (f.max_hits != 0u && cnt > f.max_hits) cnt = f.take ? f.max_hits : 0u;  // RecordSize
                for (uint32_t i = lane; i < cnt; i += 64u) {  // the record's rows as locate_queue_kernel reads them
                    uint32_t pos;
                    if (r.w & kRecResolved) {
                        pos = i == 0u ? r.z : r.x;
                    } else {
                        uint32_t row = r.x + i, back = 0;
                        if (r.w & kRecMasked) {
                            uint32_t mk = r.z;
                            for (uint32_t t = i; t > 0u; t--) mk &= mk - 1u;
                            row = r.x + static_cast<uint32_t>(__builtin_ctz(mk | 0x80000000u));
                            back = r.w & 0x1fffffu;
                        } else if (r.z != 0xffffffffu && r.y - r.x == 1u) {
                            row = r.z;
                            back = r.w & 0xffffffu;
                        }
                        const uint32_t sa = ss.sa_full != nullptr ? ss.sa_full[row] : ss.jump32[static_cast<uint64_t>(row) * 8u + 6u];
                        pos = sa - back;
                    }
                    store_hit_tab<kWide>(s_tab, ss.tab_shift, sentinels, pos, ss.hits_out, at + i);
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) s_nexc = 0;  // (the next tile's first push comes after its first barrier)
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (narrow) reinterpret_cast<uint32_t *>(offsets)[m] = static_cast<uint32_t>(sums[n_tiles]);
        else offsets[m] = sums[n_tiles];
    }
}

size_t hit_offsets_rec_temp_bytes(uint64_t m)
{
    size_t bytes = 0;
    RecordSizeIterator in(rocprim::counting_iterator<uint64_t>(0), RecordSize{nullptr, nullptr, m, 0u, false});
    uint64_t *out = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, in, out, uint64_t(0), static_cast<size_t>(m + 1),
                                  rocprim::plus<uint64_t>());
    const size_t two_pass = ((m + kScan2Tile - 1) / kScan2Tile + 2) * sizeof(unsigned long long);
    return bytes > two_pass ? bytes : two_pass;
}

void launch_hit_offsets_rec(const uint4 *d_rec, uint64_t m, uint64_t *d_hit_offsets, void *d_temp, size_t temp_bytes,
                            hipStream_t stream, uint32_t max_hits, bool take, const uint32_t *d_compact)
{
    static const int env_scan = [] { const char *e = getenv("GDX_SCAN_TWO_PASS"); return e ? atoi(e) : 1; }();  // 0: rocPRIM
    if (env_scan != 0 && m >= 1) {
        const RecordSize f{d_rec, d_compact, m, max_hits, take};
        const uint64_t n_tiles = (m + kScan2Tile - 1) / kScan2Tile;
        unsigned long long *sums = static_cast<unsigned long long *>(d_temp);
        const unsigned grid = static_cast<unsigned>(n_tiles < 65536 ? n_tiles : 65536);
        unsigned long long *const no_rest = nullptr;
        hipLaunchKernelGGL(scan2_tile_sums_kernel, dim3(grid), dim3(kBlock), 0, stream, f, m, sums, no_rest);
        hipLaunchKernelGGL(scan2_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, n_tiles, static_cast<unsigned long long *>(nullptr));
        static const unsigned cap_plain = resident_blocks(scan2_tile_scan_kernel<false, false>);
        hipLaunchKernelGGL((scan2_tile_scan_kernel<false, false>), dim3(static_cast<unsigned>(n_tiles < cap_plain ? n_tiles : cap_plain)),
                           dim3(kBlock), 0, stream, f, m, sums, d_hit_offsets, ScanStore{}, 0u);
        return;
    }
    RecordSizeIterator in(rocprim::counting_iterator<uint64_t>(0), RecordSize{d_rec, d_compact, m, max_hits, take});
    GDX_HIP(rocprim::exclusive_scan(d_temp, temp_bytes, in, d_hit_offsets, uint64_t(0),
                                    static_cast<size_t>(m + 1), rocprim::plus<uint64_t>(), stream));
}

// where launch_scan_offsets_store / launch_locate keep the chunk flags inside a locate workspace of locate_workspace_bytes(total)
// (behind the first-query table of the chunks, which takes n_chunks + 1 of the workspace's total x 4 bytes)
// The flags of a locate workspace: kFlagsHead bytes in front of them hold ONE word that says whether any chunk is flagged at
// all (the locate kernels of a sparse step read that word and leave), then one byte per chunk.  The offset is that of the
// flag bytes; what is zeroed is the region from the word on (locate_flags_region).
size_t locate_chunk_flags_offset(uint64_t total_hits)
{
    const uint64_t n_chunks = (total_hits + kLocateChunk - 1) / kLocateChunk;
    return align_up((n_chunks + 2) * sizeof(uint32_t), 256) + kFlagsHead;
}

size_t locate_chunk_flags_bytes(uint64_t total_hits) { return (total_hits + kLocateChunk - 1) / kLocateChunk + 1; }
// (what a caller zeroes before the store pass: the "any" word and the flag bytes)
void *locate_flags_region(uint8_t *d_chunk_flags) { return d_chunk_flags - kFlagsHead; }
size_t locate_flags_region_bytes(uint64_t total_hits) { return kFlagsHead + locate_chunk_flags_bytes(total_hits); }

size_t scan_totals_workspace_bytes(uint64_t m) { return ((m + kScan2Tile - 1) / kScan2Tile + 2) * sizeof(unsigned long long); }

void launch_scan_totals(const uint4 *d_rec, const uint32_t *d_compact, uint64_t m, uint32_t max_hits, bool take,
                        void *d_scan_workspace, unsigned long long *d_totals, hipStream_t stream)
{
    GDX_HIP(hipMemsetAsync(d_totals, 0, 2 * sizeof(unsigned long long), stream));
    if (m == 0) return;
    const RecordSize f{d_rec, d_compact, m, max_hits, take};
    const uint64_t n_tiles = (m + kScan2Tile - 1) / kScan2Tile;
    unsigned long long *sums = static_cast<unsigned long long *>(d_scan_workspace);
    const unsigned grid = static_cast<unsigned>(n_tiles < 4096 ? n_tiles : 4096);
    hipLaunchKernelGGL(scan2_tile_sums_kernel, dim3(grid), dim3(kBlock), 0, stream, f, m, sums, d_totals + 1);
    hipLaunchKernelGGL(scan2_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, n_tiles, d_totals);
}

// the second half of launch_scan_totals when the search call has filled the tile sums itself (SearchCall::d_tile_sums;
// d_totals[1] holds the open slots already)
void launch_scan_totals_finish(void *d_scan_workspace, uint64_t m, unsigned long long *d_totals, hipStream_t stream)
{
    static_assert(kScan2Tile == kSumTile, "the search kernels count hits per tile of the offsets scan");
    if (m == 0) return;
    const uint64_t n_tiles = (m + kScan2Tile - 1) / kScan2Tile;
    unsigned long long *sums = static_cast<unsigned long long *>(d_scan_workspace);
    hipLaunchKernelGGL(scan2_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, n_tiles, d_totals);
}

void launch_scan_offsets_store(const IndexView &ix, const uint4 *d_rec, const uint32_t *d_compact, uint64_t m, uint32_t max_hits,
                               bool take, const void *d_scan_workspace, uint64_t *d_hit_offsets, void *d_hits,
                               uint64_t hits_capacity, bool wide, hipStream_t stream, bool store, uint8_t *d_chunk_flags,
                               bool narrow_offsets, bool flags_zeroed, bool entry_sa, const unsigned long long *d_totals)
{
    const uint32_t narrow = narrow_offsets ? 1u : 0u;
    if (m == 0) {
        GDX_HIP(hipMemsetAsync(d_hit_offsets, 0, narrow ? sizeof(uint32_t) : sizeof(uint64_t), stream));
        return;
    }
    const RecordSize f{d_rec, d_compact, m, max_hits, take};
    const uint64_t n_tiles = (m + kScan2Tile - 1) / kScan2Tile;
    const unsigned long long *sums = static_cast<const unsigned long long *>(d_scan_workspace);
    if (d_chunk_flags != nullptr && !flags_zeroed)
        GDX_HIP(hipMemsetAsync(locate_flags_region(d_chunk_flags), 0, locate_flags_region_bytes(hits_capacity), stream));
    // (grids the chip holds at once: every block strides over many tiles)
    static const unsigned cap_plain = resident_blocks(scan2_tile_scan_kernel<false, false>);
    static const unsigned cap_store = resident_blocks(scan2_tile_scan_kernel<true, false>);
    static const unsigned cap_wide = resident_blocks(scan2_tile_scan_kernel<true, true>);
    auto grid_of = [&](unsigned cap) { return dim3(static_cast<unsigned>(n_tiles < cap ? n_tiles : cap)); };
    if (!store || d_compact == nullptr || d_hits == nullptr) {
        hipLaunchKernelGGL((scan2_tile_scan_kernel<false, false>), grid_of(cap_plain), dim3(kBlock), 0, stream, f, m, sums, d_hit_offsets,
                           ScanStore{}, narrow);
        return;
    }
    uint32_t shift = 0;
    while ((static_cast<uint64_t>(ix.n) >> shift) >= kTextTab) shift++;
    // (inline location of the few "see the record" queries: only where SA[row] is one fetch, launch_locate's entry_sa)
    const uint32_t *jump32 = entry_sa && ix.sa_full == nullptr && ix.jump != nullptr && ix.jump_bytes == 32 ? static_cast<const uint32_t *>(ix.jump) : nullptr;
    const ScanStore ss{ix.sentinels, ix.n_texts, shift, d_hits, hits_capacity, d_chunk_flags, entry_sa ? ix.sa_full : nullptr, jump32,
                       d_totals, entry_sa ? kScanInlineMax : 0u};
    if (wide)
        hipLaunchKernelGGL((scan2_tile_scan_kernel<true, true>), grid_of(cap_wide), dim3(kBlock), 0, stream, f, m, sums, d_hit_offsets, ss, narrow);
    else
        hipLaunchKernelGGL((scan2_tile_scan_kernel<true, false>), grid_of(cap_store), dim3(kBlock), 0, stream, f, m, sums, d_hit_offsets, ss, narrow);
}

// SA[row] of any row in one fetch (32-byte jump entries, or the full suffix array): launch_locate then never walks, and the
// store pass may locate the few "see the record" queries of a sparse batch itself
bool locate_entry_sa(const IndexView &ix, const QueryOptions &qo)
{
    const bool jump_walk = ix.layout == 0 && ix.jump != nullptr && ix.jump_bytes >= 16 && qo.locate_jump_walk != 0;
    return (jump_walk && ix.jump_bytes == 32) || (ix.layout == 0 && ix.sa_full != nullptr && qo.locate_jump_walk != 0);
}

namespace {
struct CountAt {
    const uint32_t *counts;
    uint64_t m;
    __host__ __device__ uint64_t operator()(uint64_t q) const { return q < m ? static_cast<uint64_t>(counts[q]) : 0ull; }
};
using CountIterator = rocprim::transform_iterator<rocprim::counting_iterator<uint64_t>, CountAt, uint64_t>;

__global__ __launch_bounds__(kBlock) void unpack_records_kernel(const uint4 *__restrict__ rec, uint64_t m,
                                                                uint32_t *__restrict__ counts, uint8_t *__restrict__ status,
                                                                const uint32_t *__restrict__ compact, unsigned long long *__restrict__ any_status)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; q < m; q += stride) {
        if (compact != nullptr) {
            const uint32_t c4 = compact[q];
            if (c4 != kCompactSee) {
                if (counts) counts[q] = c4 == kCompactNone ? 0u : 1u;
                if (status) status[q] = 0;
                continue;
            }
        }
        const uint4 r = rec[q];
        if (counts) counts[q] = r.y - r.x;
        if (status) status[q] = static_cast<uint8_t>(r.w >> 24);
        if (any_status != nullptr && (r.w >> 24) != 0u) atomicOr(any_status, 1ull);  // (rare: reads with symbols outside the alphabet)
    }
}

// compact results as they travel between devices (one u32 per query, gdx_compact_split_hits_dev): text id byte and
// position in that text of the only hit, position -1 = no occurrence, -2 = "see the exceptions".  Four queries per thread:
// one 16-byte load, one 4-byte and one 16-byte store.
__global__ __launch_bounds__(kBlock) void compact_split_kernel(const uint32_t *__restrict__ compact, uint64_t m,
                                                               const uint32_t *__restrict__ sentinels, uint32_t n_texts,
                                                               uint8_t *__restrict__ ids, int32_t *__restrict__ pos)
{
    __shared__ uint32_t s_sent[256];
    for (uint32_t i = threadIdx.x; i < n_texts; i += kBlock) s_sent[i] = sentinels[i];
    __syncthreads();
    auto split = [&](uint32_t c4, uint32_t &id, int32_t &p) {
        id = 0u;
        p = c4 == kCompactNone ? -1 : -2;
        if (c4 < kCompactSee) {
            id = lower_bound_u32(s_sent, n_texts, c4);
            p = static_cast<int32_t>(id == 0u ? c4 : c4 - s_sent[id - 1u] - 1u);
        }
    };
    const uint64_t quads = m / 4u;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < quads; i += stride) {
        const u32x4 c = reinterpret_cast<const u32x4 *>(compact)[i];
        uint32_t i0, i1, i2, i3;
        int32_t p0, p1, p2, p3;
        split(c.x, i0, p0);
        split(c.y, i1, p1);
        split(c.z, i2, p2);
        split(c.w, i3, p3);
        reinterpret_cast<uint32_t *>(ids)[i] = i0 | (i1 << 8) | (i2 << 16) | (i3 << 24);
        reinterpret_cast<int4 *>(pos)[i] = make_int4(p0, p1, p2, p3);
    }
    if (blockIdx.x == 0 && threadIdx.x < m - quads * 4u) {
        const uint64_t q = quads * 4u + threadIdx.x;
        uint32_t id;
        int32_t p;
        split(compact[q], id, p);
        ids[q] = static_cast<uint8_t>(id);
        pos[q] = p;
    }
}

// ---- the "found bitmap" wire of the multi-GPU gather (gdx_wire_pack_dev / gdx_wire_split_dev) ---------
// What a rank sends to the root for a count + locate shard: one BIT per read (its compact result is a position: exactly one
// hit), the text positions of those reads back to back in read order (4 bytes per found read), the number of found reads
// before every tile of 2048 reads (so that the root can split tiles independently), and the exceptions -- the reads whose
// compact result says "see the record" -- as {read, count} in read order with their hits.  3.73 bytes per read where nine
// reads in ten are found, against 4 for the compact words themselves: a position needs its 32 bits, a miss does not.
constexpr uint32_t kWireTile = kWireTileReads;  // reads per tile (2048): eight per thread, one byte of the bitmap

// x summed over the threads before this one in the block (s_w: kBlock / 64 words of LDS; two barriers)
template <class T>
__device__ __forceinline__ T block_exclusive_sum(T x, T *s_w, T &block_total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    T incl = x;
    for (int off = 1; off < 64; off <<= 1) {
        const T o = __shfl_up(incl, off);
        if (static_cast<int>(lane) >= off) incl += o;
    }
    __syncthreads();  // (s_w may still be read from an earlier call)
    if (lane == 63u) s_w[wave] = incl;
    __syncthreads();
    T before = incl - x, total = 0;
    for (uint32_t w = 0; w < kBlock / 64; w++) {
        if (w < wave) before += s_w[w];
        total += s_w[w];
    }
    block_total = total;
    return before;
}

// a thread's eight compact words (reads q0 .. q0 + 7; beyond m: "none")
__device__ __forceinline__ void wire_load8(const uint32_t *__restrict__ compact, uint64_t q0, uint64_t m, uint32_t (&c)[8])
{
    if (q0 + 8u <= m) {
        const u32x4 a = reinterpret_cast<const u32x4 *>(compact + q0)[0], b = reinterpret_cast<const u32x4 *>(compact + q0)[1];
        c[0] = a.x, c[1] = a.y, c[2] = a.z, c[3] = a.w, c[4] = b.x, c[5] = b.y, c[6] = b.z, c[7] = b.w;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) c[k] = q0 + k < m ? compact[q0 + k] : kCompactNone;
    }
}

// pass 1: per tile the found reads, the exceptions and the exceptions' hits
__global__ __launch_bounds__(kBlock) void wire_tile_counts_kernel(const uint32_t *__restrict__ compact, HitOffsets off, uint64_t m,
                                                                  unsigned long long *__restrict__ found, unsigned long long *__restrict__ see,
                                                                  unsigned long long *__restrict__ see_hits)
{
    __shared__ unsigned long long s_w[kBlock / 64];
    const uint64_t n_tiles = (m + kWireTile - 1) / kWireTile;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t q0 = tile * kWireTile + threadIdx.x * 8u;
        uint32_t c[8];
        wire_load8(compact, q0, m, c);
        unsigned long long f = 0, sq = 0, sh = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            f += c[k] < kCompactSee;
            if (c[k] == kCompactSee) {
                sq++;
                sh += off[q0 + k + 1] - off[q0 + k];
            }
        }
        // (found <= 2048 and exceptions <= 2048 travel in one word)
        unsigned long long t_fs, t_h;
        (void)block_exclusive_sum<unsigned long long>(f | (sq << 32), s_w, t_fs);
        (void)block_exclusive_sum<unsigned long long>(sh, s_w, t_h);
        if (threadIdx.x == 0) {
            found[tile] = t_fs & 0xffffffffull;
            see[tile] = t_fs >> 32;
            see_hits[tile] = t_h;
        }
    }
}

struct WireOut {
    uint8_t *bitmap;
    uint32_t *tile_found, *found_pos;
    uint64_t found_cap;
    uint32_t *exc_q, *exc_cnt;
    uint64_t exc_cap;
    uint8_t *exc_ids;
    int32_t *exc_pos;
    uint64_t exc_hits_cap;
    uint32_t *meta;  // [0] exceptions, [1] their hits, [2] found reads, [3] 0 (true numbers: what exceeds a capacity is dropped)
    // the host's form of the wire (gdx_locate_many_alloc_layout32, host_api.hip): the exceptions' hits as they are (any number
    // of texts) and the hit offset of every tile's first read, so that host threads expand tiles independently
    gdx_hit32_t *exc_hits32;  // != null: instead of exc_ids / exc_pos
    uint32_t *tile_off;       // != null: n_tiles + 1 entries
    uint64_t hits_n;          // hit slots that were stored (a step into too small a buffer leaves the rest unwritten)
    uint8_t *found_ids;       // != null: the found reads' text ids; found_pos then holds positions in that text
    const uint32_t *sentinels;
    uint32_t n_texts, shift;
};

// pass 3 (after the three tile arrays have been scanned): the wire
__global__ __launch_bounds__(kBlock) void wire_pack_kernel(const uint32_t *__restrict__ compact, HitOffsets off, const gdx_hit32_t *__restrict__ hits,
                                                           uint64_t m, const unsigned long long *__restrict__ found,
                                                           const unsigned long long *__restrict__ see,
                                                           const unsigned long long *__restrict__ see_hits, WireOut w)
{
    __shared__ unsigned long long s_w[kBlock / 64];
    __shared__ uint32_t s_tab[kTextTab + 1];
    __shared__ uint32_t s_sent[256];
    if (w.found_ids != nullptr) {
        for (uint32_t i = threadIdx.x; i < w.n_texts; i += kBlock) s_sent[i] = w.sentinels[i];
        __syncthreads();
        build_text_table(s_tab, s_sent, w.n_texts, w.shift);
        __syncthreads();
    }
    const uint64_t n_tiles = (m + kWireTile - 1) / kWireTile;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t q0 = tile * kWireTile + threadIdx.x * 8u;
        uint32_t c[8];
        wire_load8(compact, q0, m, c);
        uint32_t fbits = 0, sbits = 0;
        unsigned long long sh = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            if (c[k] < kCompactSee) fbits |= 1u << k;
            if (c[k] == kCompactSee) {
                sbits |= 1u << k;
                sh += off[q0 + k + 1] - off[q0 + k];
            }
        }
        unsigned long long t0, t1;
        const unsigned long long fs = block_exclusive_sum<unsigned long long>(static_cast<unsigned long long>(__popc(fbits)) |
                                                                              (static_cast<unsigned long long>(__popc(sbits)) << 32), s_w, t0);
        unsigned long long h_at = see_hits[tile] + block_exclusive_sum<unsigned long long>(sh, s_w, t1);
        if (q0 < m) w.bitmap[tile * (kWireTile / 8u) + threadIdx.x] = static_cast<uint8_t>(fbits);
        if (threadIdx.x == 0) {
            w.tile_found[tile] = static_cast<uint32_t>(found[tile]);
            if (w.tile_off != nullptr) w.tile_off[tile] = static_cast<uint32_t>(off[tile * kWireTile]);
        }
        uint64_t f_at = found[tile] + (fs & 0xffffffffull), e_at = see[tile] + (fs >> 32);
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            if (fbits & (1u << k)) {
                if (f_at < w.found_cap) {
                    if (w.found_ids != nullptr) {
                        const uint32_t g = c[k], b = g >> w.shift;
                        uint32_t lo = s_tab[b], hi = s_tab[b + 1];
                        while (lo < hi) {  // smallest t in [lo, hi] with g <= sentinels[t]
                            const uint32_t mid = (lo + hi) >> 1;
                            if (s_sent[mid] < g) lo = mid + 1u;
                            else hi = mid;
                        }
                        w.found_ids[f_at] = static_cast<uint8_t>(lo);
                        w.found_pos[f_at] = lo == 0u ? g : g - s_sent[lo - 1u] - 1u;
                    } else {
                        w.found_pos[f_at] = c[k];
                    }
                }
                f_at++;
            } else if (sbits & (1u << k)) {
                const uint64_t q = q0 + k, a = off[q];
                const uint64_t cnt = off[q + 1] - a;
                if (e_at < w.exc_cap) {
                    w.exc_q[e_at] = static_cast<uint32_t>(q);
                    w.exc_cnt[e_at] = static_cast<uint32_t>(cnt);
                }
                e_at++;
                for (uint64_t i = 0; i < cnt; i++)
                    if (h_at + i < w.exc_hits_cap && a + i < w.hits_n) {
                        const gdx_hit32_t h = hits[a + i];
                        if (w.exc_hits32 != nullptr) {
                            w.exc_hits32[h_at + i] = h;
                        } else {
                            w.exc_ids[h_at + i] = static_cast<uint8_t>(h.text_id);
                            w.exc_pos[h_at + i] = static_cast<int32_t>(h.position);
                        }
                    }
                h_at += cnt;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        w.tile_found[n_tiles] = static_cast<uint32_t>(found[n_tiles]);
        if (w.tile_off != nullptr) w.tile_off[n_tiles] = static_cast<uint32_t>(off[m]);
        w.meta[0] = static_cast<uint32_t>(see[n_tiles]);
        w.meta[1] = static_cast<uint32_t>(see_hits[n_tiles] < 0xffffffffull ? see_hits[n_tiles] : 0xffffffffull);
        w.meta[2] = static_cast<uint32_t>(found[n_tiles]);
        w.meta[3] = 0u;
    }
}

// the root's side: one shard's bitmap + positions -> per read a text id byte and the position in that text (-1 = no
// occurrence, -2 = an exception: the reads listed in exc_q, sorted), the form gdx_compact_split_hits_dev produces
__global__ __launch_bounds__(kBlock) void wire_split_kernel(const uint8_t *__restrict__ bitmap, const uint32_t *__restrict__ tile_found,
                                                            const uint32_t *__restrict__ found_pos, uint64_t found_cap, uint64_t m,
                                                            const uint32_t *__restrict__ exc_q, const uint32_t *__restrict__ meta,
                                                            uint64_t exc_cap, const uint32_t *__restrict__ sentinels_g, uint32_t n_texts,
                                                            uint32_t shift, uint8_t *__restrict__ ids, int32_t *__restrict__ pos)
{
    __shared__ uint32_t s_w[kBlock / 64];
    __shared__ uint32_t s_tab[kTextTab + 1];
    __shared__ uint32_t s_sent[256];
    for (uint32_t i = threadIdx.x; i < n_texts; i += kBlock) s_sent[i] = sentinels_g[i];
    __syncthreads();
    build_text_table(s_tab, s_sent, n_texts, shift);
    __syncthreads();
    const uint64_t n_exc = meta[0] < exc_cap ? meta[0] : exc_cap;
    const uint64_t n_tiles = (m + kWireTile - 1) / kWireTile;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t q0 = tile * kWireTile + threadIdx.x * 8u;
        const uint32_t fbits = q0 < m ? bitmap[tile * (kWireTile / 8u) + threadIdx.x] : 0u;
        uint32_t total;
        uint64_t at = tile_found[tile] + block_exclusive_sum<uint32_t>(static_cast<uint32_t>(__popc(fbits)), s_w, total);
        uint32_t id[8];
        int32_t p[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            id[k] = 0u;
            p[k] = -1;
            if (fbits & (1u << k)) {
                const uint32_t g = at < found_cap ? found_pos[at] : 0u;
                at++;
                const uint32_t b = g >> shift;
                uint32_t lo = s_tab[b], hi = s_tab[b + 1];
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_sent[mid] < g) lo = mid + 1u;
                    else hi = mid;
                }
                id[k] = lo;
                p[k] = static_cast<int32_t>(lo == 0u ? g : g - s_sent[lo - 1u] - 1u);
            }
        }
        if (q0 + 8u <= m) {
            *reinterpret_cast<uint2 *>(ids + q0) = make_uint2(id[0] | (id[1] << 8) | (id[2] << 16) | (id[3] << 24),
                                                              id[4] | (id[5] << 8) | (id[6] << 16) | (id[7] << 24));
            reinterpret_cast<int4 *>(pos + q0)[0] = make_int4(p[0], p[1], p[2], p[3]);
            reinterpret_cast<int4 *>(pos + q0)[1] = make_int4(p[4], p[5], p[6], p[7]);
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8; k++)
                if (q0 + k < m) {
                    ids[q0 + k] = static_cast<uint8_t>(id[k]);
                    pos[q0 + k] = p[k];
                }
        }
        if (n_exc != 0) {  // the exceptions among this tile's reads
            __syncthreads();
            const uint64_t t_lo = tile * kWireTile, t_hi = t_lo + kWireTile;
            uint64_t lo = 0, hi = n_exc;
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (exc_q[mid] < t_lo) lo = mid + 1;
                else hi = mid;
            }
            for (uint64_t e = lo + threadIdx.x; e < n_exc; e += kBlock) {
                const uint64_t q = exc_q[e];
                if (q >= t_hi || q >= m) break;
                ids[q] = 0;
                pos[q] = -2;
            }
        }
    }
}

// the queries whose compact result says "see the record", listed in no particular order (the caller sorts the few there
// are); *n counts all of them, whatever the list holds.  Four queries per thread, one atomic per wavefront that has any.
__global__ __launch_bounds__(kBlock) void compact_exceptions_kernel(const uint32_t *__restrict__ compact, uint64_t m,
                                                                    uint32_t *__restrict__ list, uint64_t capacity,
                                                                    unsigned long long *__restrict__ n)
{
    const uint64_t quads = (m + 3u) / 4u;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t i0 = static_cast<uint64_t>(blockIdx.x) * kBlock; i0 < quads; i0 += stride) {  // (block-uniform trip count)
        const uint64_t i = i0 + threadIdx.x;
        uint32_t c[4] = {0u, 0u, 0u, 0u};
        if (i < quads) {
            if (i * 4u + 4u <= m) {
                const u32x4 v = reinterpret_cast<const u32x4 *>(compact)[i];
                c[0] = v.x, c[1] = v.y, c[2] = v.z, c[3] = v.w;
            } else {
                for (uint32_t k = 0; i * 4u + k < m; k++) c[k] = compact[i * 4u + k];
            }
        }
        const uint32_t mine = (c[0] == kCompactSee) + (c[1] == kCompactSee) + (c[2] == kCompactSee) + (c[3] == kCompactSee);
        if (__ballot(mine != 0u) == 0ull) continue;
        uint32_t incl = mine;
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        unsigned long long base = 0;
        if (lane == 63u) base = atomicAdd(n, static_cast<unsigned long long>(incl));
        base = __shfl(base, 63);
        uint64_t at = base + incl - mine;
        for (uint32_t k = 0; k < 4u; k++)
            if (c[k] == kCompactSee) {
                if (at < capacity) list[at] = static_cast<uint32_t>(i * 4u + k);
                at++;
            }
    }
}
}  // namespace

void launch_compact_exceptions(const uint32_t *d_compact, uint64_t m, uint32_t *d_list, uint64_t capacity,
                               unsigned long long *d_n, hipStream_t stream)
{
    GDX_HIP(hipMemsetAsync(d_n, 0, sizeof(unsigned long long), stream));
    if (m == 0) return;
    hipLaunchKernelGGL(compact_exceptions_kernel, dim3(grid_for_items((m + 3) / 4)), dim3(kBlock), 0, stream, d_compact, m,
                       d_list, capacity, d_n);
}

void launch_compact_split(const IndexView &ix, const uint32_t *d_compact, uint64_t m, uint8_t *d_ids, int32_t *d_pos,
                          hipStream_t stream)
{
    if (m == 0) return;
    hipLaunchKernelGGL(compact_split_kernel, dim3(grid_for_items((m + 3) / 4)), dim3(kBlock), 0, stream, d_compact, m,
                       ix.sentinels, ix.n_texts, d_ids, d_pos);
}

size_t wire_pack_workspace_bytes(uint64_t m) { return 3 * ((m + kWireTile - 1) / kWireTile + 1) * sizeof(unsigned long long); }

void launch_wire_pack(const uint32_t *d_compact, const void *d_hit_offsets, bool narrow_offsets, const gdx_hit32_t *d_hits, uint64_t m,
                      uint8_t *d_bitmap, uint32_t *d_tile_found, uint32_t *d_found_pos, uint64_t found_cap, uint32_t *d_exc_q,
                      uint32_t *d_exc_cnt, uint64_t exc_cap, uint8_t *d_exc_ids, int32_t *d_exc_pos, uint64_t exc_hits_cap,
                      uint32_t *d_meta, void *d_workspace, hipStream_t stream, const WireHostForm *hf)
{
    gdx_hit32_t *d_exc_hits32 = hf ? hf->d_exc_hits32 : nullptr;
    uint32_t *d_tile_off = hf ? hf->d_tile_off : nullptr;
    uint8_t *d_found_ids = hf ? hf->d_found_ids : nullptr;
    if (d_found_ids != nullptr && (hf->ix == nullptr || hf->ix->n_texts > 256u))
        fail(GDX_ERR_INVALID_ARGUMENT, "internal: text ids as bytes need the index and at most 256 texts");
    uint32_t shift = 0;
    if (d_found_ids != nullptr)
        while ((static_cast<uint64_t>(hf->ix->n) >> shift) >= kTextTab) shift++;
    const uint64_t n_tiles = (m + kWireTile - 1) / kWireTile;
    unsigned long long *found = static_cast<unsigned long long *>(d_workspace), *see = found + n_tiles + 1, *see_hits = see + n_tiles + 1;
    const HitOffsets off{d_hit_offsets, narrow_offsets ? 1u : 0u};
    const WireOut w{d_bitmap, d_tile_found, d_found_pos, found_cap, d_exc_q, d_exc_cnt, exc_cap, d_exc_ids, d_exc_pos, exc_hits_cap, d_meta,
                    d_exc_hits32, d_tile_off, hf ? hf->hits_stored : ~0ull, d_found_ids, d_found_ids ? hf->ix->sentinels : nullptr,
                    d_found_ids ? hf->ix->n_texts : 0u, shift};
    if (m == 0) {
        GDX_HIP(hipMemsetAsync(d_meta, 0, 4 * sizeof(uint32_t), stream));
        GDX_HIP(hipMemsetAsync(d_tile_found, 0, sizeof(uint32_t), stream));
        if (d_tile_off != nullptr) GDX_HIP(hipMemsetAsync(d_tile_off, 0, sizeof(uint32_t), stream));
        return;
    }
    const unsigned grid = static_cast<unsigned>(n_tiles < 2048 ? n_tiles : 2048);
    hipLaunchKernelGGL(wire_tile_counts_kernel, dim3(grid), dim3(kBlock), 0, stream, d_compact, off, m, found, see, see_hits);
    hipLaunchKernelGGL(scan2_sums_kernel, dim3(3), dim3(1024), 0, stream, found, n_tiles, static_cast<unsigned long long *>(nullptr), n_tiles + 1);
    hipLaunchKernelGGL(wire_pack_kernel, dim3(grid), dim3(kBlock), 0, stream, d_compact, off, d_hits, m, found, see, see_hits, w);
}

void launch_wire_split(const IndexView &ix, const uint8_t *d_bitmap, const uint32_t *d_tile_found, const uint32_t *d_found_pos,
                       uint64_t found_cap, uint64_t m, const uint32_t *d_exc_q, const uint32_t *d_meta, uint64_t exc_cap, uint8_t *d_ids,
                       int32_t *d_pos, hipStream_t stream)
{
    if (m == 0) return;
    const uint64_t n_tiles = (m + kWireTile - 1) / kWireTile;
    uint32_t shift = 0;
    while ((static_cast<uint64_t>(ix.n) >> shift) >= kTextTab) shift++;
    hipLaunchKernelGGL(wire_split_kernel, dim3(static_cast<unsigned>(n_tiles < 2048 ? n_tiles : 2048)), dim3(kBlock), 0, stream, d_bitmap,
                       d_tile_found, d_found_pos, found_cap, m, d_exc_q, d_meta, exc_cap, ix.sentinels, ix.n_texts, shift, d_ids, d_pos);
}

void launch_offsets_hits(const IndexView &ix, const uint4 *d_rec, const uint32_t *d_compact, uint64_t nq, uint32_t max_hits, bool take,
                         const void *d_scan_workspace, void *d_hit_offsets, bool narrow, uint64_t total_hits, uint64_t rest_hits,
                         void *d_hits, void *d_workspace, hipStream_t stream, const QueryOptions &qo)
{
    // few open slots: the scan pass stores the compactly answered hits and flags the locate chunks with open slots, locate
    // visits only those; many (reads from repeats, short reads): the scan writes offsets only and locate streams over all slots
    const bool sparse = d_compact != nullptr && rest_hits * 16 <= total_hits;
    uint8_t *flags = (sparse && rest_hits != 0) ? static_cast<uint8_t *>(d_workspace) + locate_chunk_flags_offset(total_hits) : nullptr;
    launch_scan_offsets_store(ix, d_rec, d_compact, nq, max_hits, take, d_scan_workspace, static_cast<uint64_t *>(d_hit_offsets), d_hits,
                              total_hits, false, stream, sparse, flags, narrow, false, locate_entry_sa(ix, qo));
    if (rest_hits != 0 || !sparse)
        launch_locate(ix, nullptr, nullptr, nq, static_cast<const uint64_t *>(d_hit_offsets), total_hits, d_hits, false, d_workspace,
                      stream, nullptr, nullptr, qo, d_rec, false, d_compact, sparse, flags, narrow);
}

void launch_locate_step(const IndexView &ix, const LocateStep &step, hipStream_t st, const QueryOptions &qo)
{
    const uint64_t nq = step.call.nq;
    unsigned long long *totals = step.d_totals;
    const uint32_t *d_compact = step.call.d_compact;
    // the scan pass stores the hits the compact results answer and flags the locate chunks that hold other slots
    const bool store = d_compact != nullptr && step.hits_capacity != 0;
    uint8_t *flags = store ? static_cast<uint8_t *>(step.d_workspace) + locate_chunk_flags_offset(step.hits_capacity) : nullptr;
    ZeroSet zero;
    zero.add(totals, 2 * sizeof(unsigned long long));
    if (flags != nullptr) zero.add(locate_flags_region(flags), locate_flags_region_bytes(step.hits_capacity));
    if (nq == 0) {
        zero.add(step.d_hit_offsets, step.narrow ? sizeof(uint32_t) : sizeof(uint64_t));
        zero.flush(st);
        if (step.event_after_search) GDX_HIP(hipEventRecord(step.event_after_search, st));
        return;
    }
    SearchCall c = step.call;
    c.mode = 1;
    bool folded = false;
    if (d_compact != nullptr && !(step.take && step.max_hits != 0u)) {  // (the search's own totals count a capped query as none)
        c.d_tile_sums = static_cast<unsigned long long *>(step.d_scan_workspace);
        c.d_tile_rest = totals + 1;
        c.tile_max_hits = step.max_hits;
        c.tile_sums_done = &folded;
    }
    c.also_zero = &zero;
    launch_search_call(ix, c, st, qo);
    GDX_HIP(hipGetLastError());
    if (folded)
        launch_scan_totals_finish(step.d_scan_workspace, nq, totals, st);
    else  // (zeroes the totals again and counts from the results)
        launch_scan_totals(c.d_rec, d_compact, nq, step.max_hits, step.take, step.d_scan_workspace, totals, st);
    if (step.event_after_search) GDX_HIP(hipEventRecord(step.event_after_search, st));
    launch_scan_offsets_store(ix, c.d_rec, d_compact, nq, step.max_hits, step.take, step.d_scan_workspace,
                              static_cast<uint64_t *>(step.d_hit_offsets), step.d_hits, step.hits_capacity, false, st, store, flags,
                              step.narrow, true, locate_entry_sa(ix, qo), totals);
    if (step.hits_capacity != 0)
        launch_locate(ix, nullptr, nullptr, nq, static_cast<const uint64_t *>(step.d_hit_offsets), step.hits_capacity, step.d_hits, false,
                      step.d_workspace, st, nullptr, nullptr, qo, c.d_rec, false, d_compact, store, flags, step.narrow, totals);
}

size_t count_offsets_temp_bytes(uint64_t m)
{
    size_t bytes = 0;
    CountIterator in(rocprim::counting_iterator<uint64_t>(0), CountAt{nullptr, m});
    uint64_t *out = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, in, out, uint64_t(0), static_cast<size_t>(m + 1), rocprim::plus<uint64_t>());
    return bytes;
}

void launch_count_offsets(const uint32_t *d_counts, uint64_t m, uint64_t *d_offsets, void *d_temp, size_t temp_bytes,
                          hipStream_t stream)
{
    CountIterator in(rocprim::counting_iterator<uint64_t>(0), CountAt{d_counts, m});
    GDX_HIP(rocprim::exclusive_scan(d_temp, temp_bytes, in, d_offsets, uint64_t(0), static_cast<size_t>(m + 1),
                                    rocprim::plus<uint64_t>(), stream));
}

void launch_unpack_records(const uint4 *d_rec, uint64_t m, uint32_t *d_counts, uint8_t *d_status, hipStream_t stream,
                           const uint32_t *d_compact, unsigned long long *d_any_status)
{
    if (m == 0) return;
    hipLaunchKernelGGL(unpack_records_kernel, dim3(grid_for_items(m)), dim3(kBlock), 0, stream, d_rec, m, d_counts, d_status,
                       d_compact, d_any_status);
}

// the first-query table of the chunks (n_chunks + 1 entries), then the chunk flags (locate_chunk_flags_offset)
size_t locate_workspace_bytes(uint64_t total_hits)
{
    return locate_chunk_flags_offset(total_hits) + align_up(locate_chunk_flags_bytes(total_hits) + 4, 256) + 256;  // (+ the ticket: entry n_chunks + 1 of the table)
}

void launch_locate(const IndexView &ix, const uint32_t *d_start, const uint32_t *d_end, uint64_t m,
                   const uint64_t *d_hit_offsets, uint64_t total_hits, void *d_hits, bool wide,
                   void *d_workspace, hipStream_t stream, unsigned long long *d_step_stats, const uint2 *d_hint,
                   const QueryOptions &qo, const uint4 *d_rec, bool reference_walk, const uint32_t *d_compact,
                   bool compact_stored, const uint8_t *d_chunk_flags, bool narrow_offsets, const unsigned long long *d_total)
{
    if (total_hits == 0 || m == 0) return;
    if (d_total != nullptr && d_rec == nullptr) fail(GDX_ERR_INVALID_ARGUMENT, "internal: a device-side total goes with search records");
    if (narrow_offsets && d_rec == nullptr) fail(GDX_ERR_INVALID_ARGUMENT, "internal: narrow offsets go with search records");
    const HitOffsets offs{d_hit_offsets, narrow_offsets ? 1u : 0u};
    // (rounds 1-4 kept two lock-step variants beside the chunk kernels -- one lane per hit, eight lanes per hit on pair lines --
    // behind gdx_query_options_t.locate_kernel: 3 and 5 times slower (profiles/r01/search_variants.md), reached by no
    // configuration, and their slot -> query map cost 4 bytes of workspace per hit; removed in round 5, the option is ignored)
    // GDX_LOCATE_GRID (experiments): absolute number of blocks of the locate kernel
    static const long grid_override = [] { const char *e = getenv("GDX_LOCATE_GRID"); return e ? atol(e) : 0L; }();
    // SA[row] is one fetch and every slot is open (no chunk flags): by query, no chunk table at all
    // (GDX_LOCATE_BY_QUERY=0, experiments: the stream kernel for those too)
    static const bool by_query = [] { const char *e = getenv("GDX_LOCATE_BY_QUERY"); return e == nullptr || atoi(e) != 0; }();
    const bool sa_at_hand = ix.layout == 0 && !reference_walk && qo.locate_jump_walk != 0 &&
                            (ix.sa_full != nullptr || (ix.jump != nullptr && ix.jump_bytes == 32));
    if (by_query && sa_at_hand && d_chunk_flags == nullptr) {
        uint32_t shift = 0;
        while ((static_cast<uint64_t>(ix.n) >> shift) >= kTextTab) shift++;
        const StreamView sv{ix.sa_full, ix.sa_full == nullptr ? static_cast<const uint32_t *>(ix.jump) : nullptr, ix.sentinels,
                            ix.n_texts, shift, compact_stored ? 2u : 0u};
        static const unsigned cap_q = resident_blocks(locate_by_query_kernel<false>), cap_qw = resident_blocks(locate_by_query_kernel<true>);
        const uint64_t groups = (m + 63) / 64, blocks = (groups + kBlock / 64 - 1) / (kBlock / 64);
        const unsigned bgrid = grid_override > 0 ? static_cast<unsigned>(grid_override)
                                                 : static_cast<unsigned>(std::min<uint64_t>(blocks, 4ull * (wide ? cap_qw : cap_q)));
        if (wide)
            hipLaunchKernelGGL(locate_by_query_kernel<true>, dim3(bgrid), dim3(kBlock), 0, stream, sv, d_start, offs, m, d_hint, d_rec,
                               total_hits, d_hits, d_compact, d_total);
        else
            hipLaunchKernelGGL(locate_by_query_kernel<false>, dim3(bgrid), dim3(kBlock), 0, stream, sv, d_start, offs, m, d_hint, d_rec,
                               total_hits, d_hits, d_compact, d_total);
        return;
    }
    {
        const uint64_t n_chunks = (total_hits + kLocateChunk - 1) / kLocateChunk;
        uint32_t *first = static_cast<uint32_t *>(d_workspace);  // n_chunks + 1 entries of the workspace
        uint32_t *ticket = first + n_chunks + 1;                 // (the table's spare entry: zeroed by chunk_first_query_kernel)
        hipLaunchKernelGGL(chunk_first_query_kernel, dim3(static_cast<unsigned>((n_chunks + kBlock) / kBlock)),
                           dim3(kBlock), 0, stream, offs, m, n_chunks, kLocateChunk, total_hits, first, d_total, d_chunk_flags, ticket);
        // (with chunk flags few chunks have anything to do: a grid the chip holds at once, every block looks at its share
        // of the flags first)
        const uint64_t grid_cap = d_chunk_flags != nullptr ? 2048u : 65536u;
        const unsigned qgrid = grid_override > 0 ? static_cast<unsigned>(grid_override)
                                                 : static_cast<unsigned>(n_chunks < grid_cap ? n_chunks : grid_cap);
#define GDX_LOCATE_Q(TABLE, WIDE, JW)                                                                                   \
    hipLaunchKernelGGL((locate_queue_kernel<TABLE, WIDE, JW>), dim3(qgrid), dim3(kBlock), 0, stream, lv, d_start, offs, \
                       m, first, d_hint, d_rec, total_hits, d_hits, d_step_stats, d_compact, d_chunk_flags, d_total)
        // the walk goes through the jump table when there is one with at least two levels, unless the caller
        // counts the reference's own walk steps (reference_walk) or switched it off (QueryOptions::locate_jump_walk)
        const LocateView lv{ix.lines, ix.sb_offsets, ix.g_planes, ix.g_block_off, ix.jump, ix.sa_full, ix.count, ix.sa_samples,
                            ix.border_keys, ix.border_vals, ix.sentinels, ix.sb_stride, ix.jump_bytes, ix.n_texts,
                            ix.sa_inv, ix.sa_rot, ix.sa_limit, ix.sigma, ix.nbits, compact_stored ? 2u : 0u,
                            ix.g_kind, ix.g_wpb, ix.g_used, ix.g_sb};
        const bool jump_walk = ix.layout == 0 && ix.jump != nullptr && ix.jump_bytes >= 16 && !reference_walk &&
                               qo.locate_jump_walk != 0;
        // SA[row] inside the entries, or as an array of its own: no walk at all
        const bool entry_sa = (jump_walk && ix.jump_bytes == 32) ||
                              (ix.layout == 0 && ix.sa_full != nullptr && !reference_walk && qo.locate_jump_walk != 0);
        if (entry_sa) {  // (nothing walks: a caller's step statistics stay zero)
            uint32_t shift = 0;
            while ((static_cast<uint64_t>(ix.n) >> shift) >= kTextTab) shift++;
            const StreamView sv{ix.sa_full, ix.sa_full == nullptr ? static_cast<const uint32_t *>(ix.jump) : nullptr, ix.sentinels,
                                ix.n_texts, shift, compact_stored ? 2u : 0u};
            // a grid the chip holds at once (the text-id table is built once per block); every block strides over the chunks
            static const unsigned cap_s = resident_blocks(locate_stream_kernel<false>), cap_sw = resident_blocks(locate_stream_kernel<true>);
            const unsigned s_cap = wide ? cap_sw : cap_s;
            const unsigned sgrid = grid_override > 0 ? static_cast<unsigned>(grid_override)
                                                     : static_cast<unsigned>(n_chunks < s_cap ? n_chunks : s_cap);
            if (wide)
                hipLaunchKernelGGL(locate_stream_kernel<true>, dim3(sgrid), dim3(kBlock), 0, stream, sv, d_start, offs, m, first, d_hint,
                                   d_rec, total_hits, d_hits, d_compact, d_chunk_flags, d_total, ticket);
            else
                hipLaunchKernelGGL(locate_stream_kernel<false>, dim3(sgrid), dim3(kBlock), 0, stream, sv, d_start, offs, m, first, d_hint,
                                   d_rec, total_hits, d_hits, d_compact, d_chunk_flags, d_total, ticket);
        } else if (ix.layout == 0) {
            if (wide && jump_walk) GDX_LOCATE_Q(LineTable, true, true);
            else if (wide) GDX_LOCATE_Q(LineTable, true, false);
            else if (jump_walk) GDX_LOCATE_Q(LineTable, false, true);
            else GDX_LOCATE_Q(LineTable, false, false);
        } else {
            if (wide) GDX_LOCATE_Q(GenericTable, true, false);
            else GDX_LOCATE_Q(GenericTable, false, false);
        }
#undef GDX_LOCATE_Q
    }
}

}  // namespace gdx
