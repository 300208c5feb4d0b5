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

__global__ __launch_bounds__(kBlock) void mark_heads_kernel(const uint32_t *__restrict__ start,
                                                            const uint32_t *__restrict__ end, uint64_t m,
                                                            const uint64_t *__restrict__ hit_offsets,
                                                            uint32_t *__restrict__ heads)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; q < m; q += stride) {
        if (end[q] != start[q]) heads[hit_offsets[q]] = static_cast<uint32_t>(q) + 1u;
    }
}

// sampled_suffix_array.rs:110-138 for one suffix-array index: concatenated-text position of SA[i]
template <class Table>
__device__ __forceinline__ uint32_t walk_to_sample(const IndexView &ix, const uint32_t *s_count, uint32_t i,
                                                   uint32_t &steps)
{
    steps = 0;
    for (;;) {
        // :118 while i % sampling_rate != 0
        uint32_t slot;
        if (sampled_slot(ix, i, slot)) return ix.sa_samples[slot] + steps;  // :133-136
        uint32_t r;
        const uint32_t c = Table::symbol_and_rank(ix, i, r);
        if (c == 0) {  // :121-126 BWT sentinel: the walk reached the start of a text
            const uint32_t b = lower_bound_u32(ix.border_keys, ix.n_texts, i);
            return ix.border_vals[b] + steps;
        }
        i = s_count[c] + r;  // lf_mapping_step lib.rs:273-275
        steps++;
    }
}

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

template <class Table, bool kWide>
__global__ __launch_bounds__(kBlock) void locate_kernel(IndexView ix, const uint32_t *__restrict__ start,
                                                        const uint64_t *__restrict__ hit_offsets,
                                                        const uint32_t *__restrict__ query_of_hit,
                                                        uint64_t total, void *__restrict__ hits_out,
                                                        unsigned long long *__restrict__ step_stats)
{
    __shared__ uint32_t s_count[257];
    for (int i = threadIdx.x; i <= ix.sigma; i += kBlock) s_count[i] = ix.count[i];
    __syncthreads();
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    uint32_t walk_steps = 0;  // only reported through step_stats (bench accounting)
    for (uint64_t h = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; h < total; h += stride) {
        const uint32_t q = query_of_hit[h] - 1u;
        const uint32_t i = start[q] + static_cast<uint32_t>(h - hit_offsets[q]);  // SA index of this hit
        uint32_t steps;
        const uint32_t pos = walk_to_sample<Table>(ix, s_count, i, steps);
        walk_steps += steps;
        store_hit<kWide>(ix, pos, hits_out, h);
    }
    if (step_stats) atomicAdd(step_stats, static_cast<unsigned long long>(walk_steps));
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
                                                                   const uint8_t *__restrict__ chunk_flags)
{
    const uint64_t c = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
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
// reading the offsets again, the scan is a wavefront scan with four barriers per chunk, the loads of four slots are issued
// together, and text ids come from the coarse table.  Consecutive slots of one query are consecutive rows: their SA loads
// and hit stores are coalesced, their record loads one broadcast.
struct StreamView {
    const uint32_t *sa_full, *jump32;  // SA[row], or word 6 of the 32-byte jump entry of the row
    const uint32_t *sentinels;
    uint32_t n_texts, tab_shift;
    uint32_t skip_single;  // LocateView::skip_single
};

template <bool kWide>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(6, 8))) void locate_stream_kernel(
    StreamView sv, const uint32_t *__restrict__ start, HitOffsets hit_offsets, uint64_t m, const uint32_t *__restrict__ first_query,
    const uint2 *__restrict__ hint, const uint4 *__restrict__ rec, uint64_t total, void *__restrict__ hits_out,
    const uint32_t *__restrict__ compact, const uint8_t *__restrict__ chunk_flags, const unsigned long long *__restrict__ d_total)
{
    if (d_total != nullptr) {
        const uint64_t t = *d_total;
        total = t < total ? t : total;
    }
    const uint64_t n_chunks = (total + kLocateChunk - 1) / kLocateChunk;
    if (chunk_flags != nullptr) {  // nothing flagged among this block's chunks (the usual case on a text without repeats): done
        int any = 0;
        for (uint64_t ch = blockIdx.x + static_cast<uint64_t>(threadIdx.x) * gridDim.x; ch < n_chunks; ch += static_cast<uint64_t>(kBlock) * gridDim.x)
            any |= chunk_flags[ch] != 0;
        if (!__syncthreads_or(any)) return;
    }
    constexpr uint32_t kPer = kLocateChunk / kBlock;  // slots per thread
    constexpr uint32_t kLdsTexts = 256;
    __shared__ uint32_t s_map[kLocateChunk];  // (query of the slot, relative to the chunk's first, + 1) << 11 | the slot of its head
    __shared__ uint32_t s_tab[kTextTab + 1];
    __shared__ uint32_t s_sentinels[kLdsTexts];
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_carry;  // slots of the chunk's first query that lie in earlier chunks
    const uint32_t *sentinels = sv.sentinels;
    if (sv.n_texts <= kLdsTexts) {
        for (uint32_t i = threadIdx.x; i < sv.n_texts; i += kBlock) s_sentinels[i] = sv.sentinels[i];
        sentinels = s_sentinels;
        __syncthreads();
    }
    build_text_table(s_tab, sentinels, sv.n_texts, sv.tab_shift);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        if (chunk_flags != nullptr && chunk_flags[chunk] == 0) continue;  // (block-uniform)
        const uint64_t base = chunk * kLocateChunk;
        const uint32_t cnt = total - base < kLocateChunk ? static_cast<uint32_t>(total - base) : kLocateChunk;
        const uint32_t qa = first_query[chunk], qb = first_query[chunk + 1];
        __syncthreads();  // the previous chunk's map is read, the table is built
#pragma unroll
        for (uint32_t j = 0; j < kPer; j += 4)
            *reinterpret_cast<uint4 *>(&s_map[threadIdx.x * kPer + j]) = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (uint64_t q = static_cast<uint64_t>(qa) + threadIdx.x; q <= qb; q += kBlock) {
            const uint64_t a = hit_offsets[q], b = hit_offsets[q + 1];
            const uint64_t from = a > base ? a : base;
            if (b > from && from < base + cnt) {
                s_map[from - base] = ((static_cast<uint32_t>(q - qa) + 1u) << 11) | static_cast<uint32_t>(from - base);
                if (q == qa) s_carry = static_cast<uint32_t>(from - a);  // (the owner of slot `base`: always has slots here)
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
        // four slots at a time, a stage for all four before the next: map -> compact result -> record -> SA -> hit
#pragma unroll
        for (uint32_t half = 0; half < kPer; half += 4) {
            uint32_t q[4], within[4], c4[4], pos[4];
            bool live[4], need_sa[4];
            uint4 r[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t i = threadIdx.x + (half + j) * kBlock;
                live[j] = i < cnt;
                const uint32_t p = live[j] ? s_map[i] : (1u << 11);
                const uint32_t qrel = p >> 11;
                q[j] = qa + qrel - 1u;
                within[j] = i - (p & 2047u) + (qrel == 1u ? carry : 0u);
                c4[j] = (live[j] && compact != nullptr) ? compact[q[j]] : kCompactSee;
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                need_sa[j] = false;
                pos[j] = 0;
                if (!live[j]) continue;
                if (c4[j] < kCompactSee) {  // the position itself (skip_single 2: launch_scan_offsets_store has stored it)
                    pos[j] = c4[j];
                    live[j] = sv.skip_single != 2u;
                    continue;
                }
                if (rec != nullptr) r[j] = rec[q[j]];
                else r[j] = make_uint4(start[q[j]], 0u, 0xffffffffu, 0u);
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                if (!live[j] || c4[j] < kCompactSee) continue;
                if (sv.skip_single == 1u && hit_offsets[q[j] + 1] - hit_offsets[q[j]] == 1u) {  // in place already (scan_locate_kernel)
                    live[j] = false;
                    continue;
                }
                uint32_t row = r[j].x + within[j], back = 0;
                if (rec != nullptr) {
                    if (r[j].w & kRecResolved) {  // the search already knows the text position
                        pos[j] = r[j].z;
                        continue;
                    }
                    if (r[j].w & kRecMasked) {  // the within-th surviving row of the mask, `symbols` steps before the hit
                        uint32_t mk = r[j].z;
                        for (uint32_t t = within[j]; t > 0u; t--) mk &= mk - 1u;
                        row = r[j].x + static_cast<uint32_t>(__builtin_ctz(mk | 0x80000000u));
                        back = r[j].w & 0x1fffffu;
                    } else if (r[j].z != 0xffffffffu && r[j].y - r[j].x == 1u) {
                        row = r[j].z;
                        back = r[j].w & 0xffffffu;
                    }
                } else if (hint != nullptr && hit_offsets[q[j] + 1] - hit_offsets[q[j]] == 1u) {
                    const uint2 hv = hint[q[j]];
                    if (hv.x != 0xffffffffu && hv.y < (1u << 21)) {
                        row = hv.x;
                        back = hv.y;
                    }
                }
                need_sa[j] = true;
                pos[j] = back;
                r[j].x = row;
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (need_sa[j]) {
                    const uint32_t sa = sv.sa_full != nullptr ? sv.sa_full[r[j].x] : sv.jump32[static_cast<uint64_t>(r[j].x) * 8u + 6u];
                    pos[j] = sa - pos[j];
                }
#pragma unroll
            for (uint32_t j = 0; j < 4; j++)
                if (live[j]) store_hit_tab<kWide>(s_tab, sv.tab_shift, sentinels, pos[j], hits_out, base + threadIdx.x + (half + j) * kBlock);
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
// itself: kEntrySA.)
// (the kernel gets the few fields of the IndexView it reads -- the whole view costs SGPRs -- and its queue shares LDS
// with the slot -> query map, so that eight blocks fit a CU: it ran at 5 waves per SIMD before)
struct LocateView {
    const u32x4 *lines;
    const uint32_t *sb_offsets;
    const uint64_t *g_planes;
    const uint16_t *g_block_off;
    const void *jump;
    const uint32_t *sa_full;  // SA[row] of every row, or null (then kEntrySA reads it out of the 32-byte jump entries)
    const uint32_t *count, *sa_samples, *border_keys, *border_vals, *sentinels;
    uint32_t sb_stride, jump_bytes, n_texts, sa_inv, sa_rot, sa_limit;
    int32_t sigma, nbits;
    uint32_t skip_single;  // 1: the hits of queries with exactly one hit slot are in place already (scan_locate_kernel)
    uint32_t g_kind, g_wpb, g_used, g_sb;  // IndexView: which of the reference's table variants layout 1 is
};

// kEntrySA: the index has 32-byte jump entries, which carry SA[row] (layout.hpp): every hit is finished in phase 0 with
// one fetch of its row's entry -- or with none when the search resolved it (kRecResolved) -- and nothing ever walks.
template <class Table, bool kWide, bool kJumpWalk, bool kEntrySA>
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
    if (chunk_flags != nullptr) {  // nothing flagged among this block's chunks (the usual case on a text without repeats): done
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
            if (lv.skip_single == 1u && hit_offsets[q + 1] - first == 1u) continue;
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
                if (r.w & kRecResolved) {  // the search already knows the text position
                    store_hit<kWide>(ix, r.z, hits_out, h, sentinels);
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
            if (kEntrySA) {
                const uint32_t sa = lv.sa_full != nullptr ? lv.sa_full[row]
                                                          : static_cast<const uint32_t *>(ix.jump)[static_cast<uint64_t>(row) * 8u + 6u];
                store_hit<kWide>(ix, sa - back, hits_out, h, sentinels);
            } else if (sampled_slot(ix, row, slot)) {  // sampled_suffix_array.rs:133-136 with zero steps
                store_hit<kWide>(ix, ix.sa_samples[slot] - back, hits_out, h, sentinels);
            } else {
                // (a hinted row that is not sampled comes from the search's lazy tail: the walk starts there)
                const uint32_t k = atomicAdd(&s_n, 1u);
                s_row[k] = row;
                s_idx[k] = i | (back << 11);
            }
        }
        if (kEntrySA) continue;  // nothing was queued (the loop's first barrier orders the next chunk's LDS writes)
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
                // levels of the entry of `row` (layout.hpp; 16-byte entries here, 32-byte ones take the kEntrySA path):
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

// The same walk on pair lines, eight lanes per hit: one 128-byte fetch at row i yields bwt1[i], bwt0[i],
// LF(i) and LF(LF(i)), i.e. TWO walk steps (the second is taken only if the first did not land on a
// sampled row), so a hit costs ~1.7 line fetches + the sample instead of 3 + the sample at rate 4.
template <bool kWide>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void locate_pair_kernel(
    IndexView ix, const uint32_t *__restrict__ start, const uint64_t *__restrict__ hit_offsets,
    const uint32_t *__restrict__ query_of_hit, uint64_t total, void *__restrict__ hits_out,
    unsigned long long *__restrict__ step_stats)
{
    constexpr int kGroup = 8;
    __shared__ uint32_t s_count[257];
    for (int i = threadIdx.x; i <= ix.sigma; i += kBlock) s_count[i] = ix.count[i];
    __syncthreads();
    const uint32_t sub = threadIdx.x & 7u;
    const bool writer = sub == 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * (kBlock / kGroup);
    uint32_t walk_steps = 0;
    for (uint64_t h = static_cast<uint64_t>(blockIdx.x) * (kBlock / kGroup) + threadIdx.x / kGroup; h < total;
         h += stride) {
        const uint32_t q = query_of_hit[h] - 1u;
        uint32_t i = start[q] + static_cast<uint32_t>(h - hit_offsets[q]);
        uint32_t steps = 0, pos;
        for (;;) {
            uint32_t slot;
            if (sampled_slot(ix, i, slot)) {  // sampled_suffix_array.rs:133-136
                pos = ix.sa_samples[slot] + steps;
                break;
            }
            const u32x4 c = ix.pair_lines[(static_cast<uint64_t>(i >> kPairLineShift) << 3) + sub];
            // the symbols of row i live in the chunk of lane (i & 63) / 8
            const uint32_t bit = i & 7u;
            const uint32_t mine = (((c.x >> bit) & 1u) | (((c.x >> (8u + bit)) & 1u) << 1) |
                                   (((c.x >> (16u + bit)) & 1u) << 2) | (((c.x >> (24u + bit)) & 1u) << 3) |
                                   (((c.y >> bit) & 1u) << 4) | (((c.y >> (8u + bit)) & 1u) << 5));
            const uint32_t both = oct_sum(((i & 63u) >> 3) == sub ? mine : 0u);
            const uint32_t c1 = both & 7u, c0 = both >> 3;
            if (c1 == 0) {  // :121-126 BWT sentinel: the walk reached the start of a text
                pos = ix.border_vals[lower_bound_u32(ix.border_keys, ix.n_texts, i)] + steps;
                break;
            }
            if (c1 > 4u) {  // a symbol outside 1..4 (N): one step on the rank lines
                uint32_t r, rdummy;
                QuadLineTable::rank2(ix, c1, i, i, r, rdummy);
                i = s_count[c1] + r;
                steps++;
                continue;
            }
            // LF(c1, i) and LF(c0, LF(c1, i)) from this line (layout.hpp PairTable)
            const uint32_t f = 0xffu;
            const uint32_t n1 = ((c1 & 1u) ? 0u : f) | (((c1 & 2u) ? 0u : f) << 8) | (((c1 & 4u) ? 0u : f) << 16);
            const uint32_t t1 = c.x ^ n1;
            const uint32_t m1 = t1 & (t1 >> 8) & (t1 >> 16) & 0xffu;
            const uint32_t mask = PairTable::low_mask(i, sub);  // chunk index == lane for 8 lanes
            const bool own1 = (sub >> 1) == (c1 - 1u);
            const uint32_t i1 = oct_sum(__popc(m1 & mask) + (own1 ? ((c.y >> 16) << ((sub & 1u) * 16u)) : 0u));
            const bool sampled1 = is_sampled(ix, i1);
            if (sampled1 || c0 - 1u >= 4u) {
                i = i1;
                steps++;
                continue;
            }
            const uint32_t pair = (c0 - 1u) * 4u + (c1 - 1u);
            const uint32_t n0x = ((c0 & 1u) ? 0u : f) << 24;
            const uint32_t n0y = ((c0 & 2u) ? 0u : f) | (((c0 & 4u) ? 0u : f) << 8);
            const uint32_t t0x = c.x ^ n0x, t0y = c.y ^ n0y;
            const uint32_t m2 = m1 & (t0x >> 24) & t0y & (t0y >> 8) & 0xffu;
            const bool own2 = (pair >> 1) == sub;
            i = oct_sum(__popc(m2 & mask) + (own2 ? ((pair & 1u) ? c.w : c.z) : 0u));
            steps += 2;
        }
        walk_steps += steps;
        const uint32_t t = lower_bound_u32(ix.sentinels, ix.n_texts, pos);
        const uint32_t in_text = t == 0 ? pos : pos - ix.sentinels[t - 1] - 1u;
        if (writer) {
            if (kWide) {
                gdx_hit_t out;
                out.text_id = t;
                out.position = in_text;
                static_cast<gdx_hit_t *>(hits_out)[h] = out;
            } else {
                gdx_hit32_t out;
                out.text_id = t;
                out.position = in_text;
                static_cast<gdx_hit32_t *>(hits_out)[h] = out;
            }
        }
    }
    if (step_stats && writer) atomicAdd(step_stats, static_cast<unsigned long long>(walk_steps));
}

// ---- one pass over the search records: hit offsets AND the hit of every query that has exactly one -------------------
// The count + locate step of a read batch is search -> scan of the counts -> locate.  With resolved records (32-byte jump
// entries carry SA[row]) the last two are streaming passes over the same records: an exclusive scan that reads them
// and writes the offsets, then a kernel that reads records and offsets again and writes the hits.  This kernel does both
// at once -- a single-pass scan with decoupled look-back (tiles of 2048 queries take tickets, publish {aggregate |
// inclusive prefix} in one 64-bit word, and look back over their predecessors) whose tiles then store the hit of every
// query with exactly one hit slot: resolved -> the position is in the record; otherwise one fetch of SA[row] when the
// index has it (jump entry / full suffix array).  Queries with several hits (and single hits that would need a walk) are
// only counted (totals[1]); the queue kernel fills them in afterwards with LocateView::skip_single.  Hits beyond
// hits_capacity are not stored: the caller compares totals[0] with the capacity it offered.
constexpr uint32_t kScanTile = 2048;
constexpr unsigned long long kTileAggregate = 1ull << 62, kTilePrefix = 2ull << 62, kTileValue = (1ull << 62) - 1ull;

template <bool kWide>
__global__ __launch_bounds__(kBlock) void scan_locate_kernel(LocateView lv, const uint4 *__restrict__ rec, uint64_t m,
                                                             uint32_t max_hits, uint32_t take,
                                                             uint64_t *__restrict__ hit_offsets, void *__restrict__ hits_out,
                                                             uint64_t hits_capacity, unsigned long long *__restrict__ tile_state,
                                                             uint32_t *__restrict__ ticket,
                                                             unsigned long long *__restrict__ totals)
{
    constexpr uint32_t kPer = kScanTile / kBlock;  // consecutive queries per thread (one 128-byte line of records)
    __shared__ uint32_t s_tile;
    __shared__ unsigned long long s_warp[(kScanTile / kBlock) * (kBlock / 64)];
    __shared__ unsigned long long s_prefix;
    constexpr uint32_t kLdsTexts = 256;
    __shared__ uint32_t s_sentinels[kLdsTexts];
    IndexView ix{};
    ix.sentinels = lv.sentinels;
    ix.n_texts = lv.n_texts;
    const uint32_t *sentinels = lv.n_texts <= kLdsTexts ? s_sentinels : lv.sentinels;
    if (lv.n_texts <= kLdsTexts)
        for (uint32_t i = threadIdx.x; i < lv.n_texts; i += kBlock) s_sentinels[i] = lv.sentinels[i];
    const bool have_sa = lv.sa_full != nullptr || (lv.jump != nullptr && lv.jump_bytes == 32u);
    const uint64_t n_tiles = (m + kScanTile - 1) / kScanTile;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);  // tiles are handed out in order: a tile only waits for earlier ones
        __syncthreads();
        const uint64_t tile = s_tile;
        if (tile >= n_tiles) break;
        // striped arrangement: thread t holds queries j * 256 + t of the tile (j = 0 .. 7), so that record loads, offset
        // stores and hit stores of a wavefront touch consecutive memory; the scan is done row by row
        const uint64_t q0 = tile * kScanTile + threadIdx.x;
        uint4 r[kPer];
        uint32_t c[kPer];
        unsigned long long incl[kPer];
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint64_t q = q0 + static_cast<uint64_t>(j) * kBlock;
            r[j] = q < m ? rec[q] : make_uint4(0u, 0u, 0u, 0u);
            uint32_t cnt = r[j].y - r[j].x;
            if (max_hits != 0u && cnt > max_hits) cnt = take ? max_hits : 0u;  // RecordSize
            c[j] = cnt;
            unsigned long long x = cnt;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned long long o = __shfl_up(x, off);
                if (static_cast<int>(threadIdx.x & 63u) >= off) x += o;
            }
            incl[j] = x;
            if ((threadIdx.x & 63u) == 63u) s_warp[j * (kBlock / 64) + (threadIdx.x >> 6)] = x;
        }
        __syncthreads();
        unsigned long long before[kPer], tile_total = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            unsigned long long row_before = 0, row_total = 0;
            for (uint32_t wv = 0; wv < kBlock / 64; wv++) {
                const unsigned long long v = s_warp[j * (kBlock / 64) + wv];
                if (wv < (threadIdx.x >> 6)) row_before += v;
                row_total += v;
            }
            before[j] = tile_total + row_before + incl[j] - c[j];
            tile_total += row_total;
        }
        if (threadIdx.x < 64u) {
            // (the flag travels in the same 64-bit word as the value, so relaxed device-scope atomics are all the protocol
            // needs: acquire / release at system scope made every poll an L2 invalidation on this multi-XCD chip)
            // the tile's first wavefront looks back 64 predecessors at a time: the tiles in flight all start with an
            // aggregate only, so a tile sums up to a few thousand of them before it meets one that knows its prefix (one
            // lane walking them one by one made the pass 20 ms instead of 1)
            const uint32_t lane = threadIdx.x;
            unsigned long long prefix = 0;
            if (tile != 0 && lane == 0) __hip_atomic_store(tile_state + tile, kTileAggregate | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long p = static_cast<long long>(tile) - 1 - static_cast<long long>(lane);
            for (;;) {
                unsigned long long v = kTilePrefix;  // before tile 0: an inclusive prefix of 0
                if (p >= 0) {
                    do {
                        v = __hip_atomic_load(tile_state + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((v >> 62) == 0ull);
                }
                const unsigned long long is_prefix = __ballot((v >> 62) == 2ull);
                const int first = is_prefix ? __ffsll(static_cast<long long>(is_prefix)) - 1 : 64;
                unsigned long long part = static_cast<int>(lane) <= first ? v & kTileValue : 0ull;
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
                prefix += part;
                if (is_prefix) break;
                p -= 64;
            }
            if (lane == 0) {
                __hip_atomic_store(tile_state + tile, kTilePrefix | (prefix + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_prefix = prefix;
                if (tile == n_tiles - 1) {
                    hit_offsets[m] = prefix + tile_total;
                    totals[0] = prefix + tile_total;
                }
            }
        }
        __syncthreads();
        const unsigned long long tile_base = s_prefix;
        unsigned long long rest = 0;  // hit slots this kernel leaves to the queue kernel
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint64_t q = q0 + static_cast<uint64_t>(j) * kBlock;
            const unsigned long long off = tile_base + before[j];
            if (q < m) {
                hit_offsets[q] = off;
                if (c[j] == 1u) {
                    const uint4 rr = r[j];
                    bool done = false;
                    uint32_t pos = 0;
                    if (rr.w & kRecResolved) {
                        pos = rr.z;
                        done = true;
                    } else if (have_sa) {
                        uint32_t row = rr.x, back = 0;
                        if (rr.w & kRecMasked) {
                            row = rr.x + static_cast<uint32_t>(__builtin_ctz(rr.z | 0x80000000u));
                            back = rr.w & 0x1fffffu;
                        } else if (rr.z != 0xffffffffu && rr.y - rr.x == 1u) {
                            row = rr.z;
                            back = rr.w & 0xffffffu;
                        }
                        const uint32_t sa = lv.sa_full != nullptr ? lv.sa_full[row]
                                                                  : static_cast<const uint32_t *>(lv.jump)[static_cast<uint64_t>(row) * 8u + 6u];
                        pos = sa - back;
                        done = true;
                    }
                    if (done) {
                        if (off < hits_capacity) store_hit<kWide>(ix, pos, hits_out, off, sentinels);
                    } else {
                        rest += 1u;
                    }
                } else {
                    rest += c[j];
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) rest += __shfl_xor(rest, o);
        if ((threadIdx.x & 63u) == 0 && rest != 0ull) atomicAdd(totals + 1, rest);
    }
}

unsigned grid_for_items(uint64_t items)
{
    const uint64_t blocks = (items + kBlock - 1) / kBlock;
    const uint64_t cap = 256u * 8u;
    return static_cast<unsigned>(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

size_t max_scan_temp_bytes(uint64_t total)
{
    size_t bytes = 0;
    uint32_t *p = nullptr;
    (void)rocprim::inclusive_scan(nullptr, bytes, p, p, static_cast<size_t>(total), rocprim::maximum<uint32_t>());
    return bytes;
}

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
// the second pass runs on a grid the chip holds at once (six blocks of its 80 registers per CU), every block over many tiles
static unsigned scan2_grid(uint64_t n_tiles) { return static_cast<unsigned>(n_tiles < 1536 ? n_tiles : 1536); }

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
                        pos = r.z;
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
        hipLaunchKernelGGL((scan2_tile_scan_kernel<false, false>), dim3(scan2_grid(n_tiles)), dim3(kBlock), 0, stream, f, m, sums,
                           d_hit_offsets, ScanStore{}, 0u);
        return;
    }
    RecordSizeIterator in(rocprim::counting_iterator<uint64_t>(0), RecordSize{d_rec, d_compact, m, max_hits, take});
    GDX_HIP(rocprim::exclusive_scan(d_temp, temp_bytes, in, d_hit_offsets, uint64_t(0),
                                    static_cast<size_t>(m + 1), rocprim::plus<uint64_t>(), stream));
}

// where launch_scan_offsets_store / launch_locate keep the chunk flags inside a locate workspace of locate_workspace_bytes(total)
// (behind the first-query table of the chunks, which takes n_chunks + 1 of the workspace's total x 4 bytes)
size_t locate_chunk_flags_offset(uint64_t total_hits)
{
    const uint64_t n_chunks = (total_hits + kLocateChunk - 1) / kLocateChunk;
    return align_up((n_chunks + 2) * sizeof(uint32_t), 256);
}

size_t locate_chunk_flags_bytes(uint64_t total_hits) { return (total_hits + kLocateChunk - 1) / kLocateChunk + 1; }

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
    if (d_chunk_flags != nullptr && !flags_zeroed) GDX_HIP(hipMemsetAsync(d_chunk_flags, 0, locate_chunk_flags_bytes(hits_capacity), stream));
    const unsigned grid = scan2_grid(n_tiles);
    if (!store || d_compact == nullptr || d_hits == nullptr) {
        hipLaunchKernelGGL((scan2_tile_scan_kernel<false, false>), dim3(grid), dim3(kBlock), 0, stream, f, m, sums, d_hit_offsets,
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
        hipLaunchKernelGGL((scan2_tile_scan_kernel<true, true>), dim3(grid), dim3(kBlock), 0, stream, f, m, sums, d_hit_offsets, ss, narrow);
    else
        hipLaunchKernelGGL((scan2_tile_scan_kernel<true, false>), dim3(grid), dim3(kBlock), 0, stream, f, m, sums, d_hit_offsets, ss, narrow);
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
                                                                const uint32_t *__restrict__ compact)
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
constexpr uint32_t kWireTile = 2048;  // reads per tile: eight per thread, one byte of the bitmap

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
};

// pass 3 (after the three tile arrays have been scanned): the wire
__global__ __launch_bounds__(kBlock) void wire_pack_kernel(const uint32_t *__restrict__ compact, HitOffsets off, const gdx_hit32_t *__restrict__ hits,
                                                           uint64_t m, const unsigned long long *__restrict__ found,
                                                           const unsigned long long *__restrict__ see,
                                                           const unsigned long long *__restrict__ see_hits, WireOut w)
{
    __shared__ unsigned long long s_w[kBlock / 64];
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
        if (threadIdx.x == 0) w.tile_found[tile] = static_cast<uint32_t>(found[tile]);
        uint64_t f_at = found[tile] + (fs & 0xffffffffull), e_at = see[tile] + (fs >> 32);
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            if (fbits & (1u << k)) {
                if (f_at < w.found_cap) w.found_pos[f_at] = c[k];
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
                    if (h_at + i < w.exc_hits_cap) {
                        const gdx_hit32_t h = hits[a + i];
                        w.exc_ids[h_at + i] = static_cast<uint8_t>(h.text_id);
                        w.exc_pos[h_at + i] = static_cast<int32_t>(h.position);
                    }
                h_at += cnt;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        w.tile_found[n_tiles] = static_cast<uint32_t>(found[n_tiles]);
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
                      uint32_t *d_meta, void *d_workspace, hipStream_t stream)
{
    const uint64_t n_tiles = (m + kWireTile - 1) / kWireTile;
    unsigned long long *found = static_cast<unsigned long long *>(d_workspace), *see = found + n_tiles + 1, *see_hits = see + n_tiles + 1;
    const HitOffsets off{d_hit_offsets, narrow_offsets ? 1u : 0u};
    const WireOut w{d_bitmap, d_tile_found, d_found_pos, found_cap, d_exc_q, d_exc_cnt, exc_cap, d_exc_ids, d_exc_pos, exc_hits_cap, d_meta};
    if (m == 0) {
        GDX_HIP(hipMemsetAsync(d_meta, 0, 4 * sizeof(uint32_t), stream));
        GDX_HIP(hipMemsetAsync(d_tile_found, 0, sizeof(uint32_t), stream));
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
                           const uint32_t *d_compact)
{
    if (m == 0) return;
    hipLaunchKernelGGL(unpack_records_kernel, dim3(grid_for_items(m)), dim3(kBlock), 0, stream, d_rec, m, d_counts, d_status,
                       d_compact);
}

size_t scan_locate_workspace_bytes(uint64_t m)
{
    return align_up(((m + kScanTile - 1) / kScanTile + 1) * sizeof(unsigned long long), 256) + 256;
}

void launch_scan_locate(const IndexView &ix, const uint4 *d_rec, uint64_t m, uint32_t max_hits, bool take,
                        uint64_t *d_hit_offsets, void *d_hits, uint64_t hits_capacity, bool wide, void *d_workspace,
                        unsigned long long *d_totals, hipStream_t stream)
{
    GDX_HIP(hipMemsetAsync(d_totals, 0, 2 * sizeof(unsigned long long), stream));
    if (m == 0) {
        GDX_HIP(hipMemsetAsync(d_hit_offsets, 0, sizeof(uint64_t), stream));
        return;
    }
    const uint64_t n_tiles = (m + kScanTile - 1) / kScanTile;
    const size_t state_bytes = align_up((n_tiles + 1) * sizeof(unsigned long long), 256);
    unsigned long long *state = static_cast<unsigned long long *>(d_workspace);
    uint32_t *ticket = reinterpret_cast<uint32_t *>(static_cast<char *>(d_workspace) + state_bytes);
    GDX_HIP(hipMemsetAsync(d_workspace, 0, state_bytes + 256, stream));
    const LocateView lv{ix.lines, ix.sb_offsets, ix.g_planes, ix.g_block_off, ix.jump, ix.layout == 0 ? ix.sa_full : nullptr,
                        ix.count, ix.sa_samples, ix.border_keys, ix.border_vals, ix.sentinels, ix.sb_stride,
                        ix.layout == 0 ? ix.jump_bytes : 0u, ix.n_texts, ix.sa_inv, ix.sa_rot, ix.sa_limit, ix.sigma, ix.nbits, 0u,
                        ix.g_kind, ix.g_wpb, ix.g_used, ix.g_sb};
    // a resident grid: every block takes tiles by ticket until they run out
    static const long grid_env = [] { const char *e = getenv("GDX_SCAN_GRID"); return e ? atol(e) : 0L; }();  // experiments
    const uint64_t grid_cap = grid_env > 0 ? static_cast<uint64_t>(grid_env) : 256u * 8u;
    const unsigned grid = static_cast<unsigned>(n_tiles < grid_cap ? n_tiles : grid_cap);
    if (wide)
        hipLaunchKernelGGL(scan_locate_kernel<true>, dim3(grid), dim3(kBlock), 0, stream, lv, d_rec, m, max_hits, take ? 1u : 0u,
                           d_hit_offsets, d_hits, hits_capacity, state, ticket, d_totals);
    else
        hipLaunchKernelGGL(scan_locate_kernel<false>, dim3(grid), dim3(kBlock), 0, stream, lv, d_rec, m, max_hits, take ? 1u : 0u,
                           d_hit_offsets, d_hits, hits_capacity, state, ticket, d_totals);
}

size_t locate_workspace_bytes(uint64_t total_hits)
{
    return align_up(total_hits * sizeof(uint32_t), 256) + align_up(max_scan_temp_bytes(total_hits), 256) + 256;
}

void launch_locate(const IndexView &ix, const uint32_t *d_start, const uint32_t *d_end, uint64_t m,
                   const uint64_t *d_hit_offsets, uint64_t total_hits, void *d_hits, bool wide,
                   void *d_workspace, hipStream_t stream, unsigned long long *d_step_stats, const uint2 *d_hint,
                   const QueryOptions &qo, const uint4 *d_rec, bool reference_walk, bool skip_single, const uint32_t *d_compact,
                   bool compact_stored, const uint8_t *d_chunk_flags, bool narrow_offsets, const unsigned long long *d_total)
{
    if (total_hits == 0 || m == 0) return;
    if (d_total != nullptr && d_rec == nullptr) fail(GDX_ERR_INVALID_ARGUMENT, "internal: a device-side total goes with search records");
    if (narrow_offsets && d_rec == nullptr) fail(GDX_ERR_INVALID_ARGUMENT, "internal: narrow offsets go with search records");
    const HitOffsets offs{d_hit_offsets, narrow_offsets ? 1u : 0u};
    uint32_t *heads = static_cast<uint32_t *>(d_workspace);
    void *scan_temp = static_cast<char *>(d_workspace) + align_up(total_hits * sizeof(uint32_t), 256);
    size_t scan_bytes = max_scan_temp_bytes(total_hits);

    static const int env_variant = [] {
        const char *e = getenv("GDX_LOCATE_VARIANT");
        return !e ? 0 : (e[0] == 'l' ? 1 : (e[0] == 'p' ? 2 : 0));
    }();
    int variant = (qo.locate_variant >= 0 && qo.locate_variant <= 2) ? qo.locate_variant : env_variant;
    if (d_rec != nullptr) variant = 0;  // only the queue kernel reads search records
    if (variant != 0) {  // the lock-step variants map hit slots to queries with head marks + a max-scan over all hits
        GDX_HIP(hipMemsetAsync(heads, 0, total_hits * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(mark_heads_kernel, dim3(grid_for_items(m)), dim3(kBlock), 0, stream, d_start, d_end, m,
                           d_hit_offsets, heads);
        GDX_HIP(rocprim::inclusive_scan(scan_temp, scan_bytes, heads, heads, static_cast<size_t>(total_hits),
                                        rocprim::maximum<uint32_t>(), stream));
    }
    // GDX_LOCATE_GRID (experiments): absolute number of blocks of the walk kernel
    static const long grid_override = [] { const char *e = getenv("GDX_LOCATE_GRID"); return e ? atol(e) : 0L; }();
    const unsigned grid = grid_override > 0 ? static_cast<unsigned>(grid_override) : grid_for_items(total_hits);
#define GDX_LOCATE(TABLE, WIDE)                                                                              \
    hipLaunchKernelGGL((locate_kernel<TABLE, WIDE>), dim3(grid), dim3(kBlock), 0, stream, ix, d_start, \
                       d_hit_offsets, heads, total_hits, d_hits, d_step_stats)
    // GDX_LOCATE_VARIANT: queue (default) = locate_queue_kernel; lane = one lane per hit in lock-step; pair = 8 lanes
    // per hit on pair lines, two walk steps per fetch.  Measured per 90 M hits at rate 4 (search_variants.md):
    // lane 10.0 ms, pair 16.6 ms (latency-bound with 8x fewer hits in flight); also tried: visiting the hits in
    // suffix-array order after a radix sort of (row, slot) pairs, 12.5 ms including the sort.
    if (variant == 0) {
        const uint64_t n_chunks = (total_hits + kLocateChunk - 1) / kLocateChunk;
        uint32_t *first = heads;  // n_chunks entries of the workspace
        hipLaunchKernelGGL(chunk_first_query_kernel, dim3(static_cast<unsigned>((n_chunks + kBlock) / kBlock)),
                           dim3(kBlock), 0, stream, offs, m, n_chunks, kLocateChunk, total_hits, first, d_total, d_chunk_flags);
        // (with chunk flags few chunks have anything to do: a grid the chip holds at once, every block looks at its share
        // of the flags first)
        const uint64_t grid_cap = d_chunk_flags != nullptr ? 2048u : 65536u;
        const unsigned qgrid = grid_override > 0 ? static_cast<unsigned>(grid_override)
                                                 : static_cast<unsigned>(n_chunks < grid_cap ? n_chunks : grid_cap);
#define GDX_LOCATE_Q(TABLE, WIDE, JW)                                                                                     \
    do {                                                                                                                  \
        if (entry_sa)                                                                                                     \
            hipLaunchKernelGGL((locate_queue_kernel<TABLE, WIDE, false, true>), dim3(qgrid), dim3(kBlock), 0, stream, lv, \
                               d_start, offs, m, first, d_hint, d_rec, total_hits, d_hits, d_step_stats,                  \
                               d_compact, d_chunk_flags, d_total);                                                        \
        else                                                                                                              \
            hipLaunchKernelGGL((locate_queue_kernel<TABLE, WIDE, JW, false>), dim3(qgrid), dim3(kBlock), 0, stream, lv,   \
                               d_start, offs, m, first, d_hint, d_rec, total_hits, d_hits, d_step_stats,                  \
                               d_compact, d_chunk_flags, d_total);                                                        \
    } while (0)
        // the walk goes through the jump table when there is one with at least two levels, unless the caller
        // counts the reference's own walk steps (reference_walk) or switched it off (QueryOptions::locate_jump_walk)
        const LocateView lv{ix.lines, ix.sb_offsets, ix.g_planes, ix.g_block_off, ix.jump, ix.sa_full, ix.count, ix.sa_samples,
                            ix.border_keys, ix.border_vals, ix.sentinels, ix.sb_stride, ix.jump_bytes, ix.n_texts,
                            ix.sa_inv, ix.sa_rot, ix.sa_limit, ix.sigma, ix.nbits, compact_stored ? 2u : (skip_single ? 1u : 0u),
                            ix.g_kind, ix.g_wpb, ix.g_used, ix.g_sb};
        const bool jump_walk = ix.layout == 0 && ix.jump != nullptr && ix.jump_bytes >= 16 && !reference_walk &&
                               qo.locate_jump_walk != 0;
        // SA[row] inside the entries, or as an array of its own: no walk at all
        const bool entry_sa = (jump_walk && ix.jump_bytes == 32) ||
                              (ix.layout == 0 && ix.sa_full != nullptr && !reference_walk && qo.locate_jump_walk != 0);
        static const bool env_no_stream = getenv("GDX_LOCATE_NO_STREAM") != nullptr;  // debug: the queue kernel on every index
        if (entry_sa && d_step_stats == nullptr && !env_no_stream) {
            uint32_t shift = 0;
            while ((static_cast<uint64_t>(ix.n) >> shift) >= kTextTab) shift++;
            const StreamView sv{ix.sa_full, ix.sa_full == nullptr ? static_cast<const uint32_t *>(ix.jump) : nullptr, ix.sentinels,
                                ix.n_texts, shift, compact_stored ? 2u : (skip_single ? 1u : 0u)};
            // a grid the chip holds at once (the text-id table is built once per block); every block strides over the chunks
            const unsigned sgrid = grid_override > 0 ? static_cast<unsigned>(grid_override)
                                                     : static_cast<unsigned>(n_chunks < 2048 ? n_chunks : 2048);
            if (wide)
                hipLaunchKernelGGL(locate_stream_kernel<true>, dim3(sgrid), dim3(kBlock), 0, stream, sv, d_start, offs, m, first, d_hint,
                                   d_rec, total_hits, d_hits, d_compact, d_chunk_flags, d_total);
            else
                hipLaunchKernelGGL(locate_stream_kernel<false>, dim3(sgrid), dim3(kBlock), 0, stream, sv, d_start, offs, m, first, d_hint,
                                   d_rec, total_hits, d_hits, d_compact, d_chunk_flags, d_total);
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
    } else if (ix.layout == 0 && ix.pair_lines != nullptr && variant == 2) {
        uint64_t blocks = (total_hits + 31) / 32;
        if (blocks > 65536) blocks = 65536;
        if (grid_override > 0) blocks = static_cast<uint64_t>(grid_override);
        if (wide)
            hipLaunchKernelGGL(locate_pair_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, ix,
                               d_start, d_hit_offsets, heads, total_hits, d_hits, d_step_stats);
        else
            hipLaunchKernelGGL(locate_pair_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, ix,
                               d_start, d_hit_offsets, heads, total_hits, d_hits, d_step_stats);
    } else if (ix.layout == 0) {
        if (wide) GDX_LOCATE(LineTable, true);
        else GDX_LOCATE(LineTable, false);
    } else {
        if (wide) GDX_LOCATE(GenericTable, true);
        else GDX_LOCATE(GenericTable, false);
    }
#undef GDX_LOCATE
}

}  // namespace gdx
