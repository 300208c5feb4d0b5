// search.hip -- backward search, cursor extension and the operator-level rank / symbol_at kernels.
//
// search_pair_kernel4/8 (the default on DNA-sized alphabets): 4 or 8 lanes per query, 16 or 8 queries per
// wavefront advancing in lock-step -- the shape of the reference's 64-wide BatchComputedCursors
// (batch_computed_cursors.rs:36-73) with the swap-compaction replaced by the exec mask -- over the pair lines, the
// jump table and the top table.  search_kernel<Table, lanes>: the same search one LF step at
// a time on the rank lines (1 or 4 lanes per query) or on the generic planes of wide alphabets; also the readable
// statement of what the pair kernels compute.
//
// What is in this file, in its order (DESIGN.md section 4 has a row for each kernel with its bound and its bytes):
//   query windows        QueryWindow / PackedQueryWindow / FastWindow, query_words: a query's symbols, eight per load, from IO bytes
//                        (translated through the alphabet's table or two v_perm tables) or from 2-bit codes
//   general kernels      search_kernel<Table, lanes> (rank lines / generic planes, one LF step per fetch), search_pair_kernel4/8 and
//                        their _defer variants (pair lines: two symbols per fetch; top table, jump table), search_pair_packed_kernel4
//                        (the same on 2-bit reads, resumes from a state), search_pair_stats_kernel4/8 (the step counters the bench prints)
//   slim kernels         search_fast_kernel4 (top + jump table, count / locate), search_exact_kernel4<jump, translation, cursor, text>
//                        (exact intervals and cursor chunks; `text`: SA[row] -> text -> ISA in the jump table's place)
//   seed-table kernels   search_seed_lane_kernel (one bucket fetch per read, a lane per read: the headline's kernel),
//                        search_seed_kernel4<translation, exact, cursor> (four lanes per read: exact intervals through the ISA, the
//                        cursor API's first chunk, alphabets and buffers the lane kernel does not take), search_verify_kernel4
//                        (listed reads: up to four rows against the text), seed_text_kernel4 (long reads: the text in front of the seed)
//   operator level       extend_front_kernel (Cursor::extend_query_front), rank_many_kernel, symbol_at_kernel, lf_walk_kernel
//   tables and helpers   fill_lookup_kernel, fill_top_kernel, top_wide_kernel, pack_queries_kernel, zero_segments_kernel,
//                        fill_uniform_offsets[_list]_kernel, tile_sums_lists_kernel
//   launch_search        intervals of a batch with the general kernels (query options choose the variant)
//   launch_search_call   ONE search call of any kind (count / records / compact results / exact intervals / cursor chunk) on any
//                        index shape: picks the kernels above, chains their lists, folds the hit totals -- the only place that
//                        knows which kernel serves which shape
#include <atomic>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace gdx {

namespace {

constexpr int kBlock = 256;

// Reads the IO symbols of one query right-to-left through aligned 8-byte windows, so a lane
// issues one global load per 8 LF steps; the next window is requested one window ahead.
struct QueryWindow {
    const uint64_t *words;
    uint64_t cur;        // window holding byte `pos-1`
    uint64_t next;       // window below it (prefetched)
    uint64_t cur_word;   // index of `cur`
    uint64_t first_word; // lowest word that belongs to this query

    __device__ __forceinline__ void init(const uint8_t *qbuf, uint64_t begin, uint64_t pos)
    {
        words = reinterpret_cast<const uint64_t *>(qbuf);
        first_word = begin >> 3;
        const bool any = pos > begin;
        cur_word = any ? ((pos - 1) >> 3) : first_word;
        cur = any ? words[cur_word] : 0ull;
        next = (any && cur_word > first_word) ? words[cur_word - 1] : 0ull;
    }
    // byte at absolute offset `at` (at must walk downwards)
    __device__ __forceinline__ uint32_t get(uint64_t at)
    {
        const uint64_t w = at >> 3;
        if (w != cur_word) {
            cur_word = w;
            cur = next;
            next = (w > first_word) ? words[w - 1] : 0ull;
        }
        return static_cast<uint32_t>(cur >> ((at & 7u) * 8u)) & 0xffu;
    }
};

// The same over packed queries (include/gdx.h "packed queries": symbol j of the buffer in bits 2 (j & 7) of 16-bit unit
// j >> 3, offsets count symbols): one 2-byte load per 8 LF steps; get() returns the 2-bit code (dense symbol - 1).
struct PackedQueryWindow {
    const uint16_t *units;
    uint32_t cur, next;
    uint64_t cur_unit, first_unit;

    __device__ __forceinline__ void init(const uint8_t *qbuf, uint64_t begin, uint64_t pos)
    {
        units = reinterpret_cast<const uint16_t *>(qbuf);
        first_unit = begin >> 3;
        const bool any = pos > begin;
        cur_unit = any ? ((pos - 1) >> 3) : first_unit;
        cur = any ? units[cur_unit] : 0u;
        next = (any && cur_unit > first_unit) ? units[cur_unit - 1] : 0u;
    }
    __device__ __forceinline__ uint32_t get(uint64_t at)
    {
        const uint64_t u = at >> 3;
        if (u != cur_unit) {
            cur_unit = u;
            cur = next;
            next = (u > first_unit) ? units[u - 1] : 0u;
        }
        return (cur >> ((at & 7u) * 2u)) & 3u;
    }
};

// Where query q lies in the buffer: [qbeg[q], qend[q]) from the offsets arrays, or -- a UNIFORM batch, ulen != 0: every
// query has ulen symbols and query q starts at q * ulen (gdx_query_layout_t) -- computed, which saves the kernels the
// 8 bytes of offsets per query (the arrays may then be null).
__device__ __forceinline__ uint64_t query_begin(const uint64_t *__restrict__ qbeg, uint32_t ulen, uint64_t q)
{
    return ulen != 0u ? q * ulen : qbeg[q];
}
__device__ __forceinline__ uint64_t query_end(const uint64_t *__restrict__ qend, uint32_t ulen, uint64_t q)
{
    return ulen != 0u ? (q + 1u) * ulen : qend[q];
}
// the aligned word (packed, kXlate == 2: the 16-bit unit) that holds the first symbol of a query: fast_window's `wbase`
template <int kXlate>
__device__ __forceinline__ const uint64_t *query_words(const uint8_t *__restrict__ qbuf, uint64_t begin)
{
    if (kXlate == 2) return reinterpret_cast<const uint64_t *>(reinterpret_cast<const uint16_t *>(qbuf) + (begin >> 3));
    return reinterpret_cast<const uint64_t *>(qbuf) + (begin >> 3);
}

// The query as the pair kernels see it: dense codes, one nibble per symbol, eight symbols per 32-bit word, read
// right-to-left.  The kGroup lanes of a query translate the query COOPERATIVELY: a span of eight 8-byte words (64
// bytes: a whole 50-mer) is loaded and translated through the alphabet table in LDS by the group at once, each lane
// its own one (8 lanes) or two (4 lanes) words, and any lane fetches any translated word of the span from its
// owner with one ds_bpermute (the LDS crossbar, no LDS memory).  Translating every word in every lane -- which is
// what a plain per-lane window does -- made the kernel VALU-bound: PMC showed 880 VALU instructions per wavefront
// and 16 queries, 85 % VALU issue utilisation, and no gain from a quarter fewer DRAM requests.
__device__ __forceinline__ bool all_dna(uint32_t x);
__device__ __forceinline__ uint32_t to_2bit(uint32_t x);
constexpr uint32_t kNoCode = 0x10000u;  // compares unequal to every 16-bit entry code

// kPacked: the query buffer holds 2-bit codes (dense symbol - 1 of the four searchable symbols, symbol j of the
// buffer in bits 2 (j & 3) of byte j >> 2; include/gdx.h "packed queries") and offsets count SYMBOLS: a span word is
// then a 16-bit unit of the buffer that needs no translation at all.
template <int kGroup, bool kPacked>
struct SpanWindow {
    static constexpr int kWordsPerLane = 8 / kGroup;
    const uint64_t *base;  // base[0] holds the first byte of the query (packed: the 16-bit unit of its first symbol)
    uint32_t off0;         // byte (packed: symbol) offset of the query inside base[0] (inside its first unit)
    uint32_t top_w;        // span word k = query word top_w - k, k = 0 .. 7 (words below 0 read as 0)
    uint32_t w0, w1;       // this lane's translated span words: k = sub * kWordsPerLane (and + 1)
    uint32_t p;            // the same words as 2-bit codes (dense - 1), 16 bits each: w0 in the low half, w1 above
    uint32_t valid8;       // bit k: all eight symbols of span word k are dense codes 1..4 (the same in the group)

    static __device__ __forceinline__ uint32_t translate(uint64_t w, const uint8_t *s_dense)
    {
        uint32_t t = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) t |= static_cast<uint32_t>(s_dense[(w >> (8u * k)) & 0xffu]) << (4u * k);
        return t;
    }
    __device__ __forceinline__ void init(const uint8_t *qbuf, uint64_t begin)
    {
        if (kPacked) base = reinterpret_cast<const uint64_t *>(reinterpret_cast<const uint16_t *>(qbuf) + (begin >> 3));
        else base = reinterpret_cast<const uint64_t *>(qbuf) + (begin >> 3);
        off0 = static_cast<uint32_t>(begin & 7u);
        top_w = 0;
        w0 = w1 = p = valid8 = 0;
    }
    // eight 2-bit codes -> eight nibble codes 1..4
    static __device__ __forceinline__ uint32_t spread(uint32_t u)
    {
        uint32_t x = u & 0xffffu;
        x = (x | (x << 8)) & 0x00ff00ffu;
        x = (x | (x << 4)) & 0x0f0f0f0fu;
        x = (x | (x << 2)) & 0x33333333u;
        return x + 0x11111111u;
    }
    // positions the span so that its first word holds symbol rem - 1 (rem >= 1); group-uniform
    __device__ __forceinline__ void load(uint32_t rem, const uint8_t *s_dense)
    {
        const uint32_t sub = threadIdx.x & (kGroup - 1u);
        top_w = (off0 + rem - 1u) >> 3;
        const int32_t first = static_cast<int32_t>(top_w) - static_cast<int32_t>(sub) * kWordsPerLane;
        if (kPacked) {  // the units ARE the 2-bit span words; every symbol is one of the four searchable ones
            const uint16_t *units = reinterpret_cast<const uint16_t *>(base);
            const uint32_t u0 = first >= 0 ? units[first] : 0u;
            const uint32_t u1 = (kWordsPerLane == 2 && first >= 1) ? units[first - 1] : 0u;
            p = u0 | (u1 << 16);
            w0 = spread(u0);
            w1 = spread(u1);
            valid8 = 0xffu;
            return;
        }
        uint64_t r0 = 0, r1 = 0;
        if (kWordsPerLane == 2) {
            if (first >= 1) {  // both words with one 16-byte load (8-byte aligned)
                const u32x4 v = *reinterpret_cast<const u32x4 *>(base + (first - 1));
                r1 = static_cast<uint64_t>(v.x) | (static_cast<uint64_t>(v.y) << 32);
                r0 = static_cast<uint64_t>(v.z) | (static_cast<uint64_t>(v.w) << 32);
            } else if (first == 0) {
                r0 = base[0];
            }
        } else if (first >= 0) {
            r0 = base[first];
        }
        finish(r0, r1, s_dense);
    }
    // the second half of load(): this lane's raw span words -> translated words, 2-bit form, validity mask
    __device__ __forceinline__ void finish(uint64_t r0, uint64_t r1, const uint8_t *s_dense)
    {
        const uint32_t sub = threadIdx.x & (kGroup - 1u);
        w0 = translate(r0, s_dense);
        if (kWordsPerLane == 2) w1 = translate(r1, s_dense);
        // 2-bit form and the group's validity mask (an OR over the group's lanes by DPP)
        p = to_2bit(w0);
        uint32_t m = all_dna(w0) ? 1u : 0u;
        if (kWordsPerLane == 2) {
            p |= to_2bit(w1) << 16;
            m |= all_dna(w1) ? 2u : 0u;
        }
        m <<= sub * kWordsPerLane;
        m |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(m), 0xB1, 0xF, 0xF, true));
        m |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(m), 0x4E, 0xF, 0xF, true));
        if (kGroup == 8) m |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(m), 0x141, 0xF, 0xF, true));
        valid8 = m;
    }
    // 2-bit codes of span word k (16 bits)
    __device__ __forceinline__ uint32_t word2(uint32_t k) const
    {
        const uint32_t owner = (threadIdx.x & 63u & ~(kGroup - 1u)) + k / kWordsPerLane;
        const uint32_t v = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(owner << 2), static_cast<int>(p)));
        return (kWordsPerLane == 2 && (k & 1u)) ? v >> 16 : v & 0xffffu;
    }
    // 2-bit codes of the symbols r - 1 .. r - 8 (r >= 8, covered by the span), bits 15:14 = symbol r - 1: the form the
    // jump entries store; kNoCode when a word they touch holds a symbol outside 1..4 (conservative: the caller then
    // looks at the nibbles)
    __device__ __forceinline__ uint32_t level(uint32_t r) const
    {
        const uint32_t b = off0 + r - 1u, k = top_w - (b >> 3);
        const uint32_t s = (b & 7u) + 1u;
        const uint32_t k1 = k + 1u > 7u ? 7u : k + 1u;
        const uint32_t need = (1u << k) | (s < 8u ? (1u << k1) : 0u);
        const uint32_t x = (word2(k) << 16) | word2(k1);
        return (valid8 & need) == need ? ((x >> (2u * s)) & 0xffffu) : kNoCode;
    }
    // translated span word k (0 .. 7), k uniform in the group
    __device__ __forceinline__ uint32_t word(uint32_t k) const
    {
        const uint32_t owner = (threadIdx.x & 63u & ~(kGroup - 1u)) + k / kWordsPerLane;
        const uint32_t mine = (kWordsPerLane == 2 && (k & 1u)) ? w1 : w0;
        return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(owner << 2), static_cast<int>(mine)));
    }
    // true when the symbols r - 1 .. r - 8 (those of them that exist) lie inside the span
    __device__ __forceinline__ bool covers(uint32_t r) const
    {
        const uint32_t hi_w = (off0 + r - 1u) >> 3, lo_w = (off0 + (r >= 8u ? r - 8u : 0u)) >> 3;
        return hi_w <= top_w && top_w - lo_w <= 7u;
    }
    // codes of the 8 symbols that end with symbol r - 1 (nibble 7 = symbol r - 1); the span must cover them.
    // Nibbles of symbols before the start of the query are garbage and must not be used (callers check r).
    __device__ __forceinline__ uint32_t at(uint32_t r) const
    {
        const uint32_t b = off0 + r - 1u, k = top_w - (b >> 3);
        const uint32_t s = (b & 7u) + 1u;  // nibbles taken from span word k, the rest from word k + 1
        const uint32_t cur = word(k), next = word(k + 1u > 7u ? 7u : k + 1u);
        return s == 8u ? cur : __builtin_amdgcn_alignbit(cur, next, 4u * s);
    }
    // the same for the symbols the search consumes next (rem >= 1): moves the span down when they have left it
    __device__ __forceinline__ uint32_t code8(uint32_t rem, const uint8_t *s_dense)
    {
        if (!covers(rem)) load(rem, s_dense);
        return at(rem);
    }
};

__device__ __forceinline__ bool has_zero_nibble(uint32_t x) { return ((x - 0x11111111u) & ~x & 0x88888888u) != 0u; }

// Top table (IndexView::top): the interval after the first D = top_depth symbols of the search, i.e. the last D
// symbols of the query, when all of them are dense codes 1..4.  `a` = code8(rem), `b` = code8(rem - 8) (only its
// top D - 8 nibbles are used).  The entry of a D-mer that does not occur holds the frozen interval (start == end as
// they stood at the step that emptied it, like the reference's lookup tables, lookup_table.rs:225-258), so such a
// query is finished by the lookup.  Returns false when a symbol is outside 1..4: the caller then runs the ordinary
// steps from the start, which validate symbols as lazily as the reference does.
__device__ __forceinline__ bool all_dna(uint32_t x);
__device__ __forceinline__ uint32_t to_2bit(uint32_t x);

__device__ __forceinline__ bool top_lookup(const IndexView &ix, uint32_t a, uint32_t b, uint32_t &lo, uint32_t &hi)
{
    const uint32_t d = ix.top_depth;  // 1 .. 16: the top min(d, 8) nibbles of a and the top d - 8 of b
    const uint32_t ma = d >= 8u ? 0xffffffffu : 0xffffffffu << (4u * (8u - d));
    const uint32_t mb = d > 8u ? 0xffffffffu << (4u * (16u - d)) : 0u;
    const uint32_t va = (a & ma) | (0x11111111u & ~ma), vb = (b & mb) | (0x11111111u & ~mb);  // unused := valid
    if (!all_dna(va) || !all_dna(vb)) return false;
    const uint32_t idx = ((to_2bit(va) << 16) | to_2bit(vb)) >> (32u - 2u * d);
    const uint2 e = ix.top[idx];
    lo = e.x;
    hi = e.y;
    return true;
}

// appends the cursors of this wavefront's `alive` lanes to ca.active_out: one atomic per wavefront
__device__ __forceinline__ void compact_alive(bool alive, uint32_t q, const CursorArgs &ca)
{
    const unsigned long long mask = __ballot(alive);
    if (mask == 0ull) return;
    const uint32_t lane = __lane_id();
    const int leader = __ffsll(static_cast<long long>(mask)) - 1;
    uint32_t first = 0;
    if (static_cast<int>(lane) == leader) first = atomicAdd(ca.n_active_out, static_cast<uint32_t>(__popcll(mask)));
    first = __shfl(first, leader);
    if (alive) ca.active_out[first + __popcll(mask & ((1ull << lane) - 1ull))] = q;
}

// kGroup lanes cooperate on one query (1: LineTable / GenericTable, 4: QuadLineTable); control flow is
// uniform inside a group, lane 0 of the group writes the results.  kResume: cursor extension (see search_pair_body,
// kMode 2): start from the interval in out_start / out_end, no lookup table, active lists.
// kPacked: 2-bit queries (never with kResume); ulen: uniform batch (query_begin).
template <class Table, int kGroup, bool kResume, bool kPacked = false>
__global__ __launch_bounds__(kBlock) void search_kernel(IndexView ix, const uint8_t *__restrict__ qbuf,
                                                        const uint64_t *__restrict__ qbeg,
                                                        const uint64_t *__restrict__ qend, uint64_t nq,
                                                        uint32_t *__restrict__ out_start,
                                                        uint32_t *__restrict__ out_end,
                                                        uint32_t *__restrict__ out_count,
                                                        uint8_t *__restrict__ out_status,
                                                        unsigned long long *__restrict__ step_stats,
                                                        uint4 *__restrict__ out_rec, CursorArgs ca, uint32_t ulen)
{
    __shared__ uint8_t s_dense[256];
    __shared__ uint32_t s_count[257];
    for (int i = threadIdx.x; i < 256; i += kBlock) s_dense[i] = ix.io_to_dense[i];
    for (int i = threadIdx.x; i <= ix.sigma; i += kBlock) s_count[i] = ix.count[i];
    __syncthreads();

    const uint32_t k = static_cast<uint32_t>(ix.n_searchable);
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * (kBlock / kGroup);
    const bool writer = (threadIdx.x % kGroup) == 0;
    const uint32_t *active = ca.active_in;  // cursor lists, or the queries another kernel left over
    if (ca.n_active_in != nullptr) nq = *ca.n_active_in;
    uint32_t lf_steps = 0;  // only reported through step_stats (bench accounting, null in normal calls)
    for (uint64_t at = static_cast<uint64_t>(blockIdx.x) * (kBlock / kGroup) + threadIdx.x / kGroup; at < nq;
         at += stride) {
        const uint64_t q = active ? active[at] : at;
        uint64_t begin = query_begin(qbeg, ulen, q), end = query_end(qend, ulen, q);
        bool more_left = true;  // chunk view (CursorArgs::chunk_symbols): the query has symbols left of this chunk
        if (kResume && ca.chunk_symbols != 0u) {
            const uint64_t first = begin, skip = static_cast<uint64_t>(ca.chunk_index) * ca.chunk_symbols;
            end = end - first > skip ? end - skip : first;
            begin = end - first > ca.chunk_symbols ? end - ca.chunk_symbols : first;
            more_left = begin > first;
        }
        const uint64_t len = end - begin;
        // lib.rs:277-281 split_query_for_lookup
        const uint32_t t = kResume ? 0u
                                   : (len < static_cast<uint64_t>(ix.depth) ? static_cast<uint32_t>(len)
                                                                            : static_cast<uint32_t>(ix.depth));
        uint32_t lo = 0, hi = ix.n, status = GDX_Q_OK;
        bool stopped = false;
        if (kResume) {
            lo = out_start[q];
            hi = out_end[q];
            if (out_status != nullptr) {
                status = out_status[q];
                stopped = status != GDX_Q_OK;
            }
        }
        if (t > 0) {
            // lookup_table.rs:99-113: idx = sum (dense-1) * k^j, j = 0 is the leftmost suffix symbol
            uint64_t idx = 0, factor = 1;
            bool unsearchable = false;
            for (uint32_t j = 0; j < t; j++) {
                const uint64_t sj = end - t + j;
                const uint32_t d = kPacked ? ((static_cast<uint32_t>(qbuf[sj >> 2]) >> (2u * (sj & 3u))) & 3u) + 1u : s_dense[qbuf[sj]];
                if (d == 0) status = GDX_Q_INVALID_SYMBOL;
                unsearchable |= (d - 1u >= k);
                idx += static_cast<uint64_t>(d - 1u) * factor;
                factor *= k;
            }
            if (status == GDX_Q_OK && unsearchable) status = GDX_Q_UNSEARCHABLE_IN_LOOKUP;
            if (status == GDX_Q_OK) {
                const uint2 v = ix.lookup[lookup_offset(k, t) + idx];
                lo = v.x;
                hi = v.y;
            } else {
                lo = hi = 0;
            }
        }
        uint64_t pos = end - t;  // symbols [begin, pos) are still to be consumed, right to left
        typename std::conditional<kPacked, PackedQueryWindow, QueryWindow>::type win;
        win.init(qbuf, begin, pos);
        // lib.rs:226-232 / batch_computed_cursors.rs:62-70: stop at the empty interval
        while (pos > begin && lo != hi && !stopped) {
            const uint32_t c = kPacked ? win.get(pos - 1) + 1u : s_dense[win.get(pos - 1)];
            if (c == 0) {  // alphabet.rs:195-198
                status = GDX_Q_INVALID_SYMBOL;
                if (!kResume) lo = hi = 0;
                break;
            }
            uint32_t rlo, rhi;
            Table::rank2(ix, c, lo, hi, rlo, rhi);
            const uint32_t cc = s_count[c];  // lib.rs:273-275
            lo = cc + rlo;
            hi = cc + rhi;
            pos--;
            lf_steps++;
        }
        if (writer) {
            if (out_rec) out_rec[q] = make_uint4(lo, hi, 0xffffffffu, (status & 0xffu) << 24);
            if (out_start) out_start[q] = lo;
            if (out_end) out_end[q] = hi;
            if (out_count) out_count[q] = hi - lo;
            if (out_status) out_status[q] = static_cast<uint8_t>(status);
        }
        if (kResume && ca.active_out != nullptr)
            compact_alive(writer && lo != hi && status == GDX_Q_OK && more_left, static_cast<uint32_t>(q), ca);
    }
    if (step_stats && writer) atomicAdd(step_stats, static_cast<unsigned long long>(lf_steps));
}

// ---- length-ordered schedule inside a block -----------------------------------------------------------
// The lock-step kernel idles the lanes of a finished query until the longest query of its wavefront ends.  Every
// block searches a contiguous range of the batch (at most kMaxRange queries); when the lengths in that range are
// spread out it first orders the range by length (counting sort in LDS, 4 symbols per bucket, longest first) and
// its wavefronts take runs of that order, so the queries of a wavefront have nearly the same length.  Results
// are written at the original query index.  All reads and writes of a block stay inside its range, so the
// order costs no DRAM locality (an earlier batch-wide permutation did: its gathers were slower than the idle
// lanes it saved), and there is no pre-pass and no host round trip.  Ranges of (nearly) equal lengths skip it.
constexpr uint32_t kLenBuckets = 64;
constexpr uint32_t kMaxRange = 3072;  // queries per block and range: u16 permutation, 6 KB of LDS
constexpr uint32_t kMaxDefer = 512;   // stragglers a block can park per range (8 KB of LDS); more are finished in place
constexpr uint32_t kCursorRange = 1024;  // cursor extension: queries per range (their live list is 4 KB of LDS)

__device__ __forceinline__ uint32_t length_bucket(uint64_t len)
{
    const uint64_t b = len >> 2;
    return kLenBuckets - 1u - static_cast<uint32_t>(b < kLenBuckets - 1u ? b : kLenBuckets - 1u);
}

// Orders queries [base, base + cnt) of the batch by length bucket into s_perm (indices relative to base); returns
// false (s_perm untouched) when the lengths are uniform enough: max - min <= min / 4.  Block-wide, all threads.
__device__ __forceinline__ bool order_range_by_length(const uint64_t *__restrict__ qbeg,
                                                      const uint64_t *__restrict__ qend,
                                                      const uint32_t *__restrict__ active, uint64_t base, uint32_t cnt,
                                                      uint16_t *s_perm, uint32_t *s_cnt, uint32_t *s_minmax)
{
    if (threadIdx.x < kLenBuckets) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        s_minmax[0] = 0xffffffffu;
        s_minmax[1] = 0;
    }
    __syncthreads();
    uint32_t mn = 0xffffffffu, mx = 0;
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        const uint64_t qi = active ? active[base + i] : base + i;
        const uint64_t len = qend[qi] - qbeg[qi];
        const uint32_t l = len > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(len);
        mn = l < mn ? l : mn;
        mx = l > mx ? l : mx;
        atomicAdd(&s_cnt[length_bucket(len)], 1u);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t omn = __shfl_xor(mn, off), omx = __shfl_xor(mx, off);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicMin(&s_minmax[0], mn);
        atomicMax(&s_minmax[1], mx);
    }
    __syncthreads();
    mn = s_minmax[0];
    mx = s_minmax[1];
    if (mx - mn <= mn / 4u) return false;
    if (threadIdx.x < kLenBuckets) {  // exclusive scan of the histogram by the first wavefront
        const uint32_t v = s_cnt[threadIdx.x];
        uint32_t x = v;
        for (int off = 1; off < static_cast<int>(kLenBuckets); off <<= 1) {
            const uint32_t y = __shfl_up(x, off);
            if (static_cast<int>(threadIdx.x) >= off) x += y;
        }
        s_cnt[threadIdx.x] = x - v;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        const uint64_t qi = active ? active[base + i] : base + i;
        const uint32_t r = atomicAdd(&s_cnt[length_bucket(qend[qi] - qbeg[qi])], 1u);
        s_perm[r] = static_cast<uint16_t>(i);
    }
    __syncthreads();
    return true;
}

// all eight nibbles are dense codes 1..4 (searchable DNA symbols)
__device__ __forceinline__ bool all_dna(uint32_t x)
{
    const uint32_t y = x - 0x11111111u;  // per nibble c - 1 when there is no zero nibble (no borrow)
    return ((y & ~x & 0x88888888u) | (y & 0xccccccccu)) == 0u;
}

// eight nibble codes 1..4 -> eight 2-bit codes (c - 1), nibble 7 (the symbol consumed first) in bits 15:14: the form
// the jump entries store (layout.hpp).  Only meaningful when all_dna(x).
__device__ __forceinline__ uint32_t to_2bit(uint32_t x)
{
    uint32_t y = x - 0x11111111u;
    y = (y & 0x03030303u) | ((y & 0x30303030u) >> 2);
    y = (y & 0x000f000fu) | ((y & 0x0f000f00u) >> 4);
    return (y & 0xffu) | ((y >> 8) & 0xff00u);
}

// Live cursors of a block's range (cursor extension, mode 2) are appended to the global list IN THE ORDER OF THEIR
// POSITIONS in the range: s_live[p] = the cursor at position p, or kDeadCursor.  A list that starts sorted then stays
// sorted chunk by chunk, and the next call's gathers of start / end / status / string offsets by cursor number touch
// neighbouring lines; appending in the order of LDS atomics scattered them over sixteen lines per load instruction,
// which made a call over 34 M live cursors with empty strings cost 3.3 ms.
constexpr uint32_t kDeadCursor = 0xffffffffu;
__device__ __forceinline__ void flush_live_ordered(uint32_t *s_live, uint32_t cnt, uint32_t *s_part, uint32_t *s_base,
                                                   uint32_t *active_out, uint32_t *n_active_out)
{
    constexpr uint32_t kPer = kCursorRange / kBlock;
    uint32_t v[kPer], mine = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++) {
        const uint32_t p = threadIdx.x * kPer + j;
        v[j] = p < cnt ? s_live[p] : kDeadCursor;
        s_live[p] = kDeadCursor;  // for the next range
        mine += v[j] != kDeadCursor ? 1u : 0u;
    }
    s_part[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {  // inclusive scan
        const uint32_t o = static_cast<int>(threadIdx.x) >= off ? s_part[threadIdx.x - off] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += o;
        __syncthreads();
    }
    const uint32_t total = s_part[kBlock - 1];
    if (threadIdx.x == 0 && total != 0u) *s_base = atomicAdd(n_active_out, total);
    __syncthreads();
    uint32_t pos = *s_base + s_part[threadIdx.x] - mine;
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++)
        if (v[j] != kDeadCursor) active_out[pos++] = v[j];
    __syncthreads();
}

// Backward search on pair lines, the jump table and the top table: kGroup = 4 or 8 lanes per
// query, every loop iteration is one round of loads for the whole wavefront.
// kMode 0 (intervals): out_start / out_end are the reference's half-open SA interval, bit for bit, also for empty
//   results (cursors_for_many_queries).
// kMode 1 (count / locate): only the number of occurrences and the locate hint matter.  A query whose interval is one
//   row wide and that has fewer than eight symbols left after a jump is finished from the jump entry already in
//   registers -- its next level holds the symbols preceding that row's suffix, so comparing them with the query's
//   remaining symbols tells whether the occurrence survives, without fetching the row's pair line (the "lazy tail").
//   The interval then reported is [row, row + 1) or [row, row) of the row BEFORE those steps together with a hint
//   {row', rem'}: "SA[hit] = SA[row'] - rem'".  end - start is the reference's count; start / end themselves are only
//   meaningful to launch_locate through the hint.  Results go to out_rec (one 16-byte record per query:
//   {start, end, hint row, hint symbols | status << 24}) when it is given.
// kMode 2 (cursor extension, Cursor::extend_query_front cursor.rs:34-51 applied to every symbol of a string, right to
//   left): the search resumes from the interval in out_start / out_end (in / out) instead of [0, n); the top table
//   is used for a cursor that is still the empty cursor [0, n).  The queries are those listed in ca.active_in
//   (*ca.n_active_in of them; null = all nq), and the cursors that are still non-empty afterwards are appended to
//   ca.active_out (device-side compaction: one atomic per wavefront), so that a caller feeding long queries in
//   chunks touches only live cursors.  An invalid symbol stops its cursor where it stands (status set).
template <int kPolicy, int kGroup, bool kStats, int kJump, int kMode, bool kPacked = false, bool kDefer = false>
__device__ __forceinline__ void search_pair_body(IndexView ix, const uint8_t *__restrict__ qbuf,
                                                             const uint64_t *__restrict__ qbeg,
                                                             const uint64_t *__restrict__ qend, uint64_t nq,
                                                             uint32_t *__restrict__ out_start,
                                                             uint32_t *__restrict__ out_end,
                                                             uint32_t *__restrict__ out_count,
                                                             uint8_t *__restrict__ out_status,
                                                             unsigned long long *__restrict__ step_stats,
                                                             uint32_t range, int schedule, uint2 *__restrict__ out_hint,
                                                             uint4 *__restrict__ out_rec, CursorArgs ca, uint32_t defer_after)
{
    __shared__ uint8_t s_dense[256];
    __shared__ uint32_t s_count[257];
    __shared__ uint16_t s_perm[kMaxRange];
    __shared__ uint2 s_hint[kBlock / kGroup];  // locate hint of the query each group is searching
    __shared__ uint32_t s_cnt[kLenBuckets];
    __shared__ uint32_t s_minmax[2];
    // kMode 2: the cursors of the block's current range that stay alive, appended to ca.active_out with ONE global
    // atomic per range (an atomic per wavefront serialised the whole chip on one address: 27 ms per call at 34 M
    // live cursors)
    __shared__ uint32_t s_alive[kMode == 2 ? kCursorRange : 1];  // flush_live_ordered
    __shared__ uint32_t s_alive_part[kMode == 2 ? kBlock : 1];
    __shared__ uint32_t s_alive_base;
    // Stragglers: a query that is not done after its allowance of load rounds (a read from a repeat, whose interval
    // stays wide and is narrowed two symbols per round) is parked here {query, lo, hi, rem} and finished in a second
    // pass over the block's range in which EVERY group works on such a query, instead of holding the fifteen finished
    // queries of its wavefront for dozens of rounds (genome-like text: active lanes 0.27 without this).
    // (kDefer is a template parameter: the bookkeeping costs the plain kernel 1.6 ms per 100 M reads in registers)
    __shared__ uint4 s_defer[kDefer ? kMaxDefer : 1];
    __shared__ uint32_t s_ndefer;
    for (int i = threadIdx.x; i < 256; i += kBlock) s_dense[i] = ix.io_to_dense[i];
    for (int i = threadIdx.x; i <= ix.sigma; i += kBlock) s_count[i] = ix.count[i];
    if (kMode == 2)
        for (uint32_t i = threadIdx.x; i < kCursorRange; i += kBlock) s_alive[i] = kDeadCursor;
    if (threadIdx.x == 0) s_ndefer = 0;
    __syncthreads();

    const uint32_t k = static_cast<uint32_t>(ix.n_searchable);
    const bool writer = (threadIdx.x % kGroup) == 0;
    const bool hinting = out_hint != nullptr || out_rec != nullptr;
    uint32_t lf_steps = 0;
    unsigned long long group_iters = 0, wave_slots = 0;  // step_stats[1], [2]
    const uint32_t *active = ca.active_in;  // cursor lists (mode 2) or the leftover list of the fast path
    if (ca.n_active_in != nullptr) nq = *ca.n_active_in;
    const uint64_t n_ranges = (nq + range - 1) / range;
    for (uint64_t rg = blockIdx.x; rg < n_ranges; rg += gridDim.x) {
    const uint64_t base = rg * range;
    const uint32_t cnt = nq - base < range ? static_cast<uint32_t>(nq - base) : range;
    if (rg != blockIdx.x) __syncthreads();  // the previous range's order is no longer read
    // (chunk view: the strings of a call are chunks of one size, whatever the lengths of the queries)
    const bool ordered = schedule != 0 && !(kMode == 2 && ca.chunk_symbols != 0u) &&
                         order_range_by_length(qbeg, qend, active, base, cnt, s_perm, s_cnt, s_minmax);
    for (int phase = 0; phase < (kDefer ? 2 : 1); phase++) {  // 0: the range's queries; 1: the stragglers parked in phase 0
    uint32_t n_items = cnt;
    if (kDefer && phase == 1) {
        if (defer_after == 0u) break;
        __syncthreads();
        n_items = s_ndefer < kMaxDefer ? s_ndefer : kMaxDefer;
        if (n_items == 0u) break;
    }
    for (uint32_t slot = threadIdx.x / kGroup; slot < n_items; slot += kBlock / kGroup) {
        uint4 parked = make_uint4(0u, 0u, 0u, 0u);
        if (kDefer && phase == 1) parked = s_defer[slot];
        const uint64_t at = base + (ordered ? s_perm[phase == 0 ? slot : 0u] : slot);
        const uint64_t q = (kDefer && phase == 1) ? static_cast<uint64_t>(parked.x) : (active ? active[at] : at);
        bool resumed = kDefer && phase == 1;
        if (kMode != 2 && ca.resume_state != nullptr && !resumed) {  // a leftover of the fast path, taken up where it stood
            const uint4 st = ca.resume_state[q];
            if (st.w == 1u) {
                parked = make_uint4(static_cast<uint32_t>(q), st.x, st.y, st.z);
                resumed = true;
            }
        }
        uint64_t begin = qbeg[q], end = qend[q];
        bool more_left = true;  // kMode 2, chunk view: the query has symbols left of this chunk
        if (kMode == 2 && ca.chunk_symbols != 0u) {
            const uint64_t first = begin, skip = static_cast<uint64_t>(ca.chunk_index) * ca.chunk_symbols;
            end = end - first > skip ? end - skip : first;
            begin = end - first > ca.chunk_symbols ? end - ca.chunk_symbols : first;
            more_left = begin > first;
        }
        const uint64_t len = end - begin;
        // (packed queries skip the configured lookup table: the steps it replaces give the same interval)
        const uint32_t t = (kMode == 2 || kPacked) ? 0u
                                                   : (len < static_cast<uint64_t>(ix.depth) ? static_cast<uint32_t>(len)
                                                                                            : static_cast<uint32_t>(ix.depth));
        uint32_t lo = 0, hi = ix.n, status = GDX_Q_OK;
        bool stopped = false;  // kMode 2: a cursor that met an invalid symbol earlier stays where it stopped
        if (kMode == 2) {
            lo = out_start[q];
            hi = out_end[q];
            if (out_status != nullptr) {
                status = out_status[q];
                stopped = status != GDX_Q_OK;
            }
        }
        const bool fresh = lo == 0u && hi == ix.n;  // cursor_empty (lib.rs:202-210)
        uint32_t rem = 0;  // symbols still to consume, right to left
        SpanWindow<kGroup, kPacked> win;
        win.init(qbuf, begin);
        if (resumed) {  // a parked query resumes where it stood (it had no status and no hint)
            lo = parked.y;
            hi = parked.z;
            rem = parked.w;
            status = GDX_Q_OK;
            stopped = false;
            if (rem > 0u) win.load(rem, s_dense);
        }
        // The top table is tried first when it is deeper than the configured lookup table: a hit means the last
        // top_depth symbols are all in 1..4, hence (with at least four searchable symbols) valid and searchable,
        // which is everything the reference checks for its t-symbol suffix (lookup_table.rs:99-113); the interval
        // is the one the configured table plus the LF steps in between would give.
        bool topped = resumed;
        if (!topped && ix.top != nullptr && ix.top_depth > t && k >= 4u && len >= ix.top_depth && len <= 0xffffffffull &&
            (kMode != 2 || (fresh && !stopped))) {
            const uint32_t lo0 = lo, hi0 = hi;
            rem = static_cast<uint32_t>(len);
            win.load(rem, s_dense);
            // the index straight from the span's 2-bit codes when 16 symbols are there and clean, else from nibbles
            uint32_t l1 = kNoCode, l2 = kNoCode;
            if (rem >= 16u) {
                l1 = win.level(rem);
                l2 = win.level(rem - 8u);
            }
            if (l1 != kNoCode && l2 != kNoCode) {
                const uint2 *te = ix.top + (((l1 << 16) | l2) >> (32u - 2u * ix.top_depth));
                uint2 e;
                if (kPolicy == 1) {  // no cache allocation for a line of a 34 GB table that is never touched again
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(te));
                    e = make_uint2(v.x, v.y);
                } else {
                    e = *te;
                }
                lo = e.x;
                hi = e.y;
                topped = true;
            } else {
                const uint32_t a = win.code8(rem, s_dense);
                const uint32_t b = ix.top_depth > 8u ? win.code8(rem - 8u, s_dense) : 0u;
                topped = top_lookup(ix, a, b, lo, hi);
            }
            if (topped) {
                rem -= ix.top_depth;
                if (kStats) lf_steps += ix.top_depth;
            } else {
                lo = lo0;
                hi = hi0;
            }
        }
        if (!topped) {
            if (t > 0) {
                // lookup_table.rs:99-113: idx = sum (dense-1) * k^j, j = 0 is the leftmost suffix symbol
                uint64_t idx = 0, factor = 1;
                bool unsearchable = false;
                for (uint32_t j = 0; j < t; j++) {
                    const uint32_t d = s_dense[qbuf[end - t + j]];
                    if (d == 0) status = GDX_Q_INVALID_SYMBOL;
                    unsearchable |= (d - 1u >= k);
                    idx += static_cast<uint64_t>(d - 1u) * factor;
                    factor *= k;
                }
                if (status == GDX_Q_OK && unsearchable) status = GDX_Q_UNSEARCHABLE_IN_LOOKUP;
                if (status == GDX_Q_OK) {
                    const uint2 v = ix.lookup[lookup_offset(k, t) + idx];
                    lo = v.x;
                    hi = v.y;
                } else {
                    lo = hi = 0;
                }
            }
            rem = stopped ? 0u : static_cast<uint32_t>(len - t);
            if (rem > 0u) win.load(rem, s_dense);
        }
        // hints carry symbol counts of 21 bits (locate.hip packs them beside a slot number); longer queries
        // (2 M symbols and more) simply do not jump
        bool jump_ok = ix.jump != nullptr && ix.jump_bytes == static_cast<uint32_t>(kJump) && len < (1ull << 21);
        uint32_t iters = 0;  // fetch rounds of this query (divergence accounting, kStats only)
        // allowance of load rounds before the query is parked: what a query that jumps needs, plus defer_after
        uint32_t rounds = 0, allowance = 0xffffffffu;
        bool deferred = false;
        // Every iteration is one round of loads for the whole wavefront, whatever its queries are doing: a group
        // either reads its jump entry or the pair line(s) of its interval borders, all loads are issued, then
        // waited for once, then each group interprets what it got (a wavefront whose groups took different
        // branches, each with its own load and wait, would pay one DRAM latency per branch).
        constexpr int kChunks = 8 / kGroup;
        // levels of an entry the group can use, and codes it sees (one more than levels = a lookahead, except in
        // full 32-byte entries whose fifth code belongs to the fifth level)
        constexpr int kLevels = kJump == 8 ? 1 : ((kJump == 16 || kChunks == 1) ? 2 : 4);
        constexpr int kCodes = kJump == 8 ? 1 : ((kJump == 16 || kChunks == 1) ? 3 : 5);
        constexpr bool kHalf2 = kLevels == 4;     // the group also reads the entry's second half {t3, t4, SA, c4 | c5 << 16}
        constexpr bool kEntrySA = kJump == 32;    // entries carry SA[row]: locate resolves ANY row with one fetch, so a
                                                  // hint only has to say how many symbols a lazy tail left unmatched
        const uint32_t sub = threadIdx.x & (kGroup - 1u);
        if (kDefer && phase == 0 && defer_after != 0u) allowance = defer_after + rem / (kJumpSymbols * kLevels);
        while (rem > 0 && lo != hi) {
            if (kDefer && rounds >= allowance) {  // park it (group-uniform); a full list means it is finished in place
                uint32_t slot_d = 0;
                if (writer) slot_d = atomicAdd(&s_ndefer, 1u);
                slot_d = static_cast<uint32_t>(__shfl(static_cast<int>(slot_d), static_cast<int>(threadIdx.x & 63u & ~(kGroup - 1u))));
                if (slot_d < kMaxDefer) {
                    if (writer) s_defer[slot_d] = make_uint4(static_cast<uint32_t>(q), lo, hi, rem);
                    deferred = true;
                    break;
                }
                allowance = 0xffffffffu;
            }
            if (kDefer) rounds++;
            if (kStats) iters++;
            if (!win.covers(rem)) win.load(rem, s_dense);
            // Intervals of at most one row per lane of the group jump: lane j reads the entry of row lo + j.  The
            // rows whose stored symbols equal the query's next 8, 16, ... are exactly those that survive these LF
            // steps, and LF keeps their order, so they map onto [min target, max target + 1).
            const bool narrow = jump_ok && hi - lo <= static_cast<uint32_t>(kGroup) && rem >= kJumpSymbols;
            // level codes of the query, packed like the entries store them: qa = level 1 | level 2 << 16, qb = level 3,
            // qc = level 4 (level j + 1 <-> symbols rem - 8j - 1 .. rem - 8j - 8); qok bit j = level
            // j + 1 lies inside the query and is all symbols 1..4.  They come from the span's 2-bit words; only when
            // those cannot vouch for level 1 (a symbol outside 1..4 somewhere in its words) are the nibbles looked at.
            uint32_t qa = narrow ? win.level(rem) : kNoCode;
            bool jumping = qa != kNoCode;
            uint32_t c1 = 1u, c2 = 0u;
            if (!jumping) {
                const uint32_t code = win.at(rem);
                c1 = code >> 28;
                if (c1 == 0) {  // alphabet.rs:195-198
                    status = GDX_Q_INVALID_SYMBOL;
                    if (kMode != 2) lo = hi = 0;
                    break;
                }
                if (c1 > 4u) {  // a valid symbol outside 1..4 (N): rank lines, rare
                    uint32_t rlo, rhi;
                    QuadLineTable::rank2(ix, c1, lo, hi, rlo, rhi);
                    const uint32_t cc = s_count[c1];
                    lo = cc + rlo;
                    hi = cc + rhi;
                    rem--;
                    if (kStats) lf_steps++;
                    continue;
                }
                c2 = rem >= 2u ? ((code >> 24) & 15u) : 0u;
                jumping = narrow && all_dna(code);
                if (jumping) qa = to_2bit(code);
            }
            uint32_t qb = 0, qc = 0, qok = 0;
            uint32_t tail16 = kNoCode;  // kMode 1: codes of the last rem % 8 symbols, in the top bits
            if (jumping) {
                // every symbol the levels (and the lazy tail) look at must be inside the span: after a span load
                // at rem they are (40 symbols and their alignment are at most seven words)
                const uint32_t n_lv = rem >> 3 < static_cast<uint32_t>(kLevels) ? rem >> 3 : static_cast<uint32_t>(kLevels);
                const bool tail = kMode == 1 && (rem & 7u) != 0u && (rem >> 3) < static_cast<uint32_t>(kCodes);
                const uint32_t r_low = tail ? 8u : rem - (n_lv - 1u) * kJumpSymbols;
                if (!win.covers(r_low)) win.load(rem, s_dense);
                qok = 1u;
#pragma unroll
                for (int j = 1; j < kLevels; j++) {
                    if (rem >= (j + 1u) * kJumpSymbols) {
                        const uint32_t v = win.level(rem - j * kJumpSymbols);
                        if (v != kNoCode) {
                            if (j == 1) qa |= v << 16;
                            if (j == 2) qb = v;
                            if (j == 3) qc = v;
                            if (j == 4) qc |= v << 16;
                            qok |= 1u << j;
                        }
                    }
                }
                if (tail) {
                    // the query's first rem % 8 symbols in the top bits; level(8) = symbols 7 .. 0
                    const uint32_t v = win.level(8u);
                    if (v != kNoCode) tail16 = (v << (2u * (8u - (rem & 7u)))) & 0xffffu;
                }
            }
            const uint32_t line_lo = lo >> kPairLineShift, line_hi = hi >> kPairLineShift;
            const bool second = !jumping && line_hi != line_lo;
            const u32x4 *pa = ix.pair_lines + (static_cast<uint64_t>(line_lo) << 3) + sub;
            const u32x4 *pb = ix.pair_lines + (static_cast<uint64_t>(line_hi) << 3) + sub;
            const u32x4 *pa1 = pa + kGroup;  // second chunk of the line (4 lanes per query)
            const uint32_t row = lo + sub < hi ? lo + sub : hi - 1u;  // spare lanes repeat the last row
            if (jumping) {
                const u32x4 *tab = static_cast<const u32x4 *>(ix.jump);
                if (kJump == 8) pa = tab + (row >> 1);  // the aligned pair of 8-byte entries
                else pa = tab + static_cast<uint64_t>(row) * (kJump / 16);
                pa1 = pa + 1;  // second half of a 32-byte entry
            }
            const bool need_a1 = !jumping || kHalf2;
            const unsigned long long m_a1 = __ballot(need_a1), m_second = __ballot(second);
            u32x4 a[kChunks], b[kChunks];
            // all loads of the round, their exec masks and the one wait: a single asm statement (layout.hpp)
            if (kChunks == 2) load_round4<kPolicy>(pa, pa1, pb, pb + kGroup, m_a1, m_second, a[0], a[kChunks - 1], b[0], b[kChunks - 1]);
            else load_round2<kPolicy>(pa, pb, m_second, a[0], b[0]);
            if (jumping) {
                // entry layout (layout.hpp): a[0] = {t1, t2, c1 | c2 << 16, c3 | valid << 16},
                // a[1] = {t3, t4, SA, c4 | c5 << 16}; 8-byte entries: {t1, c1 | valid << 16}
                uint32_t valid, da, db = 0, dc = 0;  // d*: stored codes XOR query codes, packed as above
                const u32x4 e0 = a[0], e1 = a[kChunks - 1];
                if (kJump == 8) {
                    const uint32_t w = (row & 1u) ? e0.w : e0.y;
                    da = (w ^ qa) & 0xffffu;
                    valid = w >> 16;
                } else {
                    da = e0.z ^ qa;
                    db = (e0.w ^ qb) & 0xffffu;
                    valid = e0.w >> 16;
                    if (kHalf2) dc = e1.w ^ qc;
                }
                // how many levels this lane's row matches: a level needs a valid stored level (the valid bits are
                // cumulative), a usable query level and equal codes, and all the levels before it
                uint32_t good = (da & 0xffffu) == 0u ? 1u : 0u;
                if (kLevels >= 2) good |= (da >> 16) == 0u ? 2u : 0u;
                if (kHalf2) {
                    good |= db == 0u ? 4u : 0u;
                    good |= (dc & 0xffffu) == 0u ? 8u : 0u;
                }
                good &= valid & qok;
                const uint32_t lvl = static_cast<uint32_t>(__builtin_ctz(~good | (1u << kLevels)));  // trailing ones
                const uint32_t best = group_max<kGroup>(lvl);
                if (best != 0u) {
                    const bool mine = lvl == best;
                    // locate hint, candidate 1: the one-row interval this jump starts from (see below)
                    uint32_t hr = 0xffffffffu, ho = 0;
                    const bool want_hint = !kEntrySA && hinting && !(status >> 31);  // one hint per query is enough
                    if (want_hint && hi - lo == 1u && is_sampled(ix, lo)) {
                        hr = lo;
                        ho = rem;
                    }
                    uint32_t target = kJump == 8 ? ((row & 1u) ? e0.z : e0.x) : e0.x;
                    if (kLevels >= 2) target = best == 2u ? e0.y : target;
                    if (kHalf2) {
                        target = best == 3u ? e1.x : target;
                        target = best == 4u ? e1.y : target;
                    }
                    lo = group_min<kGroup>(mine ? target : 0xffffffffu);
                    hi = group_max<kGroup>(mine ? target : 0u) + 1u;
                    const uint32_t done = best * kJumpSymbols;
                    rem -= done;
                    if (kStats) lf_steps += done;
                    const bool one_row = hi - lo == 1u;
                    if (want_hint && one_row && hr == 0xffffffffu) {
                        // Locate hint: the interval is one row, i.e. one occurrence at text position p, and the
                        // suffix of row lo starts rem symbols after p (rem are still to be matched to its left), so
                        // p = SA[lo] - rem; the rows this jump passed through after 8, 16, ... steps qualify too,
                        // that many symbols earlier.  If one of them is a sampled row, locate needs no walk.
                        if (is_sampled(ix, lo)) {
                            hr = lo;
                            ho = rem;
                        }
#pragma unroll
                        for (int j = kLevels - 1; j >= 1; j--) {
                            if (hr == 0xffffffffu && best > static_cast<uint32_t>(j)) {
                                const uint32_t tj = j == 1 ? e0.x : (j == 2 ? e0.y : (j == 3 ? e1.x : e1.y));  // t_j
                                const uint32_t mid = group_min<kGroup>(mine ? tj : 0xffffffffu);
                                if (is_sampled(ix, mid)) {
                                    hr = mid;
                                    ho = rem + (best - j) * kJumpSymbols;
                                }
                            }
                        }
                    }
                    if (kMode == 1 && one_row && rem > 0u && rem < kJumpSymbols && best < static_cast<uint32_t>(kCodes)) {
                        // Lazy tail: the next level of the matching row's entry holds the eight symbols that precede
                        // the suffix of row lo; the occurrence survives iff the query's last rem symbols equal the
                        // first rem of them.  2 = survives, 1 = does not, 0 = cannot tell (that level is invalid, or
                        // a symbol outside 1..4 among the query's last rem): then the pair lines decide.
                        uint32_t nxt = e0.z >> 16;  // codes of level best + 1
                        nxt = best == 2u ? (e0.w & 0xffffu) : nxt;
                        if (kHalf2) {
                            nxt = best == 3u ? (e1.w & 0xffffu) : nxt;
                            nxt = best == 4u ? (e1.w >> 16) : nxt;
                        }
                        const bool can = mine && ((valid >> best) & 1u) && tail16 != kNoCode;
                        const uint32_t tmask = (0xffffu << (16u - 2u * rem)) & 0xffffu;
                        const uint32_t verdict = group_max<kGroup>(can ? (((nxt ^ tail16) & tmask) == 0u ? 2u : 1u) : 0u);
                        if (verdict != 0u) {
                            if (kStats) lf_steps += rem;  // (an upper bound when the occurrence does not survive)
                            if (verdict == 2u) {
                                if (hinting && hr == 0xffffffffu) {  // no sampled row on the way: locate walks from lo
                                    hr = lo;
                                    ho = rem;
                                }
                            } else {
                                hi = lo;
                            }
                            rem = 0;
                        }
                    }
                    if (hr != 0xffffffffu) {
                        // parked in LDS until the query's other results are written: a store of its own here made
                        // the L2 fetch the line for a partial write (+0.5 DRAM reads per query measured)
                        if (writer) s_hint[threadIdx.x / kGroup] = make_uint2(hr, ho);
                        status |= 0x80000000u;  // hinted (kept out of the status byte below)
                    }

                } else {
                    // the interval empties within the next 8 steps (or a stored symbol is outside 1..4): the pair
                    // lines find where, which yields the reference's frozen interval (rare: a read that occurs in
                    // the text always matches)
                    jump_ok = false;
                }
                continue;
            }
            if (!second) {
#pragma unroll
                for (int k = 0; k < kChunks; k++) b[k] = a[k];
            }
            if (c2 - 1u < 4u) {  // two LF steps: c1 is consumed first (it precedes the current suffix), then c2
                const uint32_t pair = (c2 - 1u) * 4u + (c1 - 1u);
                const uint32_t bits_x = (c1 & 1u) | ((c1 & 2u) << 7) | ((c1 & 4u) << 14) | ((c2 & 1u) << 24);
                const uint32_t bits_y = ((c2 >> 1) & 1u) | ((c2 & 4u) << 6);
                const uint32_t nx = ~(bits_x * 0xffu);
                const uint32_t ny = ~(bits_y * 0xffu) & 0xffffu;
                uint32_t plo = 0, phi = 0;
#pragma unroll
                for (int k = 0; k < kChunks; k++) {
                    plo += PairTable::pair_partial(a[k], sub + k * kGroup, pair, nx, ny, lo);
                    phi += PairTable::pair_partial(b[k], sub + k * kGroup, pair, nx, ny, hi);
                }
                const uint32_t nlo = group_sum<kGroup>(plo), nhi = group_sum<kGroup>(phi);
                if (nlo != nhi) {
                    if (!kEntrySA && hinting && !(status >> 31) && hi - lo == 1u && nhi - nlo == 1u && rem < (1u << 21)) {
                        // locate hint from a pair step of a one-row interval: the row in between, LF(c1, lo), comes
                        // out of the same line; rem - 1 symbols are still unmatched there (see the jump above)
                        const uint32_t bx1 = (c1 & 1u) | ((c1 & 2u) << 7) | ((c1 & 4u) << 14);
                        const uint32_t nx1 = ~(bx1 * 0xffu) & 0xffffffu;
                        uint32_t pm = 0;
#pragma unroll
                        for (int k = 0; k < kChunks; k++) pm += PairTable::single_partial(a[k], sub + k * kGroup, c1, nx1, lo);
                        const uint32_t mid = group_sum<kGroup>(pm);
                        if (is_sampled(ix, mid)) {
                            if (writer) s_hint[threadIdx.x / kGroup] = make_uint2(mid, rem - 1u);
                            status |= 0x80000000u;
                        }
                    }
                    lo = nlo;
                    hi = nhi;
                    rem -= 2;
                    if (kStats) lf_steps += 2;
                    continue;
                }
                // the interval empties within these two steps: take one step on the same lines so that the
                // frozen interval is the one the reference reports
            }
            {  // one LF step (PairTable::lf1): odd tail, N next, or the step before the interval empties
                const uint32_t bits_x = (c1 & 1u) | ((c1 & 2u) << 7) | ((c1 & 4u) << 14);
                const uint32_t nx = ~(bits_x * 0xffu) & 0xffffffu;
                uint32_t plo = 0, phi = 0;
#pragma unroll
                for (int k = 0; k < kChunks; k++) {
                    plo += PairTable::single_partial(a[k], sub + k * kGroup, c1, nx, lo);
                    phi += PairTable::single_partial(b[k], sub + k * kGroup, c1, nx, hi);
                }
                lo = group_sum<kGroup>(plo);
                hi = group_sum<kGroup>(phi);
                rem--;
                if (kStats) lf_steps++;
            }
        }
        if (writer && !deferred) {
            // a hint describes the one row of a non-empty result; drop it when the search went on and emptied it
            const bool hinted = (status >> 31) && hi - lo == 1u;
            const uint2 hv = hinted ? s_hint[threadIdx.x / kGroup] : make_uint2(0xffffffffu, 0u);
            if (out_rec) {
                uint4 rec;
                rec.x = lo;
                rec.y = hi;
                rec.z = hv.x;
                rec.w = (hv.y & 0xffffffu) | ((status & 0xffu) << 24);
                out_rec[q] = rec;
            }
            // (a cursor that got an empty string, or was stopped earlier, is as it was: nothing to store -- calls over
            // live lists whose reads have mostly ended are made of such cursors)
            const bool unchanged = kMode == 2 && (len == 0u || stopped);
            if (out_start && !unchanged) out_start[q] = lo;
            if (out_end && !unchanged) out_end[q] = hi;
            if (out_count) out_count[q] = hi - lo;
            if (out_status && !unchanged) out_status[q] = static_cast<uint8_t>(status);
            if (out_hint) out_hint[q] = hv;
        }
        if (kMode == 2 && ca.active_out != nullptr && writer && !deferred && lo != hi && (status & 0xffu) == 0u && more_left)
            s_alive[at - base] = static_cast<uint32_t>(q);  // the cursors that can still be extended
        if (kStats && step_stats) {  // the wavefront ran max(iters) iterations for this batch of queries
            uint32_t wave_max = iters;
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = __shfl_xor(wave_max, off);
                wave_max = o > wave_max ? o : wave_max;
            }
            group_iters += iters;
            wave_slots += wave_max;
        }
    }
    }  // phases
    if (kDefer && defer_after != 0u) {  // the parked queries of this range are done
        __syncthreads();
        if (threadIdx.x == 0) s_ndefer = 0;
    }
    if (kMode == 2 && ca.active_out != nullptr)  // the range's live cursors: one atomic, in order
        flush_live_ordered(s_alive, cnt, s_alive_part, &s_alive_base, ca.active_out, ca.n_active_out);
    }  // ranges
    if (kStats && step_stats && writer) {
        atomicAdd(step_stats, static_cast<unsigned long long>(lf_steps));
        atomicAdd(step_stats + 1, group_iters);
        atomicAdd(step_stats + 2, wave_slots);
    }
}

#define GDX_SEARCH_ARGS                                                                                      \
    IndexView ix, const uint8_t *__restrict__ qbuf, const uint64_t *__restrict__ qbeg,                       \
        const uint64_t *__restrict__ qend, uint64_t nq, uint32_t *__restrict__ out_start,                    \
        uint32_t *__restrict__ out_end, uint32_t *__restrict__ out_count, uint8_t *__restrict__ out_status,  \
        unsigned long long *__restrict__ step_stats, uint32_t range, int schedule, uint2 *__restrict__ out_hint, \
        uint4 *__restrict__ out_rec, CursorArgs ca, uint32_t defer_after
#define GDX_SEARCH_FWD \
    ix, qbuf, qbeg, qend, nq, out_start, out_end, out_count, out_status, step_stats, range, schedule, out_hint, out_rec, ca, \
        defer_after

// Register budgets: with the default budget the 8-lane kernel needs 99 SGPRs and the hardware admits only 6-7
// blocks per CU (MI355X_MICROARCH.md residency).  waves_per_eu(8, 8) -> 64 VGPRs / 78 SGPRs, 8 blocks per CU.
// The 4-lane kernel holds two chunks per lane and line.  With the jump levels, hints and ranges it no longer fits
// 64 VGPRs without spilling 19 of them (whose scratch traffic showed up as a million DRAM writes per 10 M reads),
// so it runs at 7 waves per SIMD (72 VGPRs, 5 spilled): 11.1 ms against 11.6 ms at 8 waves and 11.7 ms at 6.
template <int kPolicy, int kJump, int kMode>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void search_pair_kernel8(GDX_SEARCH_ARGS)
{
    search_pair_body<kPolicy, 8, false, kJump, kMode>(GDX_SEARCH_FWD);
}
template <int kPolicy, int kJump, int kMode>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(7, 7))) void search_pair_kernel4(GDX_SEARCH_ARGS)
{
    search_pair_body<kPolicy, 4, false, kJump, kMode>(GDX_SEARCH_FWD);
}
// the same with the straggler pass (plain loads; modes 0 and 1): for texts whose reads often stay wide after the top
// table (launch_search_call decides)
template <int kPolicy, int kJump, int kMode>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void search_pair_defer_kernel8(GDX_SEARCH_ARGS)
{
    search_pair_body<0, 8, false, kJump, kMode, false, true>(GDX_SEARCH_FWD);
}
template <int kPolicy, int kJump, int kMode>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(7, 7))) void search_pair_defer_kernel4(GDX_SEARCH_ARGS)
{
    search_pair_body<0, 4, false, kJump, kMode, false, true>(GDX_SEARCH_FWD);
}
// ---- fast path of the count / locate search ----------------------------------------------------------------------
// What almost every read of a non-repetitive text does is: top table, then jumps, then the lazy tail -- no pair line
// is ever touched.  This kernel does only that, which leaves out the second line of registers, the pair-step
// arithmetic, the N path, hints through LDS and the length order: fewer instructions and eight waves per SIMD.  A
// query it cannot finish that way (no top-table hit, an interval wider than four rows, a symbol outside A C G T among
// the symbols it looks at, fewer than eight symbols left on a multi-row interval, a tail that runs into a sentinel or
// an N of the text) is appended to `leftover` untouched and searched by the general kernel afterwards
// (launch_search_call).  Counts and hits of the queries it finishes are the general kernel's, bit for bit.
// (the few fields of the IndexView it needs, so that the kernel arguments do not eat the SGPR budget of 8+ waves)
// resume states of the seed kernel's list in their packed form (search_seed_kernel4 -> search_fast_kernel4, state_packed):
// {lo, rows << 24 | kStatePacked | symbols left, codes hi, codes lo} carries the 32 symbols in front of the seed as 2-bit
// codes (the one to be consumed next in the top bits of `codes hi`); {lo, kStatePlain | symbols left, hi, 0} an interval
// of 256+ rows; all zero = from the beginning
constexpr uint32_t kStatePacked = 1u << 23, kStatePlain = 1u << 22;
// ... and, for search_verify_kernel4 only (state_packed == 2): {index of the k-mer's record in IndexView::seed_pairs,
// 2 << 24 | kStatePacked | kStatePair | symbols left (1..32), codes hi, codes lo} -- a k-mer on exactly two rows whose seed
// entry names such a record (kSeedPairInfo): the read is decided by that one 32-byte record.  3 << 24 / 4 << 24 in the rows' place:
// the index is that of the k-mer's 64-byte record in IndexView::seed_quads (kSeedQuadInfo)
constexpr uint32_t kStatePair = 1u << 21;

struct FastView {
    const uint2 *top;
    const void *jump;
    const uint8_t *io_to_dense;
    uint32_t top_depth, sa_inv, sa_rot, sa_limit;
    uint32_t perm_code_lo, perm_code_hi, perm_exp_lo, perm_exp_hi, perm_mask;  // IndexView::perm_*
};
__device__ __forceinline__ bool is_sampled(const FastView &ix, uint32_t i)
{
    const uint32_t m = i * ix.sa_inv;
    return __builtin_amdgcn_alignbit(m, m, ix.sa_rot) <= ix.sa_limit;
}

// Instruction count is what bounds this kernel (profiles/r02/experiments.md section 9: its first version, built from
// SpanWindow, issued 479 VALU instructions per wavefront and round of 16 queries at 100 % VALU utilisation, and a
// software pipeline with four queries in flight per lane group changed nothing), hence:
//  * query bytes -> 2-bit codes with v_perm_b32 table lookups on four bytes at once (IndexView::perm_*: an 8-entry
//    table indexed by the low three bits of the byte, and the byte the entry expects, which makes the test exact)
//    instead of sixteen LDS reads and nibble arithmetic per lane; alphabets without such a table use LDS as before;
//  * all eight 16-bit level codes of the window (top-table index, jump levels, lazy tail) come from ONE funnel shift
//    per lane and four DPP broadcasts instead of two ds_bpermute and ~15 VALU per level;
//  * a one-row interval with fewer than eight symbols left and no lookahead code (lengths 16 + 40 k + 1..7 with
//    32-byte entries) is decided by the first level code of its row's entry instead of going to the general kernel.
struct FastWindow {
    uint32_t l0, l1, l2, l3;  // level 2 t | level 2 t + 1 << 16 held by lane t of the group (level j = the 2-bit codes
                              // of the symbols rem - 1 - 8 j .. rem - 8 - 8 j, bits 15:14 = the first of them)
    uint32_t valid8;          // bit k: every symbol of span word k is one of the four searchable ones
    uint32_t s0;              // symbols of a level that lie in its first span word (1 .. 8)
};
// four query bytes -> their 2-bit codes in 8 bits (byte 0 in bits 1:0); `bad` becomes non-zero on any other byte
template <class View>
__device__ __forceinline__ uint32_t fast_pack4(const View &ix, uint32_t c, uint32_t &bad)
{
    const uint32_t sel = c & 0x07070707u;
    const uint32_t code = __builtin_amdgcn_perm(ix.perm_code_hi, ix.perm_code_lo, sel);
    const uint32_t expect = __builtin_amdgcn_perm(ix.perm_exp_hi, ix.perm_exp_lo, sel);
    bad |= (c & ix.perm_mask) ^ expect;
    uint32_t t = (code << 6) | code;
    t = (t << 12) | t;
    return (t >> 18) & 0xffu;
}
__device__ __forceinline__ uint32_t fast_pack4_lds(const uint8_t *s_dense, uint32_t c, uint32_t &bad)
{
    uint32_t out = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t d = static_cast<uint32_t>(s_dense[(c >> (8u * k)) & 0xffu]) - 1u;
        bad |= d & ~3u;
        out |= (d & 3u) << (2u * k);
    }
    return out;
}
// positions the window so that level 0 starts at symbol rem - 1 (rem >= 1): each lane loads and translates two of
// the eight span words (as SpanWindow::load), then the levels are cut out of the group's 128-bit code string
// kXlate: 0 = bytes through the table in LDS, 1 = bytes through the v_perm tables, 2 = packed queries (the span words
// are 16-bit units of the buffer, wbase points at the unit of the query's first symbol and off0 counts symbols)
// the loads of fast_window: this lane's two span words, untranslated (packed queries: both units in .x)
template <int kXlate>
__device__ __forceinline__ u32x4 fast_window_load(const uint64_t *wbase, uint32_t off0, uint32_t rem, uint32_t sub)
{
    const uint32_t b = off0 + rem - 1u;
    const int32_t first = static_cast<int32_t>(b >> 3) - static_cast<int32_t>(sub) * 2;
    u32x4 raw = {0u, 0u, 0u, 0u};  // x, y = word first - 1 (span word 2 sub + 1); z, w = word first (span word 2 sub)
    if (kXlate == 2) {
        const uint16_t *units = reinterpret_cast<const uint16_t *>(wbase);
        const uint32_t u0 = first >= 0 ? units[first] : 0u;
        const uint32_t u1 = first >= 1 ? units[first - 1] : 0u;
        raw.x = u0 | (u1 << 16);
    } else if (first >= 1) {
        raw = *reinterpret_cast<const u32x4 *>(wbase + (first - 1));
    } else if (first == 0) {
        const uint64_t r0 = wbase[0];
        raw.z = static_cast<uint32_t>(r0);
        raw.w = static_cast<uint32_t>(r0 >> 32);
    }
    return raw;
}
template <int kXlate, class View>
__device__ __forceinline__ FastWindow fast_window_finish(const View &ix, const uint8_t *s_dense, u32x4 raw,
                                                         uint32_t off0, uint32_t rem, uint32_t sub)
{
    const uint32_t b = off0 + rem - 1u;
    const int32_t first = static_cast<int32_t>(b >> 3) - static_cast<int32_t>(sub) * 2;
    uint32_t bad0 = 0, bad1 = 0, p;
    if (kXlate == 2) {
        p = raw.x;
    } else {
        if (kXlate == 1) {
            p = fast_pack4(ix, raw.z, bad0) | (fast_pack4(ix, raw.w, bad0) << 8) | (fast_pack4(ix, raw.x, bad1) << 16) |
                (fast_pack4(ix, raw.y, bad1) << 24);
        } else {
            p = fast_pack4_lds(s_dense, raw.z, bad0) | (fast_pack4_lds(s_dense, raw.w, bad0) << 8) |
                (fast_pack4_lds(s_dense, raw.x, bad1) << 16) | (fast_pack4_lds(s_dense, raw.y, bad1) << 24);
        }
    }
    uint32_t m = ((bad0 == 0u && first >= 0) ? 1u : 0u) | ((bad1 == 0u && first >= 1) ? 2u : 0u);
    m <<= 2u * sub;
    m |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(m), 0xB1, 0xF, 0xF, true));
    m |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(m), 0x4E, 0xF, 0xF, true));
    FastWindow w;
    w.valid8 = m;
    w.s0 = (b & 7u) + 1u;
    // lane t: words 2 t (low half of p), 2 t + 1 (high half) and, from lane t + 1, word 2 t + 2
    const uint32_t nxt = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(p), 0xF9, 0xF, 0xF, true));  // quad_perm [1,2,3,3]
    const uint32_t even = __builtin_amdgcn_alignbit(p, p, 16);       // word 2 t << 16 | word 2 t + 1
    const uint32_t odd = (p & 0xffff0000u) | (nxt & 0xffffu);         // word 2 t + 1 << 16 | word 2 t + 2
    const uint32_t sh = 2u * w.s0;                                    // 2 .. 16
    const uint32_t lv = ((even >> sh) & 0xffffu) | ((odd >> sh) << 16);
    w.l0 = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(lv), 0x00, 0xF, 0xF, true));
    w.l1 = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(lv), 0x55, 0xF, 0xF, true));
    w.l2 = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(lv), 0xAA, 0xF, 0xF, true));
    w.l3 = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(lv), 0xFF, 0xF, 0xF, true));
    return w;
}
template <int kXlate, class View>
__device__ __forceinline__ FastWindow fast_window(const View &ix, const uint8_t *s_dense, const uint64_t *wbase,
                                                  uint32_t off0, uint32_t rem, uint32_t sub)
{
    return fast_window_finish<kXlate>(ix, s_dense, fast_window_load<kXlate>(wbase, off0, rem, sub), off0, rem, sub);
}

// kWide: intervals of up to sixteen rows (four per lane, one load round each) instead of four -- chosen per index
// (launch_search_call): on a text without repeats almost no read needs it and the loops cost the others 3 %
template <int kJump, int kXlate, bool kWide>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void search_fast_kernel4(
    FastView ix, const uint8_t *__restrict__ qbuf, const uint64_t *__restrict__ qbeg, const uint64_t *__restrict__ qend,
    uint64_t nq, uint32_t *__restrict__ out_count, uint8_t *__restrict__ out_status, uint4 *__restrict__ out_rec,
    uint32_t range, uint32_t *__restrict__ leftover, uint32_t *__restrict__ n_leftover, uint4 *__restrict__ state,
    // list != null: only the *n_list queries listed (what the seed kernel left over), each from where its state says:
    // {lo, hi, symbols left, 1} = the interval of its last symbols (a seed-table entry), {.., 0} = from the beginning
    const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_list,
    // state_packed != 0 (with list): the states are packed ones (kStatePacked) -- a read with at most 32 symbols left is
    // finished from its state and the jump entries alone, neither its offsets nor its bytes are fetched
    uint32_t state_packed)
{
    constexpr int kGroup = 4;
    constexpr uint32_t kWideRows = kWide ? 16 : kGroup;  // widest interval a round takes
    constexpr int kLevels = kJump == 8 ? 1 : (kJump == 16 ? 2 : 4);
    constexpr int kCodes = kJump == 8 ? 1 : (kJump == 16 ? 3 : 5);
    constexpr bool kHalf2 = kJump == 32;    // entries have a second half {t3, t4, SA[row], c4 | c5 << 16}
    // 32-byte entries carry SA[row] (layout.hpp): the round in which the interval becomes (or is) one row knows the hit's
    // text position, SA[that row] - symbols still to match, and the record is RESOLVED (kernels.hpp) -- no sampled-row
    // candidates to test (64 of the 420 VALU instructions per round in round 2) and no sample read in locate
    constexpr bool kEntrySA = kJump == 32;
    __shared__ uint8_t s_dense[256];
    __shared__ uint32_t s_left[kMaxRange];
    __shared__ uint32_t s_nleft, s_left_base;
    if (kXlate == 0)
        for (int i = threadIdx.x; i < 256; i += kBlock) s_dense[i] = ix.io_to_dense[i];
    if (threadIdx.x == 0) s_nleft = 0;
    __syncthreads();
    const bool writer = (threadIdx.x % kGroup) == 0;
    const uint32_t sub = threadIdx.x & (kGroup - 1u);
    const bool hinting = out_rec != nullptr;
    const uint32_t depth = ix.top_depth;
    if (list != nullptr) nq = *n_list;
    const uint64_t n_ranges = (nq + range - 1) / range;
    for (uint64_t rg = blockIdx.x; rg < n_ranges; rg += gridDim.x) {
        const uint64_t base = rg * range;
        const uint32_t cnt = nq - base < range ? static_cast<uint32_t>(nq - base) : range;
        for (uint32_t slot = threadIdx.x / kGroup; slot < cnt; slot += kBlock / kGroup) {
            const uint32_t q = list != nullptr ? list[base + slot] : static_cast<uint32_t>(base + slot);
            uint4 resume = make_uint4(0u, 0u, 0u, 0u);
            if (list != nullptr && state != nullptr) resume = state[q];
            // (a read that is finished from its packed state never asks where its bytes are)
            const bool from_state = state_packed != 0u && (resume.y & kStatePacked) != 0u && (resume.y & 0x1fffffu) <= 32u;
            uint64_t begin = 0, len = 0;
            const uint64_t *wbase = nullptr;
            uint32_t off0 = 0;
            bool located = false;
            auto locate_query = [&]() {
                begin = qbeg[q];
                len = qend[q] - begin;
                wbase = kXlate == 2 ? reinterpret_cast<const uint64_t *>(reinterpret_cast<const uint16_t *>(qbuf) + (begin >> 3))
                                    : reinterpret_cast<const uint64_t *>(qbuf) + (begin >> 3);
                off0 = static_cast<uint32_t>(begin & 7u);
                located = true;
            };
            if (!from_state) locate_query();
            bool bail = !from_state && !(len >= 16u && len >= depth && len < (1ull << 21));
            uint32_t lo = 0, hi = 0, rem = 0, hr = 0xffffffffu, ho = 0;
            FastWindow w = {0u, 0u, 0u, 0u, 0u, 8u};
            uint32_t shift = 0;  // levels of the window already used up: level i of the round is window level shift + i
            bool fresh = false;  // the window is positioned for the round to come
            bool progressed = false;  // lo, hi, rem describe the search after the top table and whole rounds
            bool masked = false;      // the result is a masked record: hr = mask of surviving rows, ho = symbols left
            bool resolved = false;    // kEntrySA: hr = text position of the hit of the one-row interval
            if (state_packed != 0u && (resume.y & kStatePlain) != 0u) {
                bail = false;
                lo = resume.x;
                hi = resume.z;
                rem = resume.y & 0x1fffffu;
                progressed = true;
            } else if (state_packed != 0u && (resume.y & kStatePacked) != 0u) {
                bail = false;
                lo = resume.x;
                hi = lo + (resume.y >> 24);
                rem = resume.y & 0x1fffffu;
                progressed = true;
                if (from_state) {
                    // the window from the state: levels 0 | 1 and 2 | 3 (eight symbols each, the first of a level in its top
                    // bits), every level inside one clean word
                    w.l0 = __builtin_amdgcn_alignbit(resume.z, resume.z, 16);
                    w.l1 = __builtin_amdgcn_alignbit(resume.w, resume.w, 16);
                    w.l2 = w.l3 = 0u;
                    w.valid8 = 0x0fu;
                    w.s0 = 8u;
                    shift = 0;
                    fresh = true;
                }
            } else if (state_packed == 0u && resume.w == 1u && len < (1ull << 21)) {
                bail = false;  // (a read shorter than the top table is deep may still have a seed)
                lo = resume.x;
                hi = resume.y;
                rem = resume.z;
                progressed = true;
            } else if (!bail) {
                rem = static_cast<uint32_t>(len);
                w = fast_window<kXlate>(ix, s_dense, wbase, off0, rem, sub);
                const uint32_t need = (w.s0 == 8u ? 1u : 3u) | (depth > 8u ? (w.s0 == 8u ? 2u : 6u) : 0u);
                if ((w.valid8 & need) != need) {
                    bail = true;
                } else {
                    const uint2 e = ix.top[__builtin_amdgcn_alignbit(w.l0, w.l0, 16) >> (32u - 2u * depth)];
                    lo = e.x;
                    hi = e.y;
                    rem -= depth;
                    shift = depth >> 3;
                    fresh = (depth & 7u) == 0u;  // depth 8 or 16: the jump levels are levels 1.. or 2.. of this window
                    progressed = true;
                }
            }
            while (!bail && rem > 0u && lo != hi) {
                const uint32_t rows = hi - lo;
                if (rows > kWideRows) {
                    bail = true;
                    break;
                }
                if (!fresh) {
                    if (!located) locate_query();
                    w = fast_window<kXlate>(ix, s_dense, wbase, off0, rem, sub);
                    shift = 0;
                }
                fresh = false;
                // the level string from level `shift` on: a0 = levels 0 | 1 << 16 of the round, a1 = 2 | 3, a2 = 4 | 5
                uint32_t a0, a1, a2;
                if (shift == 2u) {
                    a0 = w.l1;
                    a1 = w.l2;
                    a2 = w.l3;
                } else if (shift == 1u) {
                    a0 = __builtin_amdgcn_alignbit(w.l1, w.l0, 16);
                    a1 = __builtin_amdgcn_alignbit(w.l2, w.l1, 16);
                    a2 = __builtin_amdgcn_alignbit(w.l3, w.l2, 16);
                } else {
                    a0 = w.l0;
                    a1 = w.l1;
                    a2 = w.l2;
                }
                const uint32_t v8 = w.valid8 >> shift;                            // bit i: first word of level i valid
                const uint32_t vl = v8 & (w.s0 == 8u ? 0xffu : (v8 >> 1));        // bit i: level i valid
                const uint32_t n_lv = rem >> 3 < static_cast<uint32_t>(kLevels) ? rem >> 3 : static_cast<uint32_t>(kLevels);
                if (n_lv != 0u && (vl & 1u) == 0u) {  // a symbol outside A C G T among the next eight
                    bail = true;
                    break;
                }
                const uint32_t qa = a0, qb = a1 & 0xffffu, qc = __builtin_amdgcn_alignbit(a2, a1, 16);
                const uint32_t qok = vl & ((1u << n_lv) - 1u);
                // the symbols after the n_lv full levels (fewer than eight), in the top bits of the level that follows
                const uint32_t n_tail = rem - n_lv * kJumpSymbols;  // only used when < 8
                const uint32_t tw = (n_lv >> 1) == 0u ? a0 : ((n_lv >> 1) == 1u ? a1 : a2);
                const uint32_t tail16 = (n_lv & 1u) ? tw >> 16 : tw & 0xffffu;
                const bool tail_ok = ((v8 >> n_lv) & 1u) != 0u && (n_tail <= w.s0 || ((v8 >> (n_lv + 1u)) & 1u) != 0u);

                const uint32_t tmask = (0xffffu << (16u - 2u * (n_tail & 7u))) & 0xffffu;
                if (n_lv == 0u) {
                    // Fewer than eight symbols left: the first code of every row's own entry says whether the row
                    // survives them (the entry tells how many leading symbols of a code that was cut short are real).
                    // The survivors keep their order under the remaining LF steps, so the hits are
                    // SA[row] - rem of the surviving rows in row order: one row is a hint like any other, several are
                    // a masked record {first row, first row + survivors, mask, rem | 1 << 23} (kernels.hpp).
                    if (!tail_ok) {
                        bail = true;
                        break;
                    }
                    uint32_t alive = 0, undecided = 0, own_sa = 0;
                    const u32x4 *ttab = static_cast<const u32x4 *>(ix.jump);
                    // (the second half of the entry -- the same 128-byte line -- only for the SA value of a single row)
                    const unsigned long long m_sa = kEntrySA ? __ballot(hinting && rows == 1u && !resolved) : 0ull;
                    for (uint32_t r0 = lo; r0 < (kWide ? hi : lo + 1u); r0 += kGroup) {  // group-uniform trip count
                        const bool real = r0 + sub < hi;
                        const uint32_t trow = real ? r0 + sub : hi - 1u;
                        const u32x4 *tp = kJump == 8 ? ttab + (trow >> 1) : ttab + static_cast<uint64_t>(trow) * (kJump / 16);
                        u32x4 t0, t1;
                        load_round2<0>(tp, tp + 1, m_sa, t0, t1);
                        if (kEntrySA) own_sa = t1.z;
                        const uint32_t tw = kJump == 8 ? ((trow & 1u) ? t0.w : t0.y) : t0.z;
                        const uint32_t tvalid = kJump == 8 ? tw >> 16 : t0.w >> 16;
                        if (real) {
                            if (n_tail > ((tvalid >> 8) & 0xfu)) undecided = 1u;
                            else if (((tw ^ tail16) & tmask) == 0u) alive |= 1u << (trow - lo);
                        }
                    }
                    if (group_max<kGroup>(undecided) != 0u) {
                        bail = true;
                        break;
                    }
                    alive |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(alive), 0xB1, 0xF, 0xF, true));
                    alive |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(alive), 0x4E, 0xF, 0xF, true));
                    if (rows == 1u) {
                        if (alive != 0u) {
                            if (kEntrySA) {
                                if (hinting && !resolved) {
                                    hr = own_sa - rem;
                                    resolved = true;
                                }
                            } else if (hinting && hr == 0xffffffffu) {
                                hr = lo;
                                ho = rem;
                            }
                        } else {
                            hi = lo;
                        }
                    } else {
                        masked = true;
                        hr = alive;
                        ho = rem;
                        hi = lo + static_cast<uint32_t>(__popc(alive));
                    }
                    rem = 0;
                    break;
                }
                if (kWide && rows > static_cast<uint32_t>(kGroup)) {
                    // A wide interval (a read from a repeat family, or a shallow top table): every lane takes up to four
                    // rows, one load round each.  The rows that match `best` levels map onto [min, max + 1) of their
                    // level-`best` targets as in the narrow case; per level the lane keeps min and max over its rows that
                    // reach it.  No hint and no lazy tail here: they describe one row.
                    uint32_t mn[kLevels], mx[kLevels], reach = 0;
#pragma unroll
                    for (int j = 0; j < kLevels; j++) {
                        mn[j] = 0xffffffffu;
                        mx[j] = 0u;
                    }
                    const u32x4 *wtab = static_cast<const u32x4 *>(ix.jump);
                    for (uint32_t r0 = lo; r0 < hi; r0 += kGroup) {  // group-uniform trip count
                        const bool real = r0 + sub < hi;
                        const uint32_t wrow = real ? r0 + sub : hi - 1u;
                        const u32x4 *wp = kJump == 8 ? wtab + (wrow >> 1) : wtab + static_cast<uint64_t>(wrow) * (kJump / 16);
                        u32x4 w0, w1;
                        load_round2<0>(wp, wp + 1, kHalf2 ? __ballot(true) : 0ull, w0, w1);
                        uint32_t wvalid, wgood;
                        if (kJump == 8) {
                            const uint32_t ew = (wrow & 1u) ? w0.w : w0.y;
                            wvalid = ew >> 16;
                            wgood = ((ew ^ qa) & 0xffffu) == 0u ? 1u : 0u;
                        } else {
                            const uint32_t da = w0.z ^ qa;
                            wvalid = w0.w >> 16;
                            wgood = ((da & 0xffffu) == 0u ? 1u : 0u) | ((da >> 16) == 0u ? 2u : 0u);
                            if (kHalf2) {
                                const uint32_t dc = w1.w ^ qc;
                                wgood |= ((w0.w ^ qb) & 0xffffu) == 0u ? 4u : 0u;
                                wgood |= (dc & 0xffffu) == 0u ? 8u : 0u;
                            }
                        }
                        wgood &= wvalid & qok;
                        const uint32_t wl = real ? static_cast<uint32_t>(__builtin_ctz(~wgood | (1u << kLevels))) : 0u;
                        reach = wl > reach ? wl : reach;
#pragma unroll
                        for (int j = 0; j < kLevels; j++) {
                            const uint32_t tj = kJump == 8 ? ((wrow & 1u) ? w0.z : w0.x)
                                                           : (j == 0 ? w0.x : (j == 1 ? w0.y : (j == 2 ? w1.x : w1.y)));
                            if (wl > static_cast<uint32_t>(j)) {
                                mn[j] = tj < mn[j] ? tj : mn[j];
                                mx[j] = tj > mx[j] ? tj : mx[j];
                            }
                        }
                    }
                    const uint32_t wbest = group_max<kGroup>(reach);
                    if (wbest == 0u) {  // no row survives the next eight symbols
                        hi = lo;
                        break;
                    }
                    uint32_t bmn = mn[0], bmx = mx[0];
#pragma unroll
                    for (int j = 1; j < kLevels; j++) {
                        bmn = wbest == static_cast<uint32_t>(j + 1) ? mn[j] : bmn;
                        bmx = wbest == static_cast<uint32_t>(j + 1) ? mx[j] : bmx;
                    }
                    lo = group_min<kGroup>(bmn);
                    hi = group_max<kGroup>(bmx) + 1u;
                    rem -= wbest * kJumpSymbols;
                    continue;
                }
                const uint32_t row = lo + sub < hi ? lo + sub : hi - 1u;  // spare lanes repeat the last row
                const u32x4 *tab = static_cast<const u32x4 *>(ix.jump);
                const u32x4 *pa = kJump == 8 ? tab + (row >> 1) : tab + static_cast<uint64_t>(row) * (kJump / 16);
                u32x4 e0, e1;
                load_round2<0>(pa, pa + 1, kHalf2 ? __ballot(true) : 0ull, e0, e1);
                uint32_t valid, c1;
                if (kJump == 8) {
                    const uint32_t ew = (row & 1u) ? e0.w : e0.y;
                    c1 = ew & 0xffffu;
                    valid = ew >> 16;
                } else {
                    c1 = e0.z & 0xffffu;
                    valid = e0.w >> 16;
                }
                uint32_t da, db = 0, dc = 0;
                if (kJump == 8) {
                    da = (c1 ^ qa) & 0xffffu;
                } else {
                    da = e0.z ^ qa;
                    db = (e0.w ^ qb) & 0xffffu;
                    if (kHalf2) dc = e1.w ^ qc;
                }
                uint32_t good = (da & 0xffffu) == 0u ? 1u : 0u;
                if (kLevels >= 2) good |= (da >> 16) == 0u ? 2u : 0u;
                if (kHalf2) {
                    good |= db == 0u ? 4u : 0u;
                    good |= (dc & 0xffffu) == 0u ? 8u : 0u;
                }
                good &= valid & qok;
                const uint32_t lvl = static_cast<uint32_t>(__builtin_ctz(~good | (1u << kLevels)));
                const uint32_t best = group_max<kGroup>(lvl);
                if (best == 0u) {  // no row survives the next eight symbols: the count is 0 (mode 1 needs no interval)
                    hi = lo;
                    break;
                }
                const bool mine = lvl == best;
                if (!kEntrySA && hinting && hr == 0xffffffffu && rows == 1u && is_sampled(ix, lo)) {
                    hr = lo;
                    ho = rem;
                }
                uint32_t target = kJump == 8 ? ((row & 1u) ? e0.z : e0.x) : e0.x;
                if (kLevels >= 2) target = best == 2u ? e0.y : target;
                if (kHalf2) {
                    target = best == 3u ? e1.x : target;
                    target = best == 4u ? e1.y : target;
                }
                const uint32_t rem_before = rem;
                lo = group_min<kGroup>(mine ? target : 0xffffffffu);
                hi = group_max<kGroup>(mine ? target : 0u) + 1u;
                rem -= best * kJumpSymbols;
                const bool one_row = hi - lo == 1u;
                if (kEntrySA) {
                    // the one surviving row's own SA value sits in the lane that read its entry: the occurrence starts
                    // rem_before symbols before that row's suffix (and stays there whatever rows the path visits next)
                    if (hinting && one_row && !resolved) {
                        hr = group_max<kGroup>(mine ? e1.z : 0u) - rem_before;
                        resolved = true;
                    }
                } else if (hinting && one_row && hr == 0xffffffffu) {
                    if (is_sampled(ix, lo)) {
                        hr = lo;
                        ho = rem;
                    }
#pragma unroll
                    for (int j = kLevels - 1; j >= 1; j--) {
                        if (hr == 0xffffffffu && best > static_cast<uint32_t>(j)) {
                            const uint32_t tj = j == 1 ? e0.x : (j == 2 ? e0.y : (j == 3 ? e1.x : e1.y));
                            const uint32_t mid = group_min<kGroup>(mine ? tj : 0xffffffffu);
                            if (is_sampled(ix, mid)) {
                                hr = mid;
                                ho = rem + (best - j) * kJumpSymbols;
                            }
                        }
                    }
                }
                // lazy tail: the code that follows the levels just taken holds the next eight symbols of this row's path
                if (one_row && rem > 0u && rem < kJumpSymbols && best < static_cast<uint32_t>(kCodes)) {
                    uint32_t nxt = e0.z >> 16;
                    nxt = best == 2u ? (e0.w & 0xffffu) : nxt;
                    if (kHalf2) {
                        nxt = best == 3u ? (e1.w & 0xffffu) : nxt;
                        nxt = best == 4u ? (e1.w >> 16) : nxt;
                    }
                    const bool can = mine && ((valid >> best) & 1u) && tail_ok;
                    const uint32_t verdict = group_max<kGroup>(can ? (((nxt ^ tail16) & tmask) == 0u ? 2u : 1u) : 0u);
                    if (verdict == 0u) {
                        if (!tail_ok) {
                            bail = true;
                            break;
                        }
                        // the lookahead code is cut short: the next round asks the row's own entry
                    } else {
                        if (verdict == 2u) {
                            if (!kEntrySA && hinting && hr == 0xffffffffu) {  // (kEntrySA: resolved by this round)
                                hr = lo;
                                ho = rem;
                            }
                        } else {
                            hi = lo;
                        }
                        rem = 0;
                    }
                } else if (!one_row && rem > 0u && rem < kJumpSymbols && best < static_cast<uint32_t>(kCodes) && tail_ok) {
                    // The same for SEVERAL surviving rows (a read from a repeat family): every lane whose row matched all
                    // `best` levels holds the lookahead of its own path.  The rows after this jump are [lo, hi) = the
                    // images of those rows in their order, so the ones whose lookahead agrees with the query's last
                    // symbols are the hits: a masked record over [lo, hi) -- without the extra round of entry reads the
                    // tail round above would cost -- or a resolved one when a single row is left.
                    uint32_t nxt = e0.z >> 16;
                    nxt = best == 2u ? (e0.w & 0xffffu) : nxt;
                    if (kHalf2) {
                        nxt = best == 3u ? (e1.w & 0xffffu) : nxt;
                        nxt = best == 4u ? (e1.w >> 16) : nxt;
                    }
                    const bool can = ((valid >> best) & 1u) != 0u;
                    if (group_max<kGroup>(mine && !can ? 1u : 0u) == 0u) {  // every surviving row has a usable lookahead
                        const bool match = mine && ((nxt ^ tail16) & tmask) == 0u;
                        uint32_t alive = match ? 1u << (target - lo) : 0u;
                        alive |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(alive), 0xB1, 0xF, 0xF, true));
                        alive |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(alive), 0x4E, 0xF, 0xF, true));
                        const uint32_t n_alive = static_cast<uint32_t>(__popc(alive));
                        if (kEntrySA && hinting && n_alive == 1u) {
                            hr = group_max<kGroup>(match ? e1.z : 0u) - rem_before;
                            resolved = true;
                            lo += static_cast<uint32_t>(__builtin_ctz(alive));
                            hi = lo + 1u;
                        } else if (n_alive == 0u) {
                            hi = lo;
                        } else if (kEntrySA && hinting && n_alive == 2u) {
                            // two rows are left and each one's entry holds SA of its row: the record takes both occurrences
                            // along (kernels.hpp: a resolved record of two), locate has no suffix-array line to fetch
                            const uint32_t idx = target - lo, at = e1.z - rem_before;
                            const uint32_t f = static_cast<uint32_t>(__builtin_ctz(alive)), g2 = static_cast<uint32_t>(__builtin_ctz(alive & (alive - 1u)));
                            // (kept in the registers the record is written from: {lo, hi, hr} = {second, second + 2, first})
                            hr = group_max<kGroup>(match && idx == f ? at : 0u);
                            lo = group_max<kGroup>(match && idx == g2 ? at : 0u);
                            hi = lo + 2u;
                            resolved = true;
                        } else {
                            masked = true;
                            hr = alive;
                            ho = rem;
                            hi = lo + n_alive;
                        }
                        rem = 0;
                    }
                }
            }
            if (writer) {
                if (bail) {
                    s_left[atomicAdd(&s_nleft, 1u)] = q;
                    // where the general kernel goes on: after the top table and whole rounds, or from the start
                    if (state) state[q] = make_uint4(lo, hi, rem, progressed ? 1u : 0u);
                } else {
                    // (resolved with two slots: the record of two made above)
                    const bool hinted = (kEntrySA ? resolved : hr != 0xffffffffu) && (hi - lo == 1u || (kEntrySA && resolved && hi - lo == 2u));
                    if (out_rec) {
                        if (masked) out_rec[q] = make_uint4(lo, hi, hr, (ho & 0x1fffffu) | kRecMasked);
                        else if (kEntrySA) out_rec[q] = make_uint4(lo, hi, hinted ? hr : 0xffffffffu, hinted ? kRecResolved : 0u);
                        else out_rec[q] = make_uint4(lo, hi, hinted ? hr : 0xffffffffu, hinted ? (ho & 0xffffffu) : 0u);
                    }
                    if (out_count) out_count[q] = hi - lo;
                    if (out_status) out_status[q] = 0;
                }
            }
        }
        // flush the range's leftover queries: one atomic, coalesced stores
        __syncthreads();
        const uint32_t n_left = s_nleft;
        if (threadIdx.x == 0 && n_left != 0u) s_left_base = atomicAdd(n_leftover, n_left);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_left; i += kBlock) leftover[s_left_base + i] = s_left[i];
        __syncthreads();
        if (threadIdx.x == 0) s_nleft = 0;
    }
}

// ---- exact intervals and cursor chunks with the fast kernel's instruction diet --------------------------------------
// cursors_for_many_queries (mode 0) and the cursor extension calls (mode 2) need the reference's interval bit for bit,
// so no lazy tail: what is left after the whole jump levels is stepped on the pair lines, and the eight symbols within
// which an interval empties likewise (the frozen interval is the reference's).  PMC on round 2's general kernel
// (profiles/r03/pmc_general_r2_kernel.md): 730 VALU instructions per round of 16 queries at 89 % VALU busy for the
// fused exact search, ~1000 per round and cursor for a 32-symbol chunk call -- instruction issue, not requests, was
// what the cursor API ran into.  This kernel does what those calls do on clean input with the fast kernel's window
// (v_perm_b32 translation, level codes by one funnel shift) and leaves everything else -- a symbol outside A C G T
// anywhere near what it reads (status codes, N steps on the rank lines), a cursor stopped earlier, queries of 2 M
// symbols -- untouched on a list for the general kernel (launch_search_call), which is the statement of the semantics.
struct ExactView {
    const uint2 *top;
    const void *jump;
    const u32x4 *pair_lines;
    const uint8_t *io_to_dense;
    uint32_t top_depth, n;
    uint32_t lookup_depth;  // the configured lookup table is never read, but its eager symbol check must not be skipped
    uint32_t perm_code_lo, perm_code_hi, perm_exp_lo, perm_exp_hi, perm_mask;  // IndexView::perm_*
    // kText (an index without jump table that holds the text, SA[row] and ISA[position]): IndexView::sa_full, text_units, isa, seed*
    const uint32_t *sa_full, *isa;
    const u32x4 *text_units;
    const u32x4 *seed;
    uint32_t seed_buckets, seed_k, seed_tag_bits;
};

__device__ __forceinline__ uint32_t sel4(uint32_t i, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    uint32_t v = a;
    v = i == 1u ? b : v;
    v = i == 2u ? c : v;
    v = i == 3u ? d : v;
    return v;
}

// kText (round 6; an index WITHOUT jump table that holds the full suffix array, the text units and the inverse suffix array --
// the library's default shape): the jump table's place is taken by the text itself.  An interval of up to four rows that has
// eight or more symbols to go does not step through them on the pair lines, two symbols per fetch: every lane takes a row,
// fetches SA[row] and counts how many of the remaining symbols the text in front of that position matches (32 symbols per
// 64-bit compare, as search_verify_kernel4); the rows that match the most, j symbols, are the rows that survive j LF steps,
// LF keeps their order, and their new rows are ISA[SA[row] - j] -- three dependent fetches for any number of steps.  If j is
// less than what is left, the next step empties the interval and is taken on the pair lines, so that the frozen interval is
// the reference's (cursor.rs:41-48).  Random reads empty within a step or two of becoming narrow, so the route opens only after
// an interval has stayed narrow for two pair rounds (or came in narrow: a cursor's later chunks).  A cursor that is still
// cursor_empty (lib.rs:202-210) and gets seed_k symbols or more starts from its seed-table bucket as a count / locate search
// does: a k-mer that occurs once gives the position, the entry's 32 symbols in front verify the rest of a chunk of up to
// seed_k + 32 symbols, and the row is ISA[position] -- two fetches for a 32-symbol chunk instead of top entry + pair steps.
constexpr uint32_t kTextMinSymbols = 8;
template <int kJump, int kXlate, bool kCursor, bool kText = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((kText && !kCursor) ? 6 : 7, (kText && !kCursor) ? 6 : 7))) void search_exact_kernel4(
    ExactView ix, const uint8_t *__restrict__ qbuf, const uint64_t *__restrict__ qbeg, const uint64_t *__restrict__ qend,
    uint64_t nq, uint32_t *__restrict__ out_start, uint32_t *__restrict__ out_end, uint32_t *__restrict__ out_count,
    uint8_t *__restrict__ out_status, uint32_t range, int schedule, uint32_t *__restrict__ leftover,
    uint32_t *__restrict__ n_leftover, CursorArgs ca)
{
    constexpr int kGroup = 4;
    constexpr int kLevels = kJump == 0 ? 0 : (kJump == 8 ? 1 : (kJump == 16 ? 2 : 4));
    constexpr bool kHalf2 = kJump == 32;
    constexpr uint32_t kRangeCap = kCursor ? kCursorRange : kMaxRange;
    __shared__ uint8_t s_dense[256];
    __shared__ uint32_t s_left[kRangeCap];
    __shared__ uint32_t s_nleft, s_left_base;
    __shared__ uint32_t s_alive[kCursor ? kCursorRange : 1];  // flush_live_ordered
    __shared__ uint32_t s_alive_part[kCursor ? kBlock : 1];
    __shared__ uint32_t s_alive_base;
    __shared__ uint32_t s_stage_q[kCursor ? kBlock : 1], s_stage_lo[kCursor ? kBlock : 1], s_stage_hi[kCursor ? kBlock : 1],
        s_stage_len[kCursor ? kBlock : 1];
    __shared__ uint64_t s_stage_begin[kCursor ? kBlock : 1];
    __shared__ uint32_t s_stage_pos[(kCursor && kText) ? kBlock : 1];  // SA[row] of a cursor that comes in on one row (text route)
    __shared__ uint16_t s_perm[kCursor ? 1 : kMaxRange];  // order_range_by_length (fused searches of mixed lengths)
    __shared__ uint32_t s_cnt[kLenBuckets];
    __shared__ uint32_t s_minmax[2];
    if (kXlate == 0)
        for (int i = threadIdx.x; i < 256; i += kBlock) s_dense[i] = ix.io_to_dense[i];
    if (kCursor)
        for (uint32_t i = threadIdx.x; i < kCursorRange; i += kBlock) s_alive[i] = kDeadCursor;
    if (threadIdx.x == 0) s_nleft = 0;
    __syncthreads();
    const bool writer = (threadIdx.x % kGroup) == 0;
    const uint32_t sub = threadIdx.x & (kGroup - 1u);
    const uint32_t depth = ix.top_depth;
    const uint32_t *active = ca.active_in;
    if (ca.n_active_in != nullptr) nq = *ca.n_active_in;
    const uint64_t n_ranges = (nq + range - 1) / range;
    for (uint64_t rg = blockIdx.x; rg < n_ranges; rg += gridDim.x) {
        const uint64_t base = rg * range;
        const uint32_t cnt = nq - base < range ? static_cast<uint32_t>(nq - base) : range;
        // (a call's cursor strings have one size; fused searches of spread-out lengths run in length order)
        const bool ordered = !kCursor && schedule != 0 &&
                             order_range_by_length(qbeg, qend, active, base, cnt, s_perm, s_cnt, s_minmax);
        // Cursor calls stage 256 cursors at a time: every thread fetches ONE cursor's number, string offsets and state (five
        // gathers that depend on the list entry) into LDS, then the lane groups work from there.  A call was bound by
        // the per-cursor chain of dependent loads -- list -> state and offsets -> bytes -> top / jump entry, with only
        // 16 cursors in flight per wavefront (PMC: neither VALU, 74 % busy, nor requests, 63 % of the ceiling) -- and
        // this takes its first two links out of the chain, 256 at once.
        for (uint32_t s0 = 0; s0 < cnt; s0 += (kCursor ? static_cast<uint32_t>(kBlock) : cnt)) {
        const uint32_t s1 = kCursor ? (cnt - s0 < static_cast<uint32_t>(kBlock) ? cnt : s0 + kBlock) : cnt;
        if (kCursor) {
            __syncthreads();  // the previous stage has been read
            const uint32_t i = s0 + threadIdx.x;
            if (i < s1) {
                const uint32_t cq = active ? active[base + i] : static_cast<uint32_t>(base + i);
                uint64_t cb = qbeg[cq], ce = qend[cq];
                const uint32_t clo = out_start[cq], chi = out_end[cq];
                const uint32_t cst = out_status != nullptr ? out_status[cq] : 0u;
                bool more = true;  // chunk view: the query has symbols left of this chunk
                if (ca.chunk_symbols != 0u) {
                    const uint64_t first = cb, skip = static_cast<uint64_t>(ca.chunk_index) * ca.chunk_symbols;
                    ce = ce - first > skip ? ce - skip : first;
                    cb = ce - first > ca.chunk_symbols ? ce - ca.chunk_symbols : first;
                    more = cb > first;
                }
                const uint64_t clen = ce - cb;
                // stopped earlier (the general kernel knows what to do), or too long for the jump levels: bail
                const bool cbail = cst != 0u || clen >= (1ull << 21);
                s_stage_q[threadIdx.x] = cq;
                s_stage_lo[threadIdx.x] = clo;
                s_stage_hi[threadIdx.x] = chi;
                s_stage_begin[threadIdx.x] = cb;
                s_stage_len[threadIdx.x] = (cbail ? 0u : static_cast<uint32_t>(clen)) | (more ? 1u << 30 : 0u) | (cbail ? 1u << 31 : 0u);
                // (a cursor on one row with enough symbols to come takes the text route: its SA value is fetched here, 256 at
                // once and one link earlier in the chain, instead of by its lane group later)
                if (kText && chi - clo == 1u && !cbail && clen >= kTextMinSymbols) s_stage_pos[threadIdx.x] = ix.sa_full[clo];
            }
            __syncthreads();
        }
        for (uint32_t slot = s0 + threadIdx.x / kGroup; slot < s1; slot += kBlock / kGroup) {
            uint32_t q, lo = 0, hi = ix.n;
            uint64_t begin, len;
            bool more_left = true, bail;
            uint32_t staged_pos = 0;
            bool staged = false;  // staged_pos = SA[lo] of the one row the cursor came in on
            if (kCursor) {
                const uint32_t k = slot - s0;
                const uint32_t packed = s_stage_len[k];
                if (kText) {
                    staged = s_stage_hi[k] - s_stage_lo[k] == 1u && (packed >> 31) == 0u && (packed & 0x3fffffffu) >= kTextMinSymbols;
                    staged_pos = s_stage_pos[k];
                }
                q = s_stage_q[k];
                lo = s_stage_lo[k];
                hi = s_stage_hi[k];
                begin = s_stage_begin[k];
                len = packed & 0x3fffffffu;
                more_left = ((packed >> 30) & 1u) != 0u;
                bail = (packed >> 31) != 0u;
            } else {
                const uint64_t at = base + (ordered ? s_perm[slot] : slot);
                q = active ? active[at] : static_cast<uint32_t>(at);
                begin = qbeg[q];
                len = qend[q] - begin;
                bail = len >= (1ull << 21);
            }
            const bool empty_cursor = lo == 0u && hi == ix.n;  // cursor_empty (lib.rs:202-210): the top table applies
            staged = staged && !empty_cursor;  // (a one-row collection: the seed / top table routes change the interval first)
            uint32_t rem = bail ? 0u : static_cast<uint32_t>(len);
            const uint64_t *wbase = query_words<kXlate>(qbuf, begin);  // (kXlate 2: 2-bit codes, offsets count symbols)
            const uint32_t off0 = static_cast<uint32_t>(begin & 7u);
            FastWindow w = {0u, 0u, 0u, 0u, 0u, 8u};
            uint32_t shift = 0;   // whole levels of the window used up
            uint32_t part = 0;    // symbols of level `shift` used up by pair steps
            bool fresh = false;   // the window is positioned: symbol rem - 1 is symbol `part` of its level `shift`
            bool jump_ok = kJump != 0 && ix.jump != nullptr;
            bool text_ok = kText;  // the text route is open for this query
            // pair rounds begun on up to kGroup rows (a cursor that comes in narrow has shown its rows to be real)
            uint32_t narrow = (kText && kCursor && !empty_cursor && hi - lo <= static_cast<uint32_t>(kGroup)) ? 2u : 0u;
            bool seeded = false;   // the seed table has given this cursor its first interval
            if (kText && kCursor && !bail && empty_cursor && ix.seed != nullptr && rem >= ix.seed_k) {
                w = fast_window<kXlate>(ix, s_dense, wbase, off0, rem, sub);
                const uint32_t k = ix.seed_k;
                const uint32_t n_look = rem < k + 32u ? rem : k + 32u;
                const uint32_t n_words = n_look <= w.s0 ? 1u : 1u + ((n_look - w.s0 + 7u) >> 3);
                if ((w.valid8 & ((1u << n_words) - 1u)) == (1u << n_words) - 1u) {
                    // (the window as one string, the chunk's last symbol in the top bits of w0: search_verify_kernel4)
                    const uint32_t w0 = __builtin_amdgcn_alignbit(w.l0, w.l0, 16), w1 = __builtin_amdgcn_alignbit(w.l1, w.l1, 16);
                    const uint32_t w2 = __builtin_amdgcn_alignbit(w.l2, w.l2, 16), w3 = __builtin_amdgcn_alignbit(w.l3, w.l3, 16);
                    const uint64_t key = ((static_cast<uint64_t>(w0) << 32) | w1) >> (64u - 2u * k);
                    uint32_t tag;
                    uint32_t b = seed_home(key, ix.seed_tag_bits, ix.seed_buckets, tag);
                    uint32_t ex, ey, ez, ew;
                    for (uint32_t d = 0;; d++) {
                        const u32x4 *bp = ix.seed + (static_cast<uint64_t>(b) << 3) + 2u * sub;
                        const u32x4 e0 = bp[0], e1 = bp[1];
                        const uint32_t want = tag | (d << kSeedDispShift);
                        const bool m0 = (e0.x & kSeedMatchMask) == want, m1 = (e1.x & kSeedMatchMask) == want;
                        const u32x4 es = m0 ? e0 : e1;
                        const bool mt = m0 || m1;
                        ex = mt ? (es.x | kSeedFound) : (e0.x & kSeedOverflow);
                        ey = mt ? es.y : 0u;
                        ez = mt ? es.z : 0u;
                        ew = mt ? es.w : 0u;
                        ex |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ex), 0xB1, 0xF, 0xF, true));
                        ey |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ey), 0xB1, 0xF, 0xF, true));
                        ez |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ez), 0xB1, 0xF, 0xF, true));
                        ew |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ew), 0xB1, 0xF, 0xF, true));
                        ex |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ex), 0x4E, 0xF, 0xF, true));
                        ey |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ey), 0x4E, 0xF, 0xF, true));
                        ez |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ez), 0x4E, 0xF, 0xF, true));
                        ew |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ew), 0x4E, 0xF, 0xF, true));
                        if ((ex & kSeedFound) != 0u || (ex & kSeedOverflow) == 0u || d >= kSeedMaxDisp) break;
                        b = b + 1u == ix.seed_buckets ? 0u : b + 1u;
                    }
                    if ((ex & kSeedFound) != 0u && (ex & kSeedKind) != 0u) {  // several rows: the k-mer's interval, on from there
                        lo = ey;
                        hi = ez;
                        rem -= k;
                        seeded = true;
                    } else if ((ex & kSeedFound) != 0u) {
                        // one row, at text position ey: how many of the (up to 32) chunk symbols in front of the seed the
                        // entry's symbols in front of that position match
                        const uint32_t front = rem - k, n_v = front < 32u ? front : 32u;
                        uint32_t qh, ql;
                        const uint32_t sh = 32u - 2u * (k & 15u);
                        if (k == 16u) {
                            qh = w1;
                            ql = w2;
                        } else if (k < 16u) {
                            qh = __builtin_amdgcn_alignbit(w0, w1, sh);
                            ql = __builtin_amdgcn_alignbit(w1, w2, sh);
                        } else {
                            qh = __builtin_amdgcn_alignbit(w1, w2, sh);
                            ql = __builtin_amdgcn_alignbit(w2, w3, sh);
                        }
                        const uint64_t vm64 = n_v == 32u ? ~0ull : ~(~0ull >> (2u * n_v));
                        const uint64_t diff = (((static_cast<uint64_t>(qh) << 32) | ql) ^ ((static_cast<uint64_t>(ew) << 32) | ez)) & vm64;
                        const uint32_t v_code = (ex >> kSeedPartialShift) & 3u;
                        const uint32_t n_text = (ex & kSeedPartial) == 0u ? 32u : (v_code == 0u ? (ez & 63u) : 29u + v_code);
                        uint32_t got = diff != 0ull ? static_cast<uint32_t>(__builtin_clzll(diff)) >> 1 : n_v;
                        got = got < n_text ? got : n_text;
                        got = got < ey ? got : ey;
                        const uint32_t row = ix.isa[ey - got];
                        lo = row;
                        hi = row + 1u;
                        rem = front - got;
                        seeded = true;
                        narrow = 2u;
                        if (got < n_v) text_ok = false;  // the next symbol does not match: one step on the pair lines empties the interval
                    }
                    // (else: the k-mer does not occur; the frozen interval is found from the top table on)
                }
                fresh = false;
            }
            if (seeded) {
                // (the window is positioned anew by whatever comes next)
            } else if (!bail && empty_cursor && ix.top != nullptr && rem >= 16u && rem >= depth) {
                w = fast_window<kXlate>(ix, s_dense, wbase, off0, rem, sub);
                const uint32_t need = (w.s0 == 8u ? 1u : 3u) | (depth > 8u ? (w.s0 == 8u ? 2u : 6u) : 0u);
                if ((w.valid8 & need) != need) {
                    bail = true;
                } else {
                    const uint2 e = ix.top[__builtin_amdgcn_alignbit(w.l0, w.l0, 16) >> (32u - 2u * depth)];
                    lo = e.x;  // (the entry of an absent D-mer is its frozen interval, as the reference's lookup tables)
                    hi = e.y;
                    rem -= depth;
                    shift = depth >> 3;
                    part = depth & 7u;
                    fresh = true;
                }
            } else if (!kCursor && ix.lookup_depth != 0u) {
                // the reference checks every symbol of its lookup suffix before anything else (lookup_table.rs:99-113);
                // the steps below are lazy, so a query that does not pass the top table's check goes to the general kernel
                bail = true;
            }
            while (!bail && rem > 0u && lo != hi) {
                const uint32_t rows = hi - lo;
                if (kText && text_ok && rows <= static_cast<uint32_t>(kGroup) && rem >= kTextMinSymbols && narrow >= 2u) {
                    const bool real = sub < rows;
                    const uint32_t pos = staged ? staged_pos : ix.sa_full[real ? lo + sub : lo];
                    staged = false;  // (it was the SA value of the row the cursor came in on)
                    uint32_t m = 0;  // LF steps this lane's row survives = symbols of the text in front of pos that are the query's
                    bool going = real;
                    uint32_t rem_v = rem;
                    bool reuse = fresh && part == 0u && shift <= 3u;  // the window still serves the first pass
                    while (rem_v > 0u && group_max<kGroup>(going ? 1u : 0u) != 0u) {
                        // (a lane that is going has matched rem - rem_v symbols: pos >= rem - rem_v, and the pad units in front of
                        // the text read as "no match"; the text loads go out before the window's, which do not depend on them)
                        const uint64_t s0t = static_cast<uint64_t>(pos) - (rem - rem_v) + 32u * kTextPadUnits - 32u;
                        const uint32_t tb = static_cast<uint32_t>(s0t & 31u);
                        const u32x4 *tu = ix.text_units + (s0t >> 5);
                        const u32x4 u0 = going ? tu[0] : u32x4{0u, 0u, 0u, 0u};
                        const u32x4 u1 = (going && tb != 0u) ? tu[1] : u32x4{0u, 0u, 0u, 0u};
                        if (!reuse) {
                            w = fast_window<kXlate>(ix, s_dense, wbase, off0, rem_v, sub);
                            shift = 0;
                            part = 0;
                        }
                        reuse = false;
                        const uint32_t n_v = rem_v < 32u ? rem_v : 32u;
                        const uint32_t v8 = w.valid8 >> shift;
                        const uint32_t vl = v8 & (w.s0 == 8u ? 0xffu : (v8 >> 1));
                        const uint32_t n_full = n_v >> 3, n_tail = n_v & 7u;
                        bool clean = (vl & ((1u << n_full) - 1u)) == ((1u << n_full) - 1u);
                        if (n_tail != 0u) clean = clean && ((v8 >> n_full) & 1u) != 0u && (n_tail <= w.s0 || ((v8 >> (n_full + 1u)) & 1u) != 0u);
                        if (!clean) {  // a symbol outside A C G T ahead: the general kernel's
                            bail = true;
                            break;
                        }
                        uint32_t a0, a1;  // levels shift, shift + 1 | shift + 2, shift + 3 of the window
                        if (shift == 0u) {
                            a0 = w.l0;
                            a1 = w.l1;
                        } else if (shift == 1u) {
                            a0 = __builtin_amdgcn_alignbit(w.l1, w.l0, 16);
                            a1 = __builtin_amdgcn_alignbit(w.l2, w.l1, 16);
                        } else if (shift == 2u) {
                            a0 = w.l1;
                            a1 = w.l2;
                        } else {
                            a0 = __builtin_amdgcn_alignbit(w.l2, w.l1, 16);
                            a1 = __builtin_amdgcn_alignbit(w.l3, w.l2, 16);
                        }
                        // the 32 symbols [rem_v - 32, rem_v) in text order, the next one to be consumed in the top bits
                        const uint64_t qcode = (static_cast<uint64_t>(__builtin_amdgcn_alignbit(a0, a0, 16)) << 32) |
                                               static_cast<uint64_t>(__builtin_amdgcn_alignbit(a1, a1, 16));
                        const uint64_t c0 = static_cast<uint64_t>(u0.x) | (static_cast<uint64_t>(u0.y) << 32);
                        const uint64_t c1 = static_cast<uint64_t>(u1.x) | (static_cast<uint64_t>(u1.y) << 32);
                        const uint64_t tcode = tb ? (c0 >> (2u * tb)) | (c1 << (64u - 2u * tb)) : c0;
                        const uint32_t tmask = tb ? (u0.z >> tb) | (u1.z << (32u - tb)) : u0.z;
                        const uint64_t vm64 = n_v == 32u ? ~0ull : ~0ull << (2u * (32u - n_v));
                        const uint32_t vm32 = n_v == 32u ? ~0u : ~0u << (32u - n_v);
                        const uint64_t diff = (qcode ^ tcode) & vm64;
                        const uint32_t other = tmask & vm32;  // a sentinel, an N, ... in the text
                        uint32_t got = diff != 0ull ? static_cast<uint32_t>(__builtin_clzll(diff)) >> 1 : n_v;
                        const uint32_t got_m = other != 0u ? static_cast<uint32_t>(__builtin_clz(other)) : n_v;
                        got = got < got_m ? got : got_m;
                        if (going) {
                            m += got;
                            going = got == n_v;
                        }
                        rem_v -= n_v;
                    }
                    if (bail) break;
                    const uint32_t best = group_max<kGroup>(real ? m : 0u);
                    fresh = false;
                    if (best == 0u) {  // no row survives the next step: the pair lines make the frozen interval
                        text_ok = false;
                        continue;
                    }
                    const bool mine = real && m == best;
                    const uint32_t row = mine ? ix.isa[pos - best] : 0u;
                    lo = group_min<kGroup>(mine ? row : 0xffffffffu);
                    hi = group_max<kGroup>(mine ? row : 0u) + 1u;
                    rem -= best;
                    if (rem > 0u) text_ok = false;  // (every surviving row mismatches the next symbol)
                    continue;
                }
                if (kText && rows <= static_cast<uint32_t>(kGroup)) narrow++;
                const bool jumping = jump_ok && rows <= static_cast<uint32_t>(kGroup) && rem >= kJumpSymbols;
                // the window must hold what this round reads: whole levels from an aligned position for a jump (levels
                // 0 .. 6 of a window are always inside its eight words), one or two symbols for a pair step
                if (!fresh || (jumping ? (part != 0u || shift > 2u) : shift > 6u)) {
                    w = fast_window<kXlate>(ix, s_dense, wbase, off0, rem, sub);
                    shift = 0;
                    part = 0;
                }
                fresh = true;
                const uint32_t v8 = w.valid8 >> shift;                      // bit i: first word of level shift + i valid
                const uint32_t vl = v8 & (w.s0 == 8u ? 0xffu : (v8 >> 1));  // bit i: level shift + i valid
                if (jumping) {
                    if ((vl & 1u) == 0u) {  // a symbol outside A C G T among the next eight
                        bail = true;
                        break;
                    }
                    uint32_t a0, a1, a2;
                    if (shift == 2u) {
                        a0 = w.l1;
                        a1 = w.l2;
                        a2 = w.l3;
                    } else if (shift == 1u) {
                        a0 = __builtin_amdgcn_alignbit(w.l1, w.l0, 16);
                        a1 = __builtin_amdgcn_alignbit(w.l2, w.l1, 16);
                        a2 = __builtin_amdgcn_alignbit(w.l3, w.l2, 16);
                    } else {
                        a0 = w.l0;
                        a1 = w.l1;
                        a2 = w.l2;
                    }
                    const uint32_t n_lv = rem >> 3 < static_cast<uint32_t>(kLevels) ? rem >> 3 : static_cast<uint32_t>(kLevels);
                    const uint32_t qa = a0, qb = a1 & 0xffffu, qc = __builtin_amdgcn_alignbit(a2, a1, 16);
                    const uint32_t qok = vl & ((1u << n_lv) - 1u);
                    const uint32_t row = lo + sub < hi ? lo + sub : hi - 1u;  // spare lanes repeat the last row
                    const u32x4 *tab = static_cast<const u32x4 *>(ix.jump);
                    const u32x4 *pa = kJump == 8 ? tab + (row >> 1) : tab + static_cast<uint64_t>(row) * (kJump / 16);
                    u32x4 e0, e1;
                    load_round2<0>(pa, pa + 1, kHalf2 ? __ballot(true) : 0ull, e0, e1);
                    uint32_t valid, da, db = 0, dc = 0;
                    if (kJump == 8) {
                        const uint32_t ew = (row & 1u) ? e0.w : e0.y;
                        da = (ew ^ qa) & 0xffffu;
                        valid = ew >> 16;
                    } else {
                        da = e0.z ^ qa;
                        db = (e0.w ^ qb) & 0xffffu;
                        valid = e0.w >> 16;
                        if (kHalf2) dc = e1.w ^ qc;
                    }
                    uint32_t good = (da & 0xffffu) == 0u ? 1u : 0u;
                    if (kLevels >= 2) good |= (da >> 16) == 0u ? 2u : 0u;
                    if (kHalf2) {
                        good |= db == 0u ? 4u : 0u;
                        good |= (dc & 0xffffu) == 0u ? 8u : 0u;
                    }
                    good &= valid & qok;
                    const uint32_t lvl = static_cast<uint32_t>(__builtin_ctz(~good | (1u << kLevels)));
                    const uint32_t best = group_max<kGroup>(lvl);
                    if (best == 0u) {
                        // the interval empties within the next eight steps, or a stored / query symbol there is outside
                        // A C G T: the pair lines find where (the frozen interval must be the reference's)
                        jump_ok = false;
                        continue;
                    }
                    const bool mine = lvl == best;
                    uint32_t target = kJump == 8 ? ((row & 1u) ? e0.z : e0.x) : e0.x;
                    if (kLevels >= 2) target = best == 2u ? e0.y : target;
                    if (kHalf2) {
                        target = best == 3u ? e1.x : target;
                        target = best == 4u ? e1.y : target;
                    }
                    // the rows that match `best` levels are the ones that survive these LF steps, and LF keeps their order
                    lo = group_min<kGroup>(mine ? target : 0xffffffffu);
                    hi = group_max<kGroup>(mine ? target : 0u) + 1u;
                    rem -= best * kJumpSymbols;
                    shift += best;
                    continue;
                }
                // one pair-line round: two LF steps, or one (odd tail, or the step at which the interval empties)
                const uint32_t e = sel4(shift >> 1, w.l0, w.l1, w.l2, w.l3);
                const uint32_t f = sel4((shift >> 1) + 1u, w.l0, w.l1, w.l2, w.l3);
                const uint32_t lv2 = (shift & 1u) ? ((e & 0xffff0000u) | (f & 0xffffu)) : __builtin_amdgcn_alignbit(e, e, 16);
                const uint32_t x = lv2 << (2u * part);  // bits 31:30 = symbol rem - 1, 29:28 = symbol rem - 2
                const bool two = rem >= 2u;
                {
                    // symbol t of the level string from level `shift` on lies in span word shift + (t >= s0): only the
                    // words the one or two symbols of this round come from have to be clean (a tail near the query's
                    // first byte reaches into the word before it, which may hold anything)
                    const uint32_t t_last = part + (two ? 1u : 0u);
                    const uint32_t need = (part < w.s0 ? 1u : 2u) | (t_last < w.s0 ? 1u : 2u);
                    if ((v8 & need) != need) {
                        bail = true;
                        break;
                    }
                }
                const uint32_t c1 = (x >> 30) + 1u, c2 = two ? ((x >> 28) & 3u) + 1u : 0u;
                const uint32_t line_lo = lo >> kPairLineShift, line_hi = hi >> kPairLineShift;
                const bool second = line_hi != line_lo;
                const u32x4 *pa = ix.pair_lines + (static_cast<uint64_t>(line_lo) << 3) + sub;
                const u32x4 *pb = ix.pair_lines + (static_cast<uint64_t>(line_hi) << 3) + sub;
                u32x4 a[2], b[2];
                load_round4<0>(pa, pa + kGroup, pb, pb + kGroup, __ballot(true), __ballot(second), a[0], a[1], b[0], b[1]);
                if (!second) {
                    b[0] = a[0];
                    b[1] = a[1];
                }
                bool stepped = false;
                if (two) {  // c1 is consumed first (it precedes the current suffix), then c2
                    const uint32_t pair = (c2 - 1u) * 4u + (c1 - 1u);
                    const uint32_t bits_x = (c1 & 1u) | ((c1 & 2u) << 7) | ((c1 & 4u) << 14) | ((c2 & 1u) << 24);
                    const uint32_t bits_y = ((c2 >> 1) & 1u) | ((c2 & 4u) << 6);
                    const uint32_t nx = ~(bits_x * 0xffu);
                    const uint32_t ny = ~(bits_y * 0xffu) & 0xffffu;
                    uint32_t plo = 0, phi = 0;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        plo += PairTable::pair_partial(a[k], sub + k * kGroup, pair, nx, ny, lo);
                        phi += PairTable::pair_partial(b[k], sub + k * kGroup, pair, nx, ny, hi);
                    }
                    const uint32_t nlo = quad_sum(plo), nhi = quad_sum(phi);
                    if (nlo != nhi) {
                        lo = nlo;
                        hi = nhi;
                        rem -= 2u;
                        part += 2u;
                        stepped = true;
                    }
                    // else: the interval empties within these two steps -- one step on the same lines, so that the
                    // frozen interval is the one the reference reports
                }
                if (!stepped) {
                    const uint32_t bits_x = (c1 & 1u) | ((c1 & 2u) << 7) | ((c1 & 4u) << 14);
                    const uint32_t nx = ~(bits_x * 0xffu) & 0xffffffu;
                    uint32_t plo = 0, phi = 0;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        plo += PairTable::single_partial(a[k], sub + k * kGroup, c1, nx, lo);
                        phi += PairTable::single_partial(b[k], sub + k * kGroup, c1, nx, hi);
                    }
                    lo = quad_sum(plo);
                    hi = quad_sum(phi);
                    rem -= 1u;
                    part += 1u;
                }
                shift += part >> 3;
                part &= 7u;
            }
            if (writer) {
                if (bail) {
                    s_left[atomicAdd(&s_nleft, 1u)] = q;
                } else {
                    const bool unchanged = kCursor && len == 0u;  // a cursor that got an empty string is as it was
                    if (out_start && !unchanged) out_start[q] = lo;
                    if (out_end && !unchanged) out_end[q] = hi;
                    if (out_count) out_count[q] = hi - lo;
                    if (!kCursor && out_status) out_status[q] = 0;
                    if (kCursor && ca.active_out != nullptr && lo != hi && more_left) s_alive[slot] = q;
                }
            }
        }
        }  // stages
        // the range's leftover queries: one atomic, coalesced stores; its live cursors: one atomic, in position order
        __syncthreads();
        const uint32_t n_left = s_nleft;
        if (threadIdx.x == 0 && n_left != 0u) s_left_base = atomicAdd(n_leftover, n_left);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_left; i += kBlock) leftover[s_left_base + i] = s_left[i];
        __syncthreads();
        if (threadIdx.x == 0) s_nleft = 0;
        if (kCursor && ca.active_out != nullptr)
            flush_live_ordered(s_alive, cnt, s_alive_part, &s_alive_base, ca.active_out, ca.n_active_out);
    }
}

// ---- count / locate searches by comparing with the text (the low-memory alternative to the jump table) ---------------
// On an index with text units (IndexView::text_units) a search that is down to a few rows does not walk the rest of the
// query through LF steps: every lane of the group takes one row, fetches SA[row] -- from the full suffix array, from the
// 32-byte jump entry, or by the locate walk to a sampled row -- and compares the query's remaining symbols with the text
// in front of that position, 32 symbols per 64-bit compare.  The rows that match are exactly the rows that survive the
// remaining LF steps, LF keeps their order, and the hit of a surviving row is SA[row] - symbols left: one row becomes a
// RESOLVED record, several a masked one (kernels.hpp), precisely what search_fast_kernel4 writes.  Requests per read:
// top entry + SA + one text line per row, whatever the read length -- against one jump entry per 32 symbols, at a
// fraction of the memory (4 bits per symbol instead of 32 bytes per row).  Intervals wider than `max_rows` are first
// narrowed by LF steps on the rank lines (no pair lines needed).  What it cannot finish -- a symbol outside A C G T,
// a query shorter than the top table is deep -- is listed for the general kernel.
struct VerifyView {
    const uint2 *top;
    const u32x4 *text_units;
    const uint32_t *sa_full;
    const void *jump;  // 32-byte entries (SA in word 6) or null
    const u32x4 *lines;
    const uint32_t *sb_offsets, *count, *sa_samples, *border_keys, *border_vals;
    const uint8_t *io_to_dense;
    uint32_t top_depth, n, n_texts, sa_inv, sa_rot, sa_limit, max_rows;
    uint32_t perm_code_lo, perm_code_hi, perm_exp_lo, perm_exp_hi, perm_mask;  // IndexView::perm_*
    const u32x4 *seed;  // kSeed: IndexView::seed*
    uint32_t seed_buckets, seed_k, seed_tag_bits;
    const u32x4 *seed_pairs, *seed_quads;  // IndexView::seed_pairs / seed_quads (kStatePair states)
};

// kSeed: the seed table instead of the top table (layout.hpp): ONE 128-byte bucket per read answers the last seed_k
// symbols, and when that k-mer occurs once in the text -- nearly every read of a text without repeats -- the entry also
// holds the position and the 32 symbols in front of it: a read of up to seed_k + 32 symbols is counted and located with
// that single fetch (1.5 DRAM requests per len-50 read with its query bytes, against 2.7 through top and jump table),
// a longer one goes on against the text units.  A k-mer on several rows hands over its interval and the search
// continues as after a top table; an absent one is known to be absent.
template <int kXlate, bool kSeed>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(7, 7))) void search_verify_kernel4(
    VerifyView vv, const uint8_t *__restrict__ qbuf, const uint64_t *__restrict__ qbeg, const uint64_t *__restrict__ qend,
    uint64_t nq, uint32_t *__restrict__ out_count, uint8_t *__restrict__ out_status, uint4 *out_rec,
    uint32_t range, uint32_t *__restrict__ leftover, uint32_t *__restrict__ n_leftover,
    const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_list,  // list != null: only the *n_list queries listed
    uint32_t ulen,  // uniform batch (query_begin)
    // kSeed, behind the seed table's own kernel (round 6): where that kernel left a listed read -- {lo, hi, symbols in front of the
    // seed, 1} = its k-mer's interval (the bucket is not fetched again), else from the beginning
    const uint4 *state,  // (the states lie in the record slots: no __restrict__ here nor on out_rec)
    // != 0: an interval wider than max_rows is not narrowed here, one rank-line step per symbol and sixteen reads waiting for the
    // widest: the read goes to the general kernel's list with its state, which steps two symbols per pair-line fetch and has
    // nothing but such reads in its wavefronts
    uint32_t wide_to_list,
    // != 0: the states are the packed ones of the seed kernels (kStatePacked / kStatePlain): a read with at most 32 symbols in
    // front of its seed brings them along as 2-bit codes, and neither its offsets nor its bytes are fetched again -- one of the
    // four DRAM lines a read from a repeat cost here.  A read this kernel hands on gets the plain state {lo, hi, symbols, 1} back
    uint32_t state_packed)
{
    constexpr int kGroup = 4;
    __shared__ uint8_t s_dense[256];
    __shared__ uint32_t s_count[8];
    __shared__ uint32_t s_left[kMaxRange];
    __shared__ uint32_t s_nleft, s_left_base;
    // (a list's length is only known here: the grid is sized for the worst case, and on a text without repeats all but a
    // few of its blocks have nothing to do -- they leave before they touch anything)
    if (list != nullptr) nq = *n_list;
    if (static_cast<uint64_t>(blockIdx.x) * range >= nq) return;
    if (kXlate == 0)
        for (int i = threadIdx.x; i < 256; i += kBlock) s_dense[i] = vv.io_to_dense[i];
    if (threadIdx.x < 8) s_count[threadIdx.x] = threadIdx.x < 6 ? vv.count[threadIdx.x] : 0u;
    if (threadIdx.x == 0) s_nleft = 0;
    __syncthreads();
    IndexView ix{};  // what QuadLineTable / LineTable / sampled_slot read
    ix.lines = vv.lines;
    ix.sb_offsets = vv.sb_offsets;
    ix.sa_inv = vv.sa_inv;
    ix.sa_rot = vv.sa_rot;
    ix.sa_limit = vv.sa_limit;
    const bool writer = (threadIdx.x % kGroup) == 0;
    const uint32_t sub = threadIdx.x & (kGroup - 1u);
    const uint32_t depth = vv.top_depth;
    const uint64_t n_ranges = (nq + range - 1) / range;
    for (uint64_t rg = blockIdx.x; rg < n_ranges; rg += gridDim.x) {
        const uint64_t base = rg * range;
        const uint32_t cnt = nq - base < range ? static_cast<uint32_t>(nq - base) : range;
        for (uint32_t slot = threadIdx.x / kGroup; slot < cnt; slot += kBlock / kGroup) {
            const uint32_t q = list != nullptr ? list[base + slot] : static_cast<uint32_t>(base + slot);
            uint4 st = make_uint4(0u, 0u, 0u, 0u);
            if (kSeed && state != nullptr) st = state[q];
            // (a read that is finished from its packed state never asks where its bytes are: the seed kernel has checked its length)
            const bool from_state = kSeed && state != nullptr && state_packed != 0u && wide_to_list != 0u && (st.y & kStatePacked) != 0u &&
                                    (st.y & 0x1fffffu) <= 32u;
            const uint64_t begin = from_state ? 0ull : query_begin(qbeg, ulen, q);
            const uint64_t len = from_state ? 0ull : query_end(qend, ulen, q) - begin;
            bool bail = from_state ? false
                                   : (kSeed ? !(len >= vv.seed_k && len < (1ull << 21)) : !(len >= 16u && len >= depth && len < (1ull << 21)));
            uint32_t lo = 0, hi = 0, rem = 0;
            const uint64_t *wbase = query_words<kXlate>(qbuf, begin);
            const uint32_t off0 = static_cast<uint32_t>(begin & 7u);
            FastWindow w = {0u, 0u, 0u, 0u, 0u, 8u};
            uint32_t shift = 0, part = 0;  // levels / symbols of level `shift` of the window already used up
            bool single = false;           // kSeed: the k-mer occurs once, `pos` is where (no row is known)
            bool single_ok = false;        // ... and the (up to) 32 symbols in front of it are the query's
            uint32_t pos = 0;              // SA of this lane's row
            bool resumed = false;
            const uint64_t state_codes = (static_cast<uint64_t>(st.z) << 32) | st.w;  // from_state: the symbols in front of the seed
            if (from_state && (st.y & kStatePair) != 0u && (st.y >> 24) > 2u) {
                // three or four copies: the same from a 64-byte record, lane `sub` of the group decides row `sub`
                const u32x4 *qr = vv.seed_quads + 4ull * st.x;
                const u32x4 q0 = qr[0], q1 = qr[1], q2 = qr[2], q3 = qr[3];
                const uint32_t n_v = st.y & 0x1fffffu, rows_q = st.y >> 24;
                const uint64_t vm64 = n_v == 32u ? ~0ull : ~0ull << (2u * (32u - n_v));
                const uint32_t p_me = sel4(sub, q0.x, q0.y, q0.z, q0.w);
                const uint64_t t_me = (static_cast<uint64_t>(sel4(sub, q1.y, q1.w, q2.y, q2.w)) << 32) | sel4(sub, q1.x, q1.z, q2.x, q2.z);
                uint32_t alive_q = (sub < rows_q && ((state_codes ^ t_me) & vm64) == 0ull && p_me >= n_v) ? 1u << sub : 0u;
                alive_q |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(alive_q), 0xB1, 0xF, 0xF, true));
                alive_q |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(alive_q), 0x4E, 0xF, 0xF, true));
                if (writer) {
                    const uint32_t n_alive = static_cast<uint32_t>(__popc(alive_q));
                    uint4 rec;
                    if (n_alive == 0u) {
                        rec = make_uint4(0u, 0u, 0xffffffffu, 0u);
                    } else if (n_alive <= 2u) {  // resolved records of one or two positions, in the order of their rows
                        const uint32_t f = static_cast<uint32_t>(__builtin_ctz(alive_q));
                        const uint32_t h1 = sel4(f, q0.x, q0.y, q0.z, q0.w) - n_v;
                        if (n_alive == 1u) {
                            rec = make_uint4(0u, 1u, h1, kRecResolved);
                        } else {
                            const uint32_t g = static_cast<uint32_t>(__builtin_ctz(alive_q & (alive_q - 1u)));
                            const uint32_t h2 = sel4(g, q0.x, q0.y, q0.z, q0.w) - n_v;
                            rec = make_uint4(h2, h2 + 2u, h1, kRecResolved);
                        }
                    } else {  // three or four: the masked record of the rows (locate reads their suffix-array line)
                        rec = make_uint4(q3.x, q3.x + n_alive, alive_q, (n_v & 0x1fffffu) | kRecMasked);
                    }
                    if (out_rec) out_rec[q] = rec;
                    if (out_count) out_count[q] = rec.y - rec.x;
                    if (out_status) out_status[q] = 0;
                }
                continue;
            }
            if (from_state && (st.y & kStatePair) != 0u) {
                // a two-copy repeat (kStatePair): both positions and the 32 symbols in front of each in ONE 32-byte record -- no
                // suffix-array line, no text lines; the record's contexts are whole, so the compare is all there is to decide
                const u32x4 *pr = vv.seed_pairs + 2ull * st.x;
                const u32x4 r0 = pr[0], r1 = pr[1];
                const uint32_t n_v = st.y & 0x1fffffu;  // 1 .. 32 symbols in front of the seed
                const uint64_t vm64 = n_v == 32u ? ~0ull : ~0ull << (2u * (32u - n_v));
                const uint64_t t1 = (static_cast<uint64_t>(r0.w) << 32) | r0.z, t2 = (static_cast<uint64_t>(r1.y) << 32) | r1.x;
                const bool ok1 = ((state_codes ^ t1) & vm64) == 0ull && r0.x >= n_v, ok2 = ((state_codes ^ t2) & vm64) == 0ull && r0.y >= n_v;
                if (writer) {
                    const uint32_t h1 = r0.x - n_v, h2 = r0.y - n_v;  // (rows lo, lo + 1: the order of the reference's hits)
                    uint4 rec;
                    if (ok1 && ok2) rec = make_uint4(h2, h2 + 2u, h1, kRecResolved);
                    else if (ok1 || ok2) rec = make_uint4(0u, 1u, ok1 ? h1 : h2, kRecResolved);
                    else rec = make_uint4(0u, 0u, 0xffffffffu, 0u);
                    if (out_rec) out_rec[q] = rec;
                    if (out_count) out_count[q] = rec.y - rec.x;
                    if (out_status) out_status[q] = 0;
                }
                continue;
            }
            if (from_state) {
                resumed = true;
                lo = st.x;
                hi = lo + (st.y >> 24);
                rem = st.y & 0x1fffffu;
            } else if (kSeed && !bail && state != nullptr) {
                // the seed kernel found the k-mer on several rows: {lo, hi, symbols, 1}, or one of the packed forms
                const bool plain = state_packed == 0u && st.w == 1u;
                const bool p_rows = state_packed != 0u && (st.y & kStatePacked) != 0u, p_wide = state_packed != 0u && (st.y & kStatePlain) != 0u;
                if (plain || p_rows || p_wide) {
                    resumed = true;
                    rem = static_cast<uint32_t>(len);
                    w = fast_window<kXlate>(vv, s_dense, wbase, off0, rem, sub);
                    lo = st.x;
                    hi = plain ? st.y : (p_rows ? lo + (st.y >> 24) : st.z);
                    rem = plain ? st.z : st.y & 0x1fffffu;
                    shift = vv.seed_k >> 3;
                    part = vv.seed_k & 7u;
                }
            }
            if (kSeed && !bail && !resumed) {
                rem = static_cast<uint32_t>(len);
                w = fast_window<kXlate>(vv, s_dense, wbase, off0, rem, sub);
                const uint32_t k = vv.seed_k;
                // the symbols this step decides on: the k of the seed and up to 32 in front of them
                const uint32_t n_look = rem < k + 32u ? rem : k + 32u;
                const uint32_t n_words = n_look <= w.s0 ? 1u : 1u + ((n_look - w.s0 + 7u) >> 3);
                if ((w.valid8 & ((1u << n_words) - 1u)) != (1u << n_words) - 1u) {
                    bail = true;
                } else {
                    // the window as one string, the query's last symbol in the top bits of w0
                    const uint32_t w0 = __builtin_amdgcn_alignbit(w.l0, w.l0, 16), w1 = __builtin_amdgcn_alignbit(w.l1, w.l1, 16);
                    const uint32_t w2 = __builtin_amdgcn_alignbit(w.l2, w.l2, 16), w3 = __builtin_amdgcn_alignbit(w.l3, w.l3, 16);
                    const uint64_t key = ((static_cast<uint64_t>(w0) << 32) | w1) >> (64u - 2u * k);
                    uint32_t tag;
                    uint32_t b = seed_home(key, vv.seed_tag_bits, vv.seed_buckets, tag);
                    uint32_t ex, ey, ez, ew;
                    for (uint32_t d = 0;; d++) {
                        const u32x4 *bp = vv.seed + (static_cast<uint64_t>(b) << 3) + 2u * sub;
                        const u32x4 e0 = bp[0], e1 = bp[1];
                        const uint32_t want = tag | (d << kSeedDispShift);
                        const bool m0 = (e0.x & kSeedMatchMask) == want, m1 = (e1.x & kSeedMatchMask) == want;
                        const u32x4 es = m0 ? e0 : e1;
                        const bool m = m0 || m1;
                        ex = m ? (es.x | kSeedFound) : (e0.x & kSeedOverflow);
                        ey = m ? es.y : 0u;
                        ez = m ? es.z : 0u;
                        ew = m ? es.w : 0u;
                        ex |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ex), 0xB1, 0xF, 0xF, true));
                        ey |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ey), 0xB1, 0xF, 0xF, true));
                        ez |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ez), 0xB1, 0xF, 0xF, true));
                        ew |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ew), 0xB1, 0xF, 0xF, true));
                        ex |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ex), 0x4E, 0xF, 0xF, true));
                        ey |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ey), 0x4E, 0xF, 0xF, true));
                        ez |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ez), 0x4E, 0xF, 0xF, true));
                        ew |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ew), 0x4E, 0xF, 0xF, true));
                        // found, or the bucket never turned an entry away: the k-mer is not in the text
                        if ((ex & kSeedFound) != 0u || (ex & kSeedOverflow) == 0u || d >= kSeedMaxDisp) break;
                        b = b + 1u == vv.seed_buckets ? 0u : b + 1u;
                    }
                    rem -= k;
                    shift = k >> 3;
                    part = k & 7u;
                    if ((ex & kSeedFound) == 0u) {
                        lo = hi = 0u;  // count 0
                    } else if ((ex & kSeedKind) != 0u) {
                        lo = ey;
                        hi = ez;
                    } else {
                        single = true;
                        lo = 0u;
                        hi = 1u;
                        pos = ey;
                        // the 32 symbols that follow the seed in the window (in front of it in the query), in text order
                        uint32_t qh, ql;
                        const uint32_t sh = 32u - 2u * (k & 15u);
                        if (k == 16u) {
                            qh = w1;
                            ql = w2;
                        } else if (k < 16u) {
                            qh = __builtin_amdgcn_alignbit(w0, w1, sh);
                            ql = __builtin_amdgcn_alignbit(w1, w2, sh);
                        } else {
                            qh = __builtin_amdgcn_alignbit(w1, w2, sh);
                            ql = __builtin_amdgcn_alignbit(w2, w3, sh);
                        }
                        const uint32_t n_v = rem < 32u ? rem : 32u;
                        const uint64_t vm64 = n_v == 32u ? ~0ull : ~(~0ull >> (2u * n_v));
                        const uint64_t qcode = (static_cast<uint64_t>(qh) << 32) | ql;
                        const uint64_t tcode = (static_cast<uint64_t>(ew) << 32) | ez;
                        // (partial entry: only n_text < 32 symbols in front are text A C G T -- a read that needs more runs
                        // into a sentinel or an N there; one that needs fewer never looks at the bits that hold the number)
                        const uint32_t v_code = (ex >> kSeedPartialShift) & 3u;
                        const uint32_t n_text = (ex & kSeedPartial) == 0u ? 0xffffffffu : (v_code == 0u ? (ez & 63u) : 29u + v_code);
                        single_ok = ((qcode ^ tcode) & vm64) == 0ull && pos >= rem && rem <= n_text;
                    }
                }
            }
            if (!kSeed && !bail) {
                rem = static_cast<uint32_t>(len);
                w = fast_window<kXlate>(vv, s_dense, wbase, off0, rem, sub);
                const uint32_t need = (w.s0 == 8u ? 1u : 3u) | (depth > 8u ? (w.s0 == 8u ? 2u : 6u) : 0u);
                if ((w.valid8 & need) != need) {
                    bail = true;
                } else {
                    const uint2 e = vv.top[__builtin_amdgcn_alignbit(w.l0, w.l0, 16) >> (32u - 2u * depth)];
                    lo = e.x;
                    hi = e.y;
                    rem -= depth;
                    shift = depth >> 3;
                    part = depth & 7u;
                }
            }
            if (wide_to_list != 0u && !bail && rem > 0u && hi - lo > vv.max_rows) bail = true;  // (the general kernel's, with its state)
            // narrow with LF steps on the rank lines while the interval is wider than the rows a verify round takes
            while (!bail && rem > 0u && hi - lo > vv.max_rows) {
                if (shift > 6u) {
                    w = fast_window<kXlate>(vv, s_dense, wbase, off0, rem, sub);
                    shift = 0;
                    part = 0;
                }
                const uint32_t v8 = w.valid8 >> shift;
                if ((v8 & (part < w.s0 ? 1u : 2u)) == 0u) {  // the word this symbol comes from is not clean
                    bail = true;
                    break;
                }
                const uint32_t e = sel4(shift >> 1, w.l0, w.l1, w.l2, w.l3);
                const uint32_t lv = (shift & 1u) ? e >> 16 : e & 0xffffu;
                const uint32_t c = ((lv >> (14u - 2u * part)) & 3u) + 1u;
                uint32_t rlo, rhi;
                QuadLineTable::rank2(ix, c, lo, hi, rlo, rhi);
                lo = s_count[c] + rlo;  // lib.rs:273-275
                hi = s_count[c] + rhi;
                rem--;
                part++;
                shift += part >> 3;
                part &= 7u;
            }
            const uint32_t rows = hi - lo;
            uint32_t alive = 0;  // bit j = row lo + j matches
            uint32_t pair_first = 0, pair_second = 0;
            bool pair = false;  // exactly two rows match: their occurrences
            if (kSeed && single && !bail && (rem <= 32u || !single_ok)) {
                alive = single_ok ? 1u : 0u;  // decided by the entry alone
            } else if (!bail && rem > 0u && rows != 0u) {
                const bool real = sub < rows;
                const uint32_t row = real ? lo + sub : lo;
                // SA[row]: the full suffix array, the row's jump entry, or the locate walk (sampled_suffix_array.rs:110-138)
                if (kSeed && single) {
                    // (known from the entry, as are the 32 symbols in front of it: the text compare starts beyond them)
                } else if (vv.sa_full != nullptr) {
                    pos = vv.sa_full[row];
                } else if (vv.jump != nullptr) {
                    pos = static_cast<const uint32_t *>(vv.jump)[static_cast<uint64_t>(row) * 8u + 6u];
                } else {
                    uint32_t r_ = row, steps = 0;
                    for (;;) {
                        uint32_t slot_;
                        if (sampled_slot(ix, r_, slot_)) {
                            pos = vv.sa_samples[slot_] + steps;
                            break;
                        }
                        uint32_t rk;
                        const uint32_t c = LineTable::symbol_and_rank(ix, r_, rk);
                        if (c == 0) {
                            pos = vv.border_vals[lower_bound_u32(vv.border_keys, vv.n_texts, r_)] + steps;
                            break;
                        }
                        r_ = vv.count[c] + rk;
                        steps++;
                    }
                }
                bool ok = real && pos >= rem;  // the occurrence would start at pos - rem
                // compare the query's first `rem` symbols with the text in front of pos, 32 symbols per pass from the right
                uint32_t rem_v = rem;
                bool first = part == 0u && shift <= 3u;  // the window still serves the first pass
                if (kSeed && single) {
                    rem_v = rem - 32u;
                    first = false;
                }
                while (rem_v > 0u) {
                    const uint32_t n_v = rem_v < 32u ? rem_v : 32u;
                    uint64_t qcode;  // the 32 symbols [rem_v - 32, rem_v) in text order, the one next to the seed in the top bits
                    if (from_state) {
                        qcode = state_codes;  // (all of them: at most 32, clean -- the seed kernel listed the read with them)
                    } else {
                        if (!first) {
                            w = fast_window<kXlate>(vv, s_dense, wbase, off0, rem_v, sub);
                            shift = 0;
                        }
                        first = false;
                        // validity of the query words these symbols come from (group-uniform)
                        const uint32_t v8 = w.valid8 >> shift;
                        const uint32_t vl = v8 & (w.s0 == 8u ? 0xffu : (v8 >> 1));
                        const uint32_t n_full = n_v >> 3, n_tail = n_v & 7u;
                        bool clean = (vl & ((1u << n_full) - 1u)) == ((1u << n_full) - 1u);
                        if (n_tail != 0u) clean = clean && ((v8 >> n_full) & 1u) != 0u && (n_tail <= w.s0 || ((v8 >> (n_full + 1u)) & 1u) != 0u);
                        if (!clean) {
                            bail = true;
                            break;
                        }
                        uint32_t a0, a1;  // levels shift, shift + 1 | shift + 2, shift + 3 of the window
                        if (shift == 0u) {
                            a0 = w.l0;
                            a1 = w.l1;
                        } else if (shift == 1u) {
                            a0 = __builtin_amdgcn_alignbit(w.l1, w.l0, 16);
                            a1 = __builtin_amdgcn_alignbit(w.l2, w.l1, 16);
                        } else if (shift == 2u) {
                            a0 = w.l1;
                            a1 = w.l2;
                        } else {
                            a0 = __builtin_amdgcn_alignbit(w.l2, w.l1, 16);
                            a1 = __builtin_amdgcn_alignbit(w.l3, w.l2, 16);
                        }
                        // level shift + 3 lowest, level shift highest
                        qcode = (static_cast<uint64_t>(__builtin_amdgcn_alignbit(a0, a0, 16)) << 32) |
                                static_cast<uint64_t>(__builtin_amdgcn_alignbit(a1, a1, 16));
                    }
                    const uint64_t s0 = static_cast<uint64_t>(pos) - rem + rem_v + 32u * kTextPadUnits - 32u;  // (ok: pos >= rem)
                    const uint32_t b = static_cast<uint32_t>(s0 & 31u);
                    const u32x4 *tu = vv.text_units + (s0 >> 5);
                    const u32x4 u0 = ok ? tu[0] : u32x4{0u, 0u, 0u, 0u};
                    const u32x4 u1 = (ok && b != 0u) ? tu[1] : u32x4{0u, 0u, 0u, 0u};
                    const uint64_t c0 = static_cast<uint64_t>(u0.x) | (static_cast<uint64_t>(u0.y) << 32);
                    const uint64_t c1 = static_cast<uint64_t>(u1.x) | (static_cast<uint64_t>(u1.y) << 32);
                    const uint64_t tcode = b ? (c0 >> (2u * b)) | (c1 << (64u - 2u * b)) : c0;
                    const uint32_t tmask = b ? (u0.z >> b) | (u1.z << (32u - b)) : u0.z;
                    const uint64_t vm64 = n_v == 32u ? ~0ull : ~0ull << (2u * (32u - n_v));
                    const uint32_t vm32 = n_v == 32u ? ~0u : ~0u << (32u - n_v);
                    ok = ok && ((qcode ^ tcode) & vm64) == 0ull && (tmask & vm32) == 0u;
                    rem_v -= n_v;
                }
                alive = (ok && !bail) ? 1u << sub : 0u;
                alive |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(alive), 0xB1, 0xF, 0xF, true));
                alive |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(alive), 0x4E, 0xF, 0xF, true));
                // the four lanes' occurrences side by side: a read that ends on TWO rows takes both positions along in its record
                // (kernels.hpp: a resolved record of two) -- on a text of repeats that is most of the reads with more than one
                // hit, and locate then has no suffix-array line to fetch for them
                const int mine = static_cast<int>(pos - rem);
                const uint32_t o0 = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(mine, 0x00, 0xF, 0xF, true));
                const uint32_t o1 = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(mine, 0x55, 0xF, 0xF, true));
                const uint32_t o2 = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(mine, 0xAA, 0xF, 0xF, true));
                const uint32_t o3 = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(mine, 0xFF, 0xF, 0xF, true));
                if (__popc(alive) == 2) {
                    const uint32_t f = static_cast<uint32_t>(__builtin_ctz(alive)), g = static_cast<uint32_t>(__builtin_ctz(alive & (alive - 1u)));
                    pair_first = sel4(f, o0, o1, o2, o3);
                    pair_second = sel4(g, o0, o1, o2, o3);
                    pair = true;
                }
            }
            if (writer) {
                if (bail) {
                    s_left[atomicAdd(&s_nleft, 1u)] = q;
                    // (the general kernel resumes from {lo, hi, symbols, 1}: lo, hi, rem are still the seed's here)
                    if (kSeed && state_packed != 0u && resumed && out_rec) out_rec[q] = make_uint4(lo, hi, rem, 1u);
                } else {
                    uint4 rec;
                    if (kSeed && single) {
                        // (no row is known, and none is needed: a resolved record is its position)
                        rec = alive ? make_uint4(0u, 1u, pos - rem, kRecResolved) : make_uint4(0u, 0u, 0xffffffffu, 0u);
                    } else if (rem == 0u || rows == 0u) {
                        rec = make_uint4(lo, hi, 0xffffffffu, 0u);  // the interval itself (count = rows)
                    } else if (rows == 1u) {
                        // (lane 0 holds row lo: its SA value is this record's position)
                        rec = alive ? make_uint4(lo, lo + 1u, pos - rem, kRecResolved) : make_uint4(lo, lo, 0xffffffffu, 0u);
                    } else if (pair) {
                        rec = make_uint4(pair_second, pair_second + 2u, pair_first, kRecResolved);
                    } else {
                        rec = make_uint4(lo, lo + static_cast<uint32_t>(__popc(alive)), alive, (rem & 0x1fffffu) | kRecMasked);
                    }
                    if (out_rec) out_rec[q] = rec;
                    if (out_count) out_count[q] = rec.y - rec.x;
                    if (out_status) out_status[q] = 0;
                }
            }
        }
        __syncthreads();
        const uint32_t n_left = s_nleft;
        if (threadIdx.x == 0 && n_left != 0u) s_left_base = atomicAdd(n_leftover, n_left);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_left; i += kBlock) leftover[s_left_base + i] = s_left[i];
        __syncthreads();
        if (threadIdx.x == 0) s_nleft = 0;
    }
}

// ---- the seed table's own kernel ---------------------------------------------------------------------------------------
// What search_verify_kernel4<., true> does for the two cases that need nothing but the seed table and the text units --
// the k-mer is absent, or it occurs once -- as a software pipeline: PMC on that kernel (profiles/r03/experiments.md
// section 9) showed 404 VALU instructions per round of 16 reads (a third of them moving spilled scalars: the view has
// twenty pointers) at 33 G requests/s, below the gather ceiling, because a read is a chain of three dependent loads
// (offsets -> query bytes -> bucket) and a wavefront waited for each in turn.  Here the offsets of round i + 3 and the
// query bytes of round i + 2 are in flight while round i + 1 computes its key and round i looks at its bucket, the view
// is eleven words, and everything else -- a k-mer on several rows, a symbol outside A C G T, a read shorter than the
// seed -- is listed for search_verify_kernel4<., true> (which takes such a list), whose own leftovers go to the general
// kernel as before.
struct SeedView {
    const u32x4 *seed;
    const u32x4 *text_units;
    const uint32_t *isa;  // kExact: IndexView::isa
    const uint8_t *io_to_dense;
    uint32_t buckets, k, tag_bits;
    uint32_t perm_code_lo, perm_code_hi, perm_exp_lo, perm_exp_hi, perm_mask;  // IndexView::perm_*
    uint32_t n = 0;  // kCursor: IndexView::n (cursor_empty is [0, n))
};

// kExact: exact intervals (cursors_for_many_queries) instead of records: a read whose seed occurs once and whose other
// symbols agree with the text occurs once itself, and the row of its only suffix is ISA[its position] -- one more fetch
// (IndexView::isa), issued in one round and stored in the next.  Everything else needs the reference's frozen empty
// interval (or an interval wider than a row) and is listed for search_exact_kernel4.
// kCursor (with kExact; round 6): the FIRST chunk of a batch of cursors (gdx_cursor_extend_front_chunk_dev / _strings_dev).  A
// cursor that is still cursor_empty (lib.rs:202-210) and gets a chunk of seed_k .. seed_k + 32 symbols is a search of that
// chunk: the pipeline serves it as it serves a read -- bucket, entry, ISA -- and appends it to the live list; every other cursor
// (another state, a status, a chunk too short or too long, a k-mer that is absent or on several rows) is listed untouched for
// search_exact_kernel4, which is the statement of what a cursor call does.  Lock-step groups of that kernel paid for the
// slowest read of sixteen in every round (the absent k-mers' top table and pair steps); split by kind, both passes run dense.
template <int kXlate, bool kExact, bool kCursor = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void search_seed_kernel4(
    SeedView sv, const uint8_t *__restrict__ qbuf, const uint64_t *__restrict__ qbeg, const uint64_t *__restrict__ qend,
    uint64_t nq, uint32_t *__restrict__ out_count, uint8_t *__restrict__ out_status, uint4 *__restrict__ out_rec,
    uint32_t *__restrict__ out_start, uint32_t *__restrict__ out_end,
    uint32_t range, uint32_t *__restrict__ leftover, uint32_t *__restrict__ n_leftover,
    uint4 *__restrict__ state,  // != null: where a listed read stands, {lo, hi, symbols left, 1} after a seed entry that holds
                                // an interval, {0, 0, 0, 0} = from the beginning (search_fast_kernel4 goes on from there)
    uint32_t *__restrict__ out_compact,  // != null: compact results instead of records (kernels.hpp)
    // reads longer than the entry covers whose seed and 32 symbols in front agree with the text: listed for
    // seed_text_kernel4 with {position of the seed, symbols in front of it} in long_state[q * long_stride] (the first half
    // of their record slot, or an array of its own when the call has no records; exact mode: in out_start / out_end)
    uint32_t *__restrict__ long_list, uint32_t *__restrict__ n_long, uint2 *__restrict__ long_state, uint32_t long_stride,
    // state_packed != 0: a read listed with its seed entry's interval also carries the 32 symbols in front of the seed, as the
    // 2-bit codes this kernel holds anyway: {lo, rows << 24 | kStatePacked | symbols left, codes hi, codes lo} -- the fast
    // kernel then needs neither the read's offsets nor its bytes for up to 32 symbols (kStatePacked, search_fast_kernel4)
    uint32_t state_packed,
    uint32_t ulen,  // uniform batch (query_begin): the offsets stage of the pipeline computes instead of loading
    CursorArgs ca)  // kCursor: the cursors to look at (active_in), the live list (active_out), the chunk view
{
    static_assert(!kCursor || kExact, "cursor chunks are exact searches");
    constexpr uint32_t kGroup = 4, kGroups = kBlock / kGroup;
    constexpr uint32_t kNoQuery = 0xffffffffu;  // a pipeline slot beyond the range
    __shared__ uint8_t s_dense[256];
    __shared__ uint16_t s_left[kMaxRange], s_long[kMaxRange];  // slots of the range (its base is added when they are flushed)
    __shared__ uint32_t s_nleft, s_left_base, s_nlong, s_long_base;
    __shared__ uint32_t s_alive[kCursor ? kCursorRange : 1];  // flush_live_ordered (ranges of a cursor launch: <= kCursorRange)
    __shared__ uint32_t s_alive_part[kCursor ? kBlock : 1];
    __shared__ uint32_t s_alive_base;
    if (kXlate == 0)
        for (int i = threadIdx.x; i < 256; i += kBlock) s_dense[i] = sv.io_to_dense[i];
    if (kCursor)
        for (uint32_t i = threadIdx.x; i < kCursorRange; i += kBlock) s_alive[i] = kDeadCursor;
    if (threadIdx.x == 0) s_nleft = s_nlong = 0;
    __syncthreads();
    const uint32_t *active = kCursor ? ca.active_in : nullptr;
    if (kCursor && ca.n_active_in != nullptr) nq = *ca.n_active_in;
    const bool writer = (threadIdx.x % kGroup) == 0;
    const uint32_t sub = threadIdx.x & (kGroup - 1u);
    const uint32_t slot0 = threadIdx.x / kGroup;
    const uint32_t k = sv.k;
    const uint64_t n_ranges = (nq + range - 1) / range;
    for (uint64_t rg = blockIdx.x; rg < n_ranges; rg += gridDim.x) {
        const uint64_t base = rg * range;
        const uint32_t cnt = nq - base < range ? static_cast<uint32_t>(nq - base) : range;
        const int n_it = static_cast<int>((cnt + kGroups - 1u) / kGroups);
        // pipeline state: stage C = offsets loaded, B = query bytes loaded, A = bucket loaded
        uint64_t c_beg = 0, c_end = 0;
        bool c_on = false;
        u32x4 b_raw = {0u, 0u, 0u, 0u};
        uint64_t b_beg = 0;
        uint32_t b_len = kNoQuery;  // kNoQuery: nothing in this stage
        u32x4 a_e0 = {0u, 0u, 0u, 0u}, a_e1 = {0u, 0u, 0u, 0u};
        uint32_t a_tag = 0, a_bucket = 0, a_qh = 0, a_ql = 0, a_rem = kNoQuery;  // a_rem: symbols in front of the seed
        bool a_left = false;  // the read of stage A goes to the leftover list
        uint32_t z_row = 0, z_q = 0;  // kExact: the row of read z_q is on its way
        bool z_on = false;
        // kCursor: the cursor of each stage (a list entry, not base + slot), whether its string goes on left of this chunk
        uint32_t c_q = 0, b_q = 0, a_q = 0, z_slot = 0;
        bool c_more = false, c_ok = true, b_more = false, a_more = false, z_more = false;
        for (int it = -3; it <= n_it; it++) {
            if (kExact) {
                if (z_on && writer) {
                    out_start[z_q] = z_row;
                    out_end[z_q] = z_row + 1u;
                    if (out_count) out_count[z_q] = 1u;
                    if (!kCursor && out_status) out_status[z_q] = 0;
                    if (kCursor && ca.active_out != nullptr && z_more) s_alive[z_slot] = z_q;
                }
                z_on = false;
            }
            if (it == n_it) break;
            // ---- stage A -> result: round `it` looks at its bucket -------------------------------------------------
            if (it >= 0) {
                const uint32_t slot = slot0 + static_cast<uint32_t>(it) * kGroups;
                const uint32_t q = kCursor ? a_q : static_cast<uint32_t>(base + slot);
                if (a_left) {
                    if (writer) {
                        s_left[atomicAdd(&s_nleft, 1u)] = static_cast<uint16_t>(slot);
                        if (state) state[q] = make_uint4(0u, 0u, 0u, 0u);
                        if (!kExact && out_compact) out_compact[q] = kCompactSee;
                    }
                } else if (a_rem != kNoQuery) {
                    uint32_t ex, ey, ez, ew;
                    u32x4 e0 = a_e0, e1 = a_e1;
                    uint32_t b = a_bucket;
                    for (uint32_t d = 0;; d++) {
                        const uint32_t want = a_tag | (d << kSeedDispShift);
                        const bool m0 = (e0.x & kSeedMatchMask) == want, m1 = (e1.x & kSeedMatchMask) == want;
                        const u32x4 es = m0 ? e0 : e1;
                        const bool m = m0 || m1;
                        ex = m ? (es.x | kSeedFound) : (e0.x & kSeedOverflow);
                        ey = m ? es.y : 0u;
                        ez = m ? es.z : 0u;
                        ew = m ? es.w : 0u;
                        ex |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ex), 0xB1, 0xF, 0xF, true));
                        ey |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ey), 0xB1, 0xF, 0xF, true));
                        ez |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ez), 0xB1, 0xF, 0xF, true));
                        ew |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ew), 0xB1, 0xF, 0xF, true));
                        ex |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ex), 0x4E, 0xF, 0xF, true));
                        ey |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ey), 0x4E, 0xF, 0xF, true));
                        ez |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ez), 0x4E, 0xF, 0xF, true));
                        ew |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(ew), 0x4E, 0xF, 0xF, true));
                        if ((ex & kSeedFound) != 0u || (ex & kSeedOverflow) == 0u || d >= kSeedMaxDisp) break;
                        b = b + 1u == sv.buckets ? 0u : b + 1u;  // the bucket turned an entry away: look into the next one
                        const u32x4 *bp = sv.seed + (static_cast<uint64_t>(b) << 3) + 2u * sub;
                        e0 = bp[0];
                        e1 = bp[1];
                    }
                    if ((ex & kSeedFound) != 0u && (ex & kSeedKind) != 0u) {
                        if (kExact && a_rem == 0u) {  // the read IS the k-mer: the entry holds its interval
                            if (writer) {
                                out_start[q] = ey;
                                out_end[q] = ez;
                                if (out_count) out_count[q] = ez - ey;
                                if (!kCursor && out_status) out_status[q] = 0;
                                if (kCursor && ca.active_out != nullptr && a_more && ey != ez) s_alive[slot] = q;
                            }
                        } else if (writer) {  // several rows: the next kernel takes it from this interval
                            s_left[atomicAdd(&s_nleft, 1u)] = static_cast<uint16_t>(slot);
                            if (!kExact && state) {
                                if (state_packed == 0u) state[q] = make_uint4(ey, ez, a_rem, 1u);
                                else if (state_packed == 2u && (ex & (kSeedPairInfo | kSeedQuadInfo)) != 0u && a_rem - 1u < 32u)
                                    state[q] = make_uint4(ew, ((ez - ey) << 24) | kStatePacked | kStatePair | a_rem, a_qh, a_ql);
                                else if (ez - ey < 256u) state[q] = make_uint4(ey, ((ez - ey) << 24) | kStatePacked | a_rem, a_qh, a_ql);
                                else state[q] = make_uint4(ey, kStatePlain | a_rem, ez, 0u);
                            }
                            if (!kExact && out_compact) out_compact[q] = kCompactSee;
                        }
                    } else {
                        bool hit = false, left = false, is_long = false;
                        const uint32_t rem = a_rem, pos = ey;
                        if ((ex & kSeedFound) != 0u) {
                            const uint32_t n_v = rem < 32u ? rem : 32u;
                            const uint64_t vm64 = n_v == 32u ? ~0ull : ~(~0ull >> (2u * n_v));
                            const uint64_t qcode = (static_cast<uint64_t>(a_qh) << 32) | a_ql;
                            const uint64_t tcode = (static_cast<uint64_t>(ew) << 32) | ez;
                            const uint32_t v_code = (ex >> kSeedPartialShift) & 3u;
                            const uint32_t n_text = (ex & kSeedPartial) == 0u ? 0xffffffffu : (v_code == 0u ? (ez & 63u) : 29u + v_code);
                            hit = ((qcode ^ tcode) & vm64) == 0ull && pos >= rem && rem <= n_text;
                            is_long = hit && rem > 32u;  // the rest against the text units: seed_text_kernel4
                        }
                        if (kCursor && is_long) {  // (a chunk longer than an entry covers: the exact kernel's text route)
                            if (writer) s_left[atomicAdd(&s_nleft, 1u)] = static_cast<uint16_t>(slot);
                        } else if (is_long) {
                            if (writer) {
                                s_long[atomicAdd(&s_nlong, 1u)] = static_cast<uint16_t>(slot);
                                if (kExact) {
                                    out_start[q] = pos;
                                    out_end[q] = rem;
                                } else {
                                    long_state[static_cast<uint64_t>(q) * long_stride] = make_uint2(pos, rem);
                                    if (out_compact) out_compact[q] = kCompactSee;  // (seed_text_kernel4 writes the result)
                                }
                            }
                        } else if (kExact) {
                            if (left || !hit) {  // no occurrence: the reference's frozen empty interval is the exact kernel's to find
                                if (writer) s_left[atomicAdd(&s_nleft, 1u)] = static_cast<uint16_t>(slot);
                            } else {
                                z_row = sv.isa[pos - rem];
                                z_q = q;
                                z_on = true;
                                z_slot = slot;
                                z_more = a_more;
                            }
                        } else if (writer) {
                            if (left) {
                                s_left[atomicAdd(&s_nleft, 1u)] = static_cast<uint16_t>(slot);
                                if (state) state[q] = make_uint4(0u, 0u, 0u, 0u);
                                if (out_compact) out_compact[q] = kCompactSee;
                            } else {
                                // (no row is known, and none is needed: a resolved record is its position)
                                if (out_compact) out_compact[q] = hit ? pos - rem : kCompactNone;
                                else if (out_rec) out_rec[q] = hit ? make_uint4(0u, 1u, pos - rem, kRecResolved) : make_uint4(0u, 0u, 0xffffffffu, 0u);
                                if (out_count) out_count[q] = hit ? 1u : 0u;
                                if (out_status) out_status[q] = 0;
                            }
                        }
                    }
                }
            }
            // ---- stage B -> A: round `it + 1` turns its query bytes into key, bucket address and the 32 symbols in front
            a_left = false;
            a_rem = kNoQuery;
            a_q = b_q;
            a_more = b_more;
            if (b_len != kNoQuery) {
                if (b_len < k || b_len >= (1u << 21)) {
                    a_left = true;
                } else {
                    const uint32_t off0 = static_cast<uint32_t>(b_beg & 7u);
                    const FastWindow w = fast_window_finish<kXlate>(sv, s_dense, b_raw, off0, b_len, sub);
                    const uint32_t n_look = b_len < k + 32u ? b_len : k + 32u;
                    const uint32_t n_words = n_look <= w.s0 ? 1u : 1u + ((n_look - w.s0 + 7u) >> 3);
                    if ((w.valid8 & ((1u << n_words) - 1u)) != (1u << n_words) - 1u) {
                        a_left = true;
                    } else {
                        const uint32_t w0 = __builtin_amdgcn_alignbit(w.l0, w.l0, 16), w1 = __builtin_amdgcn_alignbit(w.l1, w.l1, 16);
                        const uint32_t w2 = __builtin_amdgcn_alignbit(w.l2, w.l2, 16), w3 = __builtin_amdgcn_alignbit(w.l3, w.l3, 16);
                        const uint64_t key = ((static_cast<uint64_t>(w0) << 32) | w1) >> (64u - 2u * k);
                        a_bucket = seed_home(key, sv.tag_bits, sv.buckets, a_tag);
                        const u32x4 *bp = sv.seed + (static_cast<uint64_t>(a_bucket) << 3) + 2u * sub;
                        a_e0 = bp[0];
                        a_e1 = bp[1];
                        const uint32_t sh = 32u - 2u * (k & 15u);
                        if (k == 16u) {
                            a_qh = w1;
                            a_ql = w2;
                        } else if (k < 16u) {
                            a_qh = __builtin_amdgcn_alignbit(w0, w1, sh);
                            a_ql = __builtin_amdgcn_alignbit(w1, w2, sh);
                        } else {
                            a_qh = __builtin_amdgcn_alignbit(w1, w2, sh);
                            a_ql = __builtin_amdgcn_alignbit(w2, w3, sh);
                        }
                        a_rem = b_len - k;
                    }
                }
            }
            // ---- stage C -> B: round `it + 2` knows where its query is and asks for its last 64 bytes ----------------
            b_len = kNoQuery;
            b_q = c_q;
            b_more = c_more;
            if (c_on) {
                const uint64_t len = c_end - c_beg;
                b_beg = c_beg;
                b_len = len < (1ull << 21) ? static_cast<uint32_t>(len) : (1u << 21);
                if (kCursor && !c_ok) b_len = 0u;  // not cursor_empty, or stopped: listed untouched
                if (b_len >= k && b_len < (1u << 21))
                    b_raw = fast_window_load<kXlate>(query_words<kXlate>(qbuf, c_beg), static_cast<uint32_t>(c_beg & 7u), b_len, sub);
            }
            // ---- -> stage C: the offsets of round `it + 3` -------------------------------------------------------------
            c_on = false;
            if (it + 3 < n_it) {
                const uint32_t slot = slot0 + static_cast<uint32_t>(it + 3) * kGroups;
                if (slot < cnt) {
                    if (kCursor) {
                        const uint32_t q = active != nullptr ? active[base + slot] : static_cast<uint32_t>(base + slot);
                        uint64_t cb = qbeg[q], ce = qend[q];
                        const uint32_t clo = out_start[q], chi = out_end[q];
                        const uint32_t cst = out_status != nullptr ? out_status[q] : 0u;
                        bool more = true;  // chunk view: the query has symbols left of this chunk (search_exact_kernel4)
                        if (ca.chunk_symbols != 0u) {
                            const uint64_t first = cb, skip = static_cast<uint64_t>(ca.chunk_index) * ca.chunk_symbols;
                            ce = ce - first > skip ? ce - skip : first;
                            cb = ce - first > ca.chunk_symbols ? ce - ca.chunk_symbols : first;
                            more = cb > first;
                        }
                        c_beg = cb;
                        c_end = ce;
                        c_q = q;
                        c_more = more;
                        c_ok = clo == 0u && chi == sv.n && cst == 0u;
                    } else {
                        const uint32_t q = static_cast<uint32_t>(base + slot);
                        c_beg = query_begin(qbeg, ulen, q);
                        c_end = query_end(qend, ulen, q);
                    }
                    c_on = true;
                }
            }
        }
        // flush the range's leftover queries: one atomic, coalesced stores
        __syncthreads();
        const uint32_t n_left = s_nleft;
        if (threadIdx.x == 0 && n_left != 0u) s_left_base = atomicAdd(n_leftover, n_left);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_left; i += kBlock)
            leftover[s_left_base + i] = (kCursor && active != nullptr) ? active[base + s_left[i]] : static_cast<uint32_t>(base) + s_left[i];
        const uint32_t n_lng = s_nlong;
        if (threadIdx.x == 0 && n_lng != 0u) s_long_base = atomicAdd(n_long, n_lng);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_lng; i += kBlock) long_list[s_long_base + i] = static_cast<uint32_t>(base) + s_long[i];
        __syncthreads();
        if (threadIdx.x == 0) s_nleft = s_nlong = 0;
        if (kCursor && ca.active_out != nullptr) flush_live_ordered(s_alive, cnt, s_alive_part, &s_alive_base, ca.active_out, ca.n_active_out);
    }
}

// ---- the seed table, one LANE per read ---------------------------------------------------------------------------------
// search_seed_kernel4 gives every read four lanes, and everything that is the same for the four -- offsets, translation,
// key, hash, the compare with the entry, the result -- is computed four times over: 250 VALU instructions per round of 16
// reads, 65 % of the SIMDs' issue slots at 3.5-4.1 ms per 100 M reads, and occupancy beyond seven waves bought nothing
// (profiles/r04/README.md): the kernel had become instruction-bound at two thirds of the gather ceiling.  Here a wavefront
// takes 64 reads at a time and splits the work by what it is:
//   S1  lane = read: its last 56 symbols (packed: a 112-bit field of the buffer, five dwords; ASCII: 64 bytes through the
//       v_perm tables) -> key, bucket, tag, the 32 symbols in front of the seed; {bucket, tag} goes to LDS;
//   G   four rounds, lane group g of round r fetches the bucket of read 16 r + g (4 lanes x 32 bytes = the 128-byte line, the
//       access shape the DRAM likes: all 64 lines of the wavefront in flight at once) and the lane that holds the matching
//       entry puts it into the read's LDS slot;
//   S2  lane = read: the entry against the read's own symbols -> count, position, compact result; coalesced stores.
// The scalar work is done once per read, the 128-byte fetches keep their four-lane shape.
// A read whose k-mer is not in its home bucket while that bucket has turned entries away (5.6 % of the reads at 70 % load)
// must look into the next bucket: such reads are parked in a queue of the wavefront (LDS: bucket, tag | displacement and
// what S2 needs) and, whenever 64 of them have gathered, take a G pass of their own -- dense like the others, instead of
// a dependent second fetch that the other 63 lanes wait for.  The queue lives across the ranges of a block and is drained
// at the end of its last range.  Results, lists and states are those of search_seed_kernel4 (a k-mer on several rows, a
// read shorter than the seed, a symbol outside A C G T: listed for the next kernel); the lists are collected in LDS and
// flushed with one atomic per block and list (a text of repeats lists a third of its reads).  Count / locate searches only.
constexpr uint32_t kLaneRange = 2048;  // reads per block and range of search_seed_lane_kernel (its lists: 8 KB of LDS)
static_assert(kLaneRange <= kSumTile && kLaneRange % 64 == 0, "a range spans at most two scan tiles, a chunk of 64 lies in one");
template <int kXlate, bool kUniform>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kXlate == 2 ? 5 : 4))) void search_seed_lane_kernel(
    SeedView sv, const uint8_t *__restrict__ qbuf, const uint64_t *__restrict__ qbeg, const uint64_t *__restrict__ qend,
    uint64_t nq, uint32_t *__restrict__ out_count, uint8_t *__restrict__ out_status, uint4 *__restrict__ out_rec,
    uint32_t range, uint32_t *__restrict__ leftover, uint32_t *__restrict__ n_leftover, uint4 *__restrict__ state,
    uint32_t *__restrict__ out_compact, uint32_t *__restrict__ long_list, uint32_t *__restrict__ n_long,
    uint2 *__restrict__ long_state, uint32_t long_stride, uint32_t state_packed, uint32_t ulen,
    unsigned long long *__restrict__ tile_sums)  // != null: += the hits this kernel answers itself, per kSumTile queries
{
    static_assert(kXlate == 1 || kXlate == 2, "v_perm tables or packed queries");
    constexpr uint32_t kWaves = kBlock / 64;
    constexpr uint32_t kNoTag = 0xffffffffu;  // matches no entry (kSeedMatchMask leaves 26 bits)
    constexpr int kRaw = kXlate == 2 ? 5 : 15;  // dwords a lane loads for its read
    constexpr uint32_t kQueue = 128;            // parked reads per wavefront: < 64 before a chunk adds up to 64
    __shared__ uint2 s_bt[kWaves][64];
    __shared__ u32x4 s_e[kWaves][64];
    __shared__ uint2 s_qc[kWaves][64];  // a read's 32 symbols in front of the seed while its bucket is on the way (two registers
                                        // less across the loads: the packed variant fits five waves per SIMD without spilling)
    __shared__ uint32_t s_pq[kWaves][kQueue], s_pb[kWaves][kQueue], s_pt[kWaves][kQueue], s_pr[kWaves][kQueue];
    __shared__ uint2 s_pc[kWaves][kQueue];
    // the block's lists of a range (slots of the range; a read that comes out of the parked queue may belong to an earlier
    // range and is listed by its number), flushed with ONE atomic per list and range: on a text of repeats a third of the
    // reads is listed, and an atomic per wavefront on the one counter of a list cost 18 of 22 ms there
    constexpr uint32_t kLate = 128;
    __shared__ uint16_t s_left[kLaneRange], s_long[kLaneRange];
    __shared__ uint32_t s_late[2][kLate];
    __shared__ uint32_t s_n[4], s_base[4];  // counts / global bases of: left, long, late left, late long
    // tile_sums: the hits of the range's (at most two) scan tiles are summed here and leave with the lists -- an atomic per
    // chunk and per parked read on the tile counters in global memory was a twentieth of the kernel's memory requests
    __shared__ uint32_t s_tile_hits[2];
    if (threadIdx.x < 4) s_n[threadIdx.x] = 0;
    if (threadIdx.x < 2) s_tile_hits[threadIdx.x] = 0;
    __syncthreads();
    uint32_t tile0 = 0;  // the scan tile of the current range's first read
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, sub = lane & 3u, grp = lane >> 2;
    constexpr uint32_t kWin = 56;  // the window a lane loads: the last 56 symbols of its read (k <= 24: k + 32 <= 56)
    const uint32_t k = sv.k, span = k + 32u;  // symbols of a read this kernel looks at: the last `span` of the window
    const uint32_t drop2 = 2u * (kWin - span);  // bits of the window's code string in front of them (0 .. 32)
    const uint32_t *qw = reinterpret_cast<const uint32_t *>(qbuf);
    uint32_t n_parked = 0;  // wave-uniform

    // G: every lane has put {bucket, tag | displacement} (or kNoTag) of its read into s_bt; returns the read's entry with
    // kSeedFound set, or {overflow bit of the bucket, 0, 0, 0}
    auto gather = [&]() __attribute__((always_inline)) -> u32x4 {
        __builtin_amdgcn_wave_barrier();
        u32x4 e0[4], e1[4];
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) {
            const uint32_t bk = s_bt[wave][16u * r + grp].x;
            const u32x4 *bp = sv.seed + (static_cast<uint64_t>(bk) << 3) + 2u * sub;
            e0[r] = bp[0];
            e1[r] = bp[1];
        }
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) {
            const uint32_t want = s_bt[wave][16u * r + grp].y;  // (read again: four registers less across the loads)
            const bool m0 = (e0[r].x & kSeedMatchMask) == want, m1 = (e1[r].x & kSeedMatchMask) == want;
            // no entry matches: the slot says whether the bucket ever turned one away (bit 31 of every entry)
            if (sub == 0u) s_e[wave][16u * r + grp] = u32x4{e0[r].x & kSeedOverflow, 0u, 0u, 0u};
            if (m0 || m1) {
                u32x4 es = m0 ? e0[r] : e1[r];
                es.x |= kSeedFound;
                s_e[wave][16u * r + grp] = es;
            }
        }
        __builtin_amdgcn_wave_barrier();
        return s_e[wave][lane];
    };
    // S2 of a read whose entry (or absence) is known: 0 = answered, no occurrence; 1 = answered, one hit stored; kListLeft /
    // kListLong = to be listed for the next kernel / for seed_text_kernel4 (its state is written here, list_slot / list_late list it)
    constexpr uint32_t kListLeft = 2, kListLong = 3;
    auto finish = [&](uint32_t q, uint32_t rem, uint64_t qcode, const u32x4 &en) __attribute__((always_inline)) -> uint32_t {
        const uint32_t ex = en.x, ey = en.y, ez = en.z, ew = en.w;
        const bool found = (ex & kSeedFound) != 0u;
        if (found && (ex & kSeedKind) != 0u) {
            // several rows: the next kernel takes it from this interval
            if (state) {
                if (state_packed == 0u) state[q] = make_uint4(ey, ez, rem, 1u);
                else if (state_packed == 2u && (ex & (kSeedPairInfo | kSeedQuadInfo)) != 0u && rem - 1u < 32u)  // (kStatePair: ew = the record's index)
                    state[q] = make_uint4(ew, ((ez - ey) << 24) | kStatePacked | kStatePair | rem, static_cast<uint32_t>(qcode >> 32),
                                          static_cast<uint32_t>(qcode));
                else if (ez - ey < 256u)
                    state[q] = make_uint4(ey, ((ez - ey) << 24) | kStatePacked | rem, static_cast<uint32_t>(qcode >> 32),
                                          static_cast<uint32_t>(qcode));
                else state[q] = make_uint4(ey, kStatePlain | rem, ez, 0u);
            }
            if (out_compact) out_compact[q] = kCompactSee;
            return kListLeft;
        }
        bool hit = false;
        const uint32_t pos = ey;
        if (found) {
            const uint32_t n_v = rem < 32u ? rem : 32u;
            const uint64_t vm64 = n_v == 32u ? ~0ull : ~(~0ull >> (2u * n_v));
            const uint64_t tcode = (static_cast<uint64_t>(ew) << 32) | ez;
            const uint32_t v_code = (ex >> kSeedPartialShift) & 3u;
            const uint32_t n_text = (ex & kSeedPartial) == 0u ? 0xffffffffu : (v_code == 0u ? (ez & 63u) : 29u + v_code);
            hit = ((qcode ^ tcode) & vm64) == 0ull && pos >= rem && rem <= n_text;
        }
        if (hit && rem > 32u) {  // the rest against the text units: seed_text_kernel4
            long_state[static_cast<uint64_t>(q) * long_stride] = make_uint2(pos, rem);
            if (out_compact) out_compact[q] = kCompactSee;  // (seed_text_kernel4 writes the result)
            return kListLong;
        }
        // (no row is known, and none is needed: a resolved record is its position)
        if (out_compact) out_compact[q] = hit ? pos - rem : kCompactNone;
        else if (out_rec) out_rec[q] = hit ? make_uint4(0u, 1u, pos - rem, kRecResolved) : make_uint4(0u, 0u, 0xffffffffu, 0u);
        if (out_count) out_count[q] = hit ? 1u : 0u;
        if (out_status) out_status[q] = 0;
        return hit ? 1u : 0u;
    };
    // lists a read of the current range (by its slot) or one that left the parked queue (by its number; if the block's
    // small list for those is full: straight into the global list)
    auto list_slot = [&](uint32_t kind, uint32_t slot) __attribute__((always_inline)) {
        // (called by every lane of the wavefront: ONE LDS atomic per list and call -- a batch of long reads lists most of
        // its reads for seed_text_kernel4, and 64 atomics on one LDS word serialise)
#pragma unroll
        for (uint32_t w = 0; w < 2; w++) {
            const bool mine = kind == kListLeft + w;
            const unsigned long long mask = __ballot(mine);
            if (mask == 0ull) continue;
            const int leader = __ffsll(static_cast<long long>(mask)) - 1;
            uint32_t first = 0;
            if (static_cast<int>(lane) == leader) first = atomicAdd(&s_n[w], static_cast<uint32_t>(__popcll(mask)));
            first = __shfl(first, leader);
            if (mine) {
                const uint32_t at = first + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
                if (w == 0) s_left[at] = static_cast<uint16_t>(slot);
                else s_long[at] = static_cast<uint16_t>(slot);
            }
        }
    };
    auto list_late = [&](uint32_t kind, uint32_t q) __attribute__((always_inline)) {
        if (kind == kListLeft) {
            const uint32_t at = atomicAdd(&s_n[2], 1u);
            if (at < kLate) s_late[0][at] = q;
            else leftover[atomicAdd(n_leftover, 1u)] = q;  // (the small list is full: rare, one atomic per read)
        } else if (kind == kListLong) {
            const uint32_t at = atomicAdd(&s_n[3], 1u);
            if (at < kLate) s_late[1][at] = q;
            else long_list[atomicAdd(n_long, 1u)] = q;
        }
    };
    // the block's lists into the global ones (all threads; between two barriers of the caller's)
    auto flush_lists = [&](uint64_t base) __attribute__((always_inline)) {
        __syncthreads();
        if (threadIdx.x == 0 || threadIdx.x == 2) {  // (left lists)
            uint32_t n = s_n[threadIdx.x];
            if (threadIdx.x == 2 && n > kLate) n = kLate;
            s_base[threadIdx.x] = n != 0u ? atomicAdd(n_leftover, n) : 0u;
        } else if (threadIdx.x == 1 || threadIdx.x == 3) {  // (long lists)
            uint32_t n = s_n[threadIdx.x];
            if (threadIdx.x == 3 && n > kLate) n = kLate;
            s_base[threadIdx.x] = n != 0u ? atomicAdd(n_long, n) : 0u;
        }
        __syncthreads();
        const uint32_t n0 = s_n[0], n1 = s_n[1], n2 = s_n[2] < kLate ? s_n[2] : kLate, n3 = s_n[3] < kLate ? s_n[3] : kLate;
        for (uint32_t i = threadIdx.x; i < n0; i += kBlock) leftover[s_base[0] + i] = static_cast<uint32_t>(base) + s_left[i];
        for (uint32_t i = threadIdx.x; i < n1; i += kBlock) long_list[s_base[1] + i] = static_cast<uint32_t>(base) + s_long[i];
        for (uint32_t i = threadIdx.x; i < n2; i += kBlock) leftover[s_base[2] + i] = s_late[0][i];
        for (uint32_t i = threadIdx.x; i < n3; i += kBlock) long_list[s_base[3] + i] = s_late[1][i];
        if (tile_sums != nullptr && threadIdx.x < 2 && s_tile_hits[threadIdx.x] != 0u)
            atomicAdd(&tile_sums[base / kSumTile + threadIdx.x], static_cast<unsigned long long>(s_tile_hits[threadIdx.x]));
        __syncthreads();
        if (threadIdx.x < 4) s_n[threadIdx.x] = 0;
        if (threadIdx.x < 2) s_tile_hits[threadIdx.x] = 0;
        __syncthreads();
    };
    // parks the reads of the lanes with `again` set: the next bucket, one more displacement
    auto park = [&](bool again, uint32_t q, uint32_t bucket, uint32_t tagd, uint32_t rem, uint64_t qcode) __attribute__((always_inline)) {
        const unsigned long long mask = __ballot(again);
        if (mask == 0ull) return;
        if (again) {
            const uint32_t at = n_parked + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
            s_pq[wave][at] = q;
            s_pb[wave][at] = bucket + 1u == sv.buckets ? 0u : bucket + 1u;
            s_pt[wave][at] = tagd + (1u << kSeedDispShift);
            s_pr[wave][at] = rem;
            s_pc[wave][at] = make_uint2(static_cast<uint32_t>(qcode), static_cast<uint32_t>(qcode >> 32));
        }
        n_parked += static_cast<uint32_t>(__popcll(mask));
    };
    // true: the k-mer is not in this bucket, but may sit further on (the bucket turned entries away)
    auto goes_on = [&](const u32x4 &en, uint32_t tagd) __attribute__((always_inline)) {
        return (en.x & kSeedFound) == 0u && (en.x & kSeedOverflow) != 0u && (tagd >> kSeedDispShift) < kSeedMaxDisp;
    };
    // one G pass over (up to) 64 parked reads
    auto parked_pass = [&]() __attribute__((always_inline)) {
        const uint32_t take = n_parked < 64u ? n_parked : 64u;
        const uint32_t first = n_parked - take;
        __builtin_amdgcn_wave_barrier();
        const bool mine = lane < take;
        uint32_t q = 0, bucket = 0, tagd = kNoTag, rem = 0;
        uint64_t qcode = 0;
        if (mine) {
            q = s_pq[wave][first + lane];
            bucket = s_pb[wave][first + lane];
            tagd = s_pt[wave][first + lane];
            rem = s_pr[wave][first + lane];
            const uint2 c = s_pc[wave][first + lane];
            qcode = (static_cast<uint64_t>(c.y) << 32) | c.x;
        }
        __builtin_amdgcn_wave_barrier();
        n_parked = first;
        s_bt[wave][lane] = make_uint2(bucket, tagd);
        const u32x4 en = gather();
        const bool again = mine && goes_on(en, tagd);
        const uint32_t kind = (mine && !again) ? finish(q, rem, qcode, en) : 0u;
        if (kind == 1u && tile_sums != nullptr) {
            const uint32_t t = q / kSumTile - tile0;  // (a read parked in an earlier range of the block: straight to its tile)
            if (t < 2u) atomicAdd(&s_tile_hits[t], 1u);
            else atomicAdd(&tile_sums[q / kSumTile], 1ull);
        }
        list_late(kind, q);
        park(again, q, bucket, tagd, rem, qcode);
    };

    const uint64_t n_ranges = (nq + range - 1) / range;
    for (uint64_t rg = blockIdx.x; rg < n_ranges; rg += gridDim.x) {
        const uint64_t base = rg * range;
        const uint32_t cnt = nq - base < range ? static_cast<uint32_t>(nq - base) : range;
        const uint32_t n_chunks = (cnt + 63u) >> 6;
        tile0 = static_cast<uint32_t>(base / kSumTile);
        // ---- where read `slot` of the range ends and how long it is; its raw dwords (prefetched one chunk ahead) ----
        uint64_t r_end = 0;
        uint32_t r_len = 0;
        bool r_on = false, r_load = false;
        uint32_t raw[kRaw];
#pragma unroll
        for (int i = 0; i < kRaw; i++) raw[i] = 0u;
        auto fetch = [&](uint32_t ch) __attribute__((always_inline)) {
            const uint32_t slot = ch * 64u + lane;
            r_on = ch < n_chunks && slot < cnt;
            r_load = false;
            if (!r_on) return;
            const uint64_t q = base + slot;
            uint64_t beg;
            if (kUniform) {
                beg = q * ulen;
                r_end = beg + ulen;
            } else {
                beg = qbeg[q];
                r_end = qend[q];
            }
            const uint64_t len = r_end - beg;
            r_len = len < (1ull << 21) ? static_cast<uint32_t>(len) : (1u << 21);
            // (a read that ends inside the first 56 symbols of the buffer would make the window start before the buffer)
            r_load = r_len >= k && r_len < (1u << 21) && r_end >= kWin;
            if (!r_load) return;
            if (kXlate == 2) {
                const uint64_t bit = 2ull * (r_end - kWin);
                const uint32_t *p = qw + (bit >> 5);
                const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(p);
                raw[0] = v.x, raw[1] = v.y, raw[2] = v.z, raw[3] = v.w;
                raw[4] = p[4];
            } else {
                // the window's 56 bytes start o = (r_end - 56) & 3 bytes into dword 0: dwords 0 .. 13, and dword 14 only when
                // o != 0 -- with o == 0 it would begin at r_end + 0 and end 4 bytes past it, beyond the 8-byte padding the
                // contract asks for when the buffer's last query ends on a multiple of 8 (gdx.h; a 16th dword is never used)
                const uint32_t *p1 = qw + ((r_end - kWin) >> 2);
                const u32x4_a4 *p = reinterpret_cast<const u32x4_a4 *>(p1);
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const u32x4_a4 v = p[i];
                    raw[4 * i] = v.x, raw[4 * i + 1] = v.y, raw[4 * i + 2] = v.z, raw[4 * i + 3] = v.w;
                }
                raw[12] = p1[12];
                raw[13] = p1[13];
                raw[14] = ((r_end - kWin) & 3u) != 0u ? p1[14] : 0u;
            }
        };
        fetch(wave);
        for (uint32_t ch = wave; ch < n_chunks; ch += kWaves) {
            const uint32_t q = static_cast<uint32_t>(base + ch * 64u + lane);
            const bool on = r_on;
            // ---- S1: the raw dwords -> the 2-bit codes of the window, f0 (its first 16 symbols, the first lowest) .. f3; the
            // last `span` of them are the 32 symbols in front of the seed (qcode, as a text unit) and the k-mer (key)
            bool left = on && !r_load;  // shorter than the seed, too long, at the very start of the buffer: the next kernel
            const uint32_t rem = r_len - k;
            uint32_t f0 = 0, f1 = 0, f2 = 0, f3 = 0;
            if (kXlate == 2) {
                const uint32_t sh = static_cast<uint32_t>(2ull * (r_end - kWin)) & 31u;
                f0 = __builtin_amdgcn_alignbit(raw[1], raw[0], sh);
                f1 = __builtin_amdgcn_alignbit(raw[2], raw[1], sh);
                f2 = __builtin_amdgcn_alignbit(raw[3], raw[2], sh);
                f3 = __builtin_amdgcn_alignbit(raw[4], raw[3], sh);
            } else {
                const uint32_t o = static_cast<uint32_t>(r_end - kWin) & 3u;
                // symbols that count: [lo, 56) of the window (the ones before are not looked at, or belong to the read in front)
                const uint32_t lo = kWin - (r_len < span ? r_len : span);
                uint32_t bad = 0, c[14];
#pragma unroll
                for (uint32_t i = 0; i < 14; i++) {
                    const uint32_t bytes = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], o);
                    uint32_t b = 0;
                    c[i] = fast_pack4(sv, bytes, b);
                    const uint32_t m = 4u * i + 4u <= lo ? 0u : (4u * i >= lo ? 0xffffffffu : 0xffffffffu << (8u * (lo - 4u * i)));
                    bad |= b & m;
                }
                f0 = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
                f1 = c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24);
                f2 = c[8] | (c[9] << 8) | (c[10] << 16) | (c[11] << 24);
                f3 = c[12] | (c[13] << 8);
                if (bad != 0u) left = on;  // a symbol outside A C G T among those looked at
            }
            const uint64_t wq = (static_cast<uint64_t>(f1) << 32) | f0, wk = (static_cast<uint64_t>(f3) << 32) | f2;
            const uint64_t qcode0 = drop2 == 0u ? wq : (wq >> drop2) | (wk << (64u - drop2));
            const uint64_t key = (wk >> drop2) & ((1ull << (2u * k)) - 1ull);
            uint32_t tag = 0;
            const uint32_t bucket = seed_home(key, sv.tag_bits, sv.buckets, tag);
            const bool look = on && !left;
            s_bt[wave][lane] = look ? make_uint2(bucket, tag) : make_uint2(0u, kNoTag);
            s_qc[wave][lane] = make_uint2(static_cast<uint32_t>(qcode0), static_cast<uint32_t>(qcode0 >> 32));
            // ---- the next chunk's offsets and raw dwords are on their way while this one looks at its buckets ----
            fetch(ch + kWaves);
            const u32x4 en = gather();
            uint32_t li = lane;
            asm volatile("" : "+v"(li));  // (the compiler must read the value back instead of keeping it in registers)
            const uint2 qc2 = s_qc[wave][li];
            const uint64_t qcode = (static_cast<uint64_t>(qc2.y) << 32) | qc2.x;
            // ---- S2: lane = read again
            if (left) {  // from the beginning, by the next kernel
                if (state) state[q] = make_uint4(0u, 0u, 0u, 0u);
                if (out_compact) out_compact[q] = kCompactSee;
            }
            const bool again = look && goes_on(en, tag);
            const uint32_t kind = left ? kListLeft : ((look && !again) ? finish(q, rem, qcode, en) : 0u);
            list_slot(kind, ch * 64u + lane);
            if (tile_sums != nullptr) {  // (a chunk of 64 lies inside one tile: ranges are multiples of 64, at most a tile long)
                const unsigned long long hm = __ballot(kind == 1u);
                if (lane == 0u && hm != 0ull)
                    atomicAdd(&s_tile_hits[static_cast<uint32_t>((base + ch * 64u) / kSumTile) - tile0], static_cast<uint32_t>(__popcll(hm)));
            }
            park(again, q, bucket, tag, rem, qcode);
            // (below 64 again before the next chunk adds up to 64: the queue holds 128)
            while (n_parked >= 64u) parked_pass();
        }
        // the block's last range (with a grid of one block per range: its only one): the queue is drained before the lists go
        if (rg + gridDim.x >= n_ranges)
            while (n_parked != 0u) parked_pass();
        flush_lists(base);
    }  // (the chains are short: the largest displacement of a table is in gdx_index_seed_info)
}

// The reads search_seed_kernel4 listed as "long": seed and the 32 symbols in front agree with the text at `pos`, `rem` symbols
// are in front of the seed in all.  The rest against the text units, every lane of the group its own 32 symbols: lane `sub`
// of round r takes the symbols [e - 32, e) of the query with e = rem - 32 - 32 (4 r + sub) (clipped at the query's start),
// reads them as bytes and compares their 2-bit codes with the text in front of the occurrence -- four independent (query,
// text) fetches in flight per read.  (Inside the seed kernel this was a loop of dependent fetches, one 32-symbol pass after
// the other, and most of its time on reads of 20..150 symbols; done there lane-parallel it cost the pipeline 32 spilled
// registers.)  Results as search_seed_kernel4 writes them; a symbol outside A C G T goes on the leftover list.
template <int kXlate, bool kExact>
__global__ __launch_bounds__(kBlock) void seed_text_kernel4(
    SeedView sv, const uint8_t *__restrict__ qbuf, const uint64_t *__restrict__ qbeg, const uint32_t *__restrict__ list,
    const uint32_t *__restrict__ n_list, const uint2 *__restrict__ long_state, uint32_t long_stride,
    uint32_t *__restrict__ out_count, uint8_t *__restrict__ out_status, uint4 *__restrict__ out_rec,
    uint32_t *__restrict__ out_start, uint32_t *__restrict__ out_end, uint32_t *__restrict__ out_compact,
    uint4 *__restrict__ state, uint32_t *__restrict__ leftover, uint32_t *__restrict__ n_leftover, uint32_t ulen)
{
    constexpr uint32_t kGroup = 4, kGroups = kBlock / kGroup;
    __shared__ uint8_t s_dense[256];
    if (kXlate == 0) {
        for (int i = threadIdx.x; i < 256; i += kBlock) s_dense[i] = sv.io_to_dense[i];
        __syncthreads();
    }
    const uint32_t sub = threadIdx.x & (kGroup - 1u);
    const bool writer = sub == 0u;
    const uint64_t n = *n_list;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kGroups + threadIdx.x / kGroup; i < n;
         i += static_cast<uint64_t>(gridDim.x) * kGroups) {
        const uint32_t q = list[i];
        uint32_t pos, rem;
        if (kExact) {
            pos = out_start[q];
            rem = out_end[q];
        } else {
            const uint2 st = long_state[static_cast<uint64_t>(q) * long_stride];
            pos = st.x;
            rem = st.y;
        }
        const uint64_t begin = query_begin(qbeg, ulen, q);
        const uint32_t rest = rem - 32u;  // (rem > 32: why the read is here)
        uint32_t bad_q = 0, bad_t = 0;    // a query symbol outside A C G T / a mismatch, in any lane
        for (uint32_t done = 0; done < rest; done += 128u) {
            const uint32_t skip = done + 32u * sub;
            if (skip < rest) {
                const uint32_t e = rest - skip, n_c = e < 32u ? e : 32u, qs = e - n_c;
                const uint64_t at = begin + qs;
                // the text [tp, tp + n_c) in front of the occurrence, tp = pos - rem + qs
                const uint64_t s0 = static_cast<uint64_t>(pos) - rem + qs + 32u * kTextPadUnits;
                const uint32_t tb = static_cast<uint32_t>(s0 & 31u);
                const u32x4 *tu = sv.text_units + (s0 >> 5);
                uint64_t qc = 0;  // the 2-bit codes of the symbols [qs, qs + n_c): symbol qs + i in bits 2 i + 1 : 2 i
                uint32_t inv = 0;
                u32x4 u0, u1;
                if (kXlate == 2) {
                    // packed: the codes are the buffer's bits from 2 * at on -- the 16-bit units that hold them (1 .. 5)
                    const uint16_t *up = reinterpret_cast<const uint16_t *>(qbuf) + (at >> 3);
                    const uint32_t sh = static_cast<uint32_t>(at & 7u) * 2u;
                    const uint32_t n_need = (static_cast<uint32_t>(at & 7u) + n_c + 7u) >> 3;
                    uint64_t lo64 = up[0];
                    uint32_t hi16 = 0;
                    if (n_need > 1u) lo64 |= static_cast<uint64_t>(up[1]) << 16;
                    if (n_need > 2u) lo64 |= static_cast<uint64_t>(up[2]) << 32;
                    if (n_need > 3u) lo64 |= static_cast<uint64_t>(up[3]) << 48;
                    if (n_need > 4u) hi16 = up[4];
                    u0 = tu[0];
                    u1 = tb != 0u ? tu[1] : u32x4{0u, 0u, 0u, 0u};
                    qc = sh != 0u ? (lo64 >> sh) | (static_cast<uint64_t>(hi16) << (64u - sh)) : lo64;
                } else {
                // query bytes [qs, qs + n_c) as aligned 8-byte words (only the words that hold one of them)
                const uint64_t *wp = reinterpret_cast<const uint64_t *>(qbuf) + (at >> 3);
                const uint32_t sh = static_cast<uint32_t>(at & 7u) * 8u;
                const uint32_t n_need = (static_cast<uint32_t>(at & 7u) + n_c + 7u) >> 3;  // 1 .. 5
                uint64_t w0 = wp[0], w1 = 0, w2 = 0, w3 = 0, w4 = 0;
                if (n_need > 1u) w1 = wp[1];
                if (n_need > 2u) w2 = wp[2];
                if (n_need > 3u) w3 = wp[3];
                if (n_need > 4u) w4 = wp[4];
                u0 = tu[0];
                u1 = tb != 0u ? tu[1] : u32x4{0u, 0u, 0u, 0u};
                if (sh != 0u) {
                    w0 = (w0 >> sh) | (w1 << (64u - sh));
                    w1 = (w1 >> sh) | (w2 << (64u - sh));
                    w2 = (w2 >> sh) | (w3 << (64u - sh));
                    w3 = (w3 >> sh) | (w4 << (64u - sh));
                }
                // 32 bytes -> 64 bits of codes (byte i in bits 2 i + 1 : 2 i); b[g] != 0: group g (bytes 4 g .. 4 g + 3) holds a byte
                // that is not A C G T (the v_perm translation says which byte, the LDS table only that there is one)
                uint32_t b[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
                const uint32_t wd[8] = {static_cast<uint32_t>(w0), static_cast<uint32_t>(w0 >> 32), static_cast<uint32_t>(w1),
                                        static_cast<uint32_t>(w1 >> 32), static_cast<uint32_t>(w2), static_cast<uint32_t>(w2 >> 32),
                                        static_cast<uint32_t>(w3), static_cast<uint32_t>(w3 >> 32)};
#pragma unroll
                for (uint32_t g = 0; g < 8; g++) {
                    const uint32_t code = kXlate == 1 ? fast_pack4(sv, wd[g], b[g]) : fast_pack4_lds(s_dense, wd[g], b[g]);
                    qc |= static_cast<uint64_t>(code) << (8u * g);
                }
                const uint32_t full = n_c >> 2, part = n_c & 3u;
#pragma unroll
                for (uint32_t g = 0; g < 8; g++) {
                    // (LDS path: a group is flagged as a whole; the bytes behind n_c are later symbols of the same read, whose
                    // own chunk flags them too, so nothing is sent the slow way that would not go there anyway)
                    const uint32_t bg = kXlate == 1 ? b[g] : (b[g] != 0u ? 0xffffffffu : 0u);
                    if (g < full) inv |= bg;
                    else if (g == full && part != 0u) inv |= bg & ((1u << (8u * part)) - 1u);
                }
                }
                const uint64_t c0 = static_cast<uint64_t>(u0.x) | (static_cast<uint64_t>(u0.y) << 32);
                const uint64_t c1 = static_cast<uint64_t>(u1.x) | (static_cast<uint64_t>(u1.y) << 32);
                const uint64_t tc = tb ? (c0 >> (2u * tb)) | (c1 << (64u - 2u * tb)) : c0;
                const uint32_t tm = tb ? (u0.z >> tb) | (u1.z << (32u - tb)) : u0.z;
                const uint64_t m64 = n_c == 32u ? ~0ull : (1ull << (2u * n_c)) - 1ull;
                const uint32_t m32 = n_c == 32u ? ~0u : (1u << n_c) - 1u;
                if (inv != 0u) bad_q = 1u;
                if (((qc ^ tc) & m64) != 0ull || (tm & m32) != 0u) bad_t = 1u;
            }
        }
        const bool left = group_max<kGroup>(bad_q) != 0u;  // a symbol outside A C G T further front: the general route
        const bool hit = group_max<kGroup>(bad_t) == 0u;
        if (kExact) {
            if (left || !hit) {  // (no occurrence: the reference's frozen empty interval is the exact kernel's to find)
                if (writer) leftover[atomicAdd(n_leftover, 1u)] = q;
            } else {
                const uint32_t row = sv.isa[pos - rem];
                if (writer) {
                    out_start[q] = row;
                    out_end[q] = row + 1u;
                    if (out_count) out_count[q] = 1u;
                    if (out_status) out_status[q] = 0;
                }
            }
        } else if (writer) {
            if (left) {
                leftover[atomicAdd(n_leftover, 1u)] = q;
                if (state) state[q] = make_uint4(0u, 0u, 0u, 0u);
                if (out_compact) out_compact[q] = kCompactSee;
            } else {
                if (out_compact) out_compact[q] = hit ? pos - rem : kCompactNone;
                else if (out_rec) out_rec[q] = hit ? make_uint4(0u, 1u, pos - rem, kRecResolved) : make_uint4(0u, 0u, 0xffffffffu, 0u);
                if (out_count) out_count[q] = hit ? 1u : 0u;
                if (out_status) out_status[q] = 0;
            }
        }
    }
}

// packed queries (2 bits per symbol): 4 lanes per query, plain loads
template <int kJump, int kMode>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(7, 7))) void search_pair_packed_kernel4(GDX_SEARCH_ARGS)
{
    search_pair_body<0, 4, false, kJump, kMode, true>(GDX_SEARCH_FWD);
}
// accounting variants (gdx_search_step_stats_dev): the counters cost registers, so they are kept out of the
// timed kernels; always the exact mode, whose LF steps are the reference's
template <int kPolicy, int kJump, int kMode>
__global__ __launch_bounds__(kBlock) void search_pair_stats_kernel8(GDX_SEARCH_ARGS)
{
    search_pair_body<kPolicy, 8, true, kJump, kMode>(GDX_SEARCH_FWD);
}
template <int kPolicy, int kJump, int kMode>
__global__ __launch_bounds__(kBlock) void search_pair_stats_kernel4(GDX_SEARCH_ARGS)
{
    search_pair_body<kPolicy, 4, true, kJump, kMode>(GDX_SEARCH_FWD);
}

// Cursor::extend_query_front for m independent cursors (cursor.rs:34-51).  kGroup lanes per cursor as in
// search_kernel; with pair lines (kPair) a symbol in 1..4 costs one 128-byte fetch and no table lookup.
template <class Table, int kGroup, bool kPair>
__global__ __launch_bounds__(kBlock) void extend_front_kernel(IndexView ix, uint32_t *__restrict__ start,
                                                              uint32_t *__restrict__ end,
                                                              const uint8_t *__restrict__ io_symbols, uint64_t m,
                                                              uint8_t *__restrict__ out_status)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * (kBlock / kGroup);
    const bool writer = (threadIdx.x % kGroup) == 0;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * (kBlock / kGroup) + threadIdx.x / kGroup; i < m; i += stride) {
        uint32_t lo = start[i], hi = end[i];
        const uint32_t c = ix.io_to_dense[io_symbols[i]];  // cursor.rs:34-38: translated before anything else
        uint32_t status = GDX_Q_OK;
        if (c == 0) {
            status = GDX_Q_INVALID_SYMBOL;
        } else if (lo != hi) {  // cursor.rs:41-48
            if (kPair && c <= 4u) {
                PairTable::lf1<1, 8>(ix, c, lo, hi, lo, hi);
            } else {
                uint32_t rlo, rhi;
                Table::rank2(ix, c, lo, hi, rlo, rhi);
                const uint32_t cc = ix.count[c];
                lo = cc + rlo;
                hi = cc + rhi;
            }
            if (writer) {
                start[i] = lo;
                end[i] = hi;
            }
        }
        if (out_status && writer) out_status[i] = static_cast<uint8_t>(status);
    }
}

template <class Table>
__global__ __launch_bounds__(kBlock) void rank_many_kernel(IndexView ix, const uint8_t *__restrict__ symbols,
                                                           const uint32_t *__restrict__ idx, uint64_t m,
                                                           uint32_t *__restrict__ out, uint32_t *error)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < m; i += stride) {
        const uint32_t c = symbols[i], p = idx[i];
        if (c >= static_cast<uint32_t>(ix.sigma) || p > ix.n) {  // mod.rs:107-108
            *error = 1;
            out[i] = 0;
            continue;
        }
        out[i] = Table::rank(ix, c, p);
    }
}

template <class Table>
__global__ __launch_bounds__(kBlock) void symbol_at_kernel(IndexView ix, const uint32_t *__restrict__ idx,
                                                           uint64_t m, uint8_t *__restrict__ out, uint32_t *error)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < m; i += stride) {
        const uint32_t p = idx[i];
        if (p >= ix.n) {  // condensed.rs:344
            *error = 1;
            out[i] = 0;
            continue;
        }
        out[i] = static_cast<uint8_t>(Table::symbol_at(ix, p));
    }
}

// gdx_bench_lf_walk_dev: the text read backwards through the occurrence table alone (symbol_at + rank + count, i.e.
// sampled_suffix_array.rs:118-131 without the samples)
template <class Table>
__global__ __launch_bounds__(kBlock) void lf_walk_kernel(IndexView ix, const uint32_t *__restrict__ rows, uint64_t m,
                                                         uint32_t steps, uint8_t *__restrict__ symbols,
                                                         uint32_t *__restrict__ end_rows)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < m; i += stride) {
        uint32_t row = rows[i];
        uint8_t *out = symbols + i * steps;
        uint32_t j = 0;
        for (; j < steps && row < ix.n; j++) {
            uint32_t r;
            const uint32_t c = Table::symbol_and_rank(ix, row, r);
            out[j] = static_cast<uint8_t>(c);
            if (c == 0) {
                j++;
                break;
            }
            row = ix.count[c] + r;
        }
        for (; j < steps; j++) out[j] = 0xffu;
        if (end_rows) end_rows[i] = row;
    }
}

// lookup_table.rs:163-258: table[depth][idx] = interval of the depth-mer whose j-th symbol is digit j
// of idx in base k (+1 to skip the sentinel).  Plain backward search with freeze-on-empty gives the
// same (start, end) as the reference's recursive fill through the smaller tables.
template <class Table>
__global__ __launch_bounds__(kBlock) void fill_lookup_kernel(IndexView ix, uint2 *__restrict__ lookup, int depth,
                                                             uint64_t entries)
{
    const uint32_t k = static_cast<uint32_t>(ix.n_searchable);
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    const uint64_t first = lookup_offset(k, static_cast<uint32_t>(depth));
    for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; e < entries; e += stride) {
        uint64_t pw = 1;
        for (int j = 1; j < depth; j++) pw *= k;  // k^(depth-1)
        uint32_t lo = 0, hi = ix.n;
        for (int j = depth - 1; j >= 0 && lo != hi; j--) {
            const uint32_t c = static_cast<uint32_t>((e / pw) % k) + 1u;
            uint32_t rlo, rhi;
            Table::rank2(ix, c, lo, hi, rlo, rhi);
            const uint32_t cc = ix.count[c];
            lo = cc + rlo;
            hi = cc + rhi;
            pw /= k;
        }
        lookup[first + e] = make_uint2(lo, hi);
    }
}

// top[idx]: idx holds the 2-bit codes (dense - 1) of the D symbols in consumption order, first consumed symbol in
// the highest bit pair (top_lookup builds the same index from the query's nibble codes).  Plain backward search
// with freeze-on-empty, exactly fill_lookup_kernel's.
__global__ __launch_bounds__(kBlock) void fill_top_kernel(IndexView ix, uint2 *__restrict__ top, uint32_t depth,
                                                          uint64_t entries)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; e < entries; e += stride) {
        uint32_t lo = 0, hi = ix.n;
        for (uint32_t s = 0; s < depth && lo != hi; s++) {
            const uint32_t c = (static_cast<uint32_t>(e >> (2u * (depth - 1u - s))) & 3u) + 1u;
            uint32_t rlo, rhi;
            LineTable::rank2(ix, c, lo, hi, rlo, rhi);
            const uint32_t cc = ix.count[c];
            lo = cc + rlo;
            hi = cc + rhi;
        }
        top[e] = make_uint2(lo, hi);  // an empty interval stays as it was when it emptied (the loop stopped there)
    }
}

}  // namespace

#define GDX_DISPATCH_TABLE(ix, KERNEL, grid, stream, ...)                                        \
    do {                                                                                         \
        if ((ix).layout == 0)                                                                    \
            hipLaunchKernelGGL(KERNEL<LineTable>, dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); \
        else                                                                                     \
            hipLaunchKernelGGL(KERNEL<GenericTable>, dim3(grid), dim3(kBlock), 0, stream, __VA_ARGS__); \
    } while (0)

static unsigned grid_for_items(uint64_t items)
{
    const uint64_t blocks = (items + kBlock - 1) / kBlock;
    const uint64_t cap = 256u * 8u;  // 8 blocks of 256 threads per CU
    return static_cast<unsigned>(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

// Kernel used on rank lines: 2 = pair lines when the index has them (default), 0 = quad, 1 = one lane
// per query.  Settable through GDX_SEARCH_VARIANT=pair|quad|lane or gdx_debug_set_search_variant()
// (A/B measurements and parity tests of every variant).
static std::atomic<int> g_search_variant{-1};

void set_search_variant(int v) { g_search_variant.store(v); }

static int search_variant()
{
    int v = g_search_variant.load();
    if (v >= 0) return v;
    const char *e = getenv("GDX_SEARCH_VARIANT");
    v = 2;
    if (e && std::string(e) == "lane") v = 1;
    if (e && std::string(e) == "quad") v = 0;
    g_search_variant.store(v);
    return v;
}

void launch_search(const IndexView &ix, const uint8_t *d_qbuf, const uint64_t *d_qoff, uint64_t nq,
                   uint32_t *d_out_start, uint32_t *d_out_end, uint32_t *d_out_count, uint8_t *d_out_status,
                   hipStream_t stream, unsigned long long *d_step_stats, uint2 *d_hint, const QueryOptions &qo)
{
    SearchCall c;
    c.d_qbuf = d_qbuf;
    c.d_qbeg = d_qoff;
    c.d_qend = d_qoff + 1;
    c.nq = nq;
    c.d_start = d_out_start;
    c.d_end = d_out_end;
    c.d_count = d_out_count;
    c.d_status = d_out_status;
    c.d_hint = d_hint;
    c.d_step_stats = d_step_stats;
    launch_search_call(ix, c, stream, qo);
}

// ZeroSet (kernels.hpp): up to eight small regions zeroed by one launch
struct ZeroSegments {
    uint32_t *p[ZeroSet::kMax];
    uint32_t end[ZeroSet::kMax];  // running totals of the regions' words
    int n;
};
__global__ __launch_bounds__(kBlock) void zero_segments_kernel(ZeroSegments z)
{
    const uint32_t total = z.end[z.n - 1];
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        int s = 0;
        while (i >= z.end[s]) s++;
        z.p[s][i - (s == 0 ? 0u : z.end[s - 1])] = 0u;
    }
}

void ZeroSet::add(void *ptr, size_t bytes)
{
    if (ptr == nullptr || bytes == 0) return;
    if (n == kMax || (reinterpret_cast<uintptr_t>(ptr) & 3u) != 0 || bytes > 0xfffffff0ull)
        fail(GDX_ERR_INVALID_ARGUMENT, "internal: ZeroSet takes up to %d 4-byte aligned regions", kMax);
    p[n] = static_cast<uint32_t *>(ptr);
    words[n] = static_cast<uint32_t>((bytes + 3) / 4);
    n++;
}

void ZeroSet::flush(hipStream_t stream)
{
    if (n == 0) return;
    ZeroSegments z{};
    uint64_t total = 0;
    for (int i = 0; i < n; i++) {
        z.p[i] = p[i];
        total += words[i];
        if (total > 0xffffffffull) fail(GDX_ERR_INVALID_ARGUMENT, "internal: ZeroSet regions beyond 16 GB");
        z.end[i] = static_cast<uint32_t>(total);
    }
    z.n = n;
    const uint64_t blocks = (total + kBlock * 4 - 1) / (kBlock * 4);
    hipLaunchKernelGGL(zero_segments_kernel, dim3(static_cast<unsigned>(blocks < 1024 ? blocks : 1024)), dim3(kBlock), 0, stream, z);
    n = 0;
}

// offsets of a uniform batch, for the kernels that read them from memory (fill_uniform_offsets_kernel)
__global__ __launch_bounds__(kBlock) void fill_uniform_offsets_kernel(uint64_t *__restrict__ off, uint64_t n, uint32_t ulen)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) off[i] = i * ulen;
}

// The tile sums of a locate that the seed table's lane kernel began (SearchCall::d_tile_sums): the reads on its two lists,
// once every kernel of the call has written their results.  left: everything listed for the next kernels (compact result
// "see the record": the record's count, also into *rest); long: reads finished by seed_text_kernel4 (those it handed on
// are on the left list as well, and counted there).
// the same for the queries of a list only (its length is a device value): off[q], off[q + 1] -- what the offset-reading
// kernels behind the seed chain need of a uniform batch, instead of 8 bytes per query of the whole batch
__global__ __launch_bounds__(kBlock) void fill_uniform_offsets_list_kernel(uint64_t *__restrict__ off, const uint32_t *__restrict__ list,
                                                                          const uint32_t *__restrict__ n_list, uint32_t ulen)
{
    const uint64_t n = *n_list;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * kBlock) {
        const uint64_t q = list[i];
        off[q] = q * ulen;
        off[q + 1] = (q + 1) * ulen;
    }
}

__global__ __launch_bounds__(kBlock) void tile_sums_lists_kernel(const uint4 *__restrict__ rec, const uint32_t *__restrict__ compact,
                                                                 uint32_t max_hits, const uint32_t *__restrict__ left,
                                                                 const uint32_t *__restrict__ n_left, const uint32_t *__restrict__ lng,
                                                                 const uint32_t *__restrict__ n_lng,
                                                                 unsigned long long *__restrict__ tile_sums,
                                                                 unsigned long long *__restrict__ rest)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    const uint64_t nl = *n_left, ng = *n_lng;
    const uint64_t n = nl + ng, n_round = (n + 63) / 64 * 64;  // (whole wavefronts: the ballots below need every lane)
    unsigned long long open_slots = 0;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_round; i += stride) {
        unsigned long long c = 0;
        uint32_t tile = 0xffffffffu;
        if (i < n) {
            const bool is_left = i < nl;
            const uint32_t q = is_left ? left[i] : lng[i - nl];
            const uint32_t c4 = compact[q];
            tile = q / kSumTile;
            if (c4 == kCompactSee) {
                if (is_left) {
                    const uint2 v = *reinterpret_cast<const uint2 *>(rec + q);
                    const uint32_t m = v.y - v.x;
                    c = (max_hits != 0u && m > max_hits) ? 0ull : static_cast<unsigned long long>(m);
                    open_slots += c;
                }
            } else {
                // long list: answered by seed_text_kernel4.  left list: no kernel of today's chains turns a listed read's
                // compact result into a position -- if one ever does, it counts as RecordSize counts it (a read that
                // seed_text_kernel4 handed on sits on both lists and must then be counted on one of them only)
                c = c4 == kCompactNone ? 0ull : 1ull;
            }
        }
        // the lists are filled block by block, range by range: the 64 entries of a wavefront mostly lie in one tile --
        // then one atomic for all of them (a batch of long reads lists tens of millions)
        const uint32_t tile0 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(tile)));
        if (__ballot(tile != tile0 && c != 0ull) == 0ull) {
            unsigned long long sum = tile == tile0 ? c : 0ull;
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
            if ((threadIdx.x & 63u) == 0u && sum != 0ull && tile0 != 0xffffffffu) atomicAdd(&tile_sums[tile0], sum);
        } else if (c != 0ull) {
            atomicAdd(&tile_sums[tile], c);
        }
    }
    for (int off = 32; off > 0; off >>= 1) open_slots += __shfl_xor(open_slots, off);
    if ((threadIdx.x & 63u) == 0u && open_slots != 0ull) atomicAdd(rest, open_slots);
}

void launch_search_call(const IndexView &ix, const SearchCall &call, hipStream_t stream, const QueryOptions &qo)
{
    SearchCall c = call;
    const uint64_t nq = c.nq;
    if (nq == 0) return;
    if (c.mode < 0 || c.mode > 2) fail(GDX_ERR_INVALID_ARGUMENT, "internal: search mode %d", c.mode);
    const int variant = qo.search_variant >= 0 ? qo.search_variant : search_variant();
    if (c.packed && (ix.layout != 0 || ix.n_searchable < 4))
        fail(GDX_ERR_UNSUPPORTED, "packed queries need the rank-line layout (sigma <= 8) with dense symbols 1..4 searchable");
    if (c.packed && (c.mode == 2 || c.d_step_stats != nullptr)) fail(GDX_ERR_UNSUPPORTED, "packed queries: search and count / locate only");
    // Uniform batches (SearchCall::uniform_len): the seed chain and the rank-line kernels compute where a query lies; the
    // pair-line kernels read offsets, which are then written once into scratch (8 bytes per query, one streaming pass)
    // (a count / locate call may leave all but a short list to the seed chain, which computes too: decided below)
    const bool pair_offsets = c.uniform_len != 0u && ix.layout == 0 && variant == 2 && ix.pair_lines != nullptr;
    uint64_t *d_uniform_off = nullptr;
    auto dense_uniform_offsets = [&] {
        d_uniform_off = static_cast<uint64_t *>(stream_scratch(stream, 15, (nq + 1) * sizeof(uint64_t)));
        hipLaunchKernelGGL(fill_uniform_offsets_kernel, dim3(grid_for_items(nq + 1)), dim3(kBlock), 0, stream, d_uniform_off, nq + 1,
                           c.uniform_len);
        c.d_qbeg = d_uniform_off;
        c.d_qend = d_uniform_off + 1;
        c.uniform_len = 0;
    };
    if (pair_offsets && c.mode != 1) dense_uniform_offsets();
    const uint32_t ulen = c.uniform_len;
    // the call's counters (and what the caller wants zeroed with them) are zeroed by one launch before its first kernel
    ZeroSet zs;
    if (c.also_zero != nullptr) zs = *c.also_zero;
    // Launch geometry of the group kernels (measured on MI355X, hg38-scale index, 100 M reads,
    // profiles/r01/search_variants.md): many short-lived blocks beat a resident grid -- 65536 blocks: 78 ms,
    // 1792 (7 per CU): 89 ms, 2048 (8 per CU, all resident, lock-step): 112 ms.  So: about 48 queries per
    // group, at most 65536 blocks, never fewer blocks than groups need.
    // Experiments: GDX_SEARCH_GRID = absolute number of blocks, GDX_SEARCH_PAD = dynamic LDS bytes per block.
    static const long grid_override = [] { const char *e = getenv("GDX_SEARCH_GRID"); return e ? atol(e) : 0L; }();
    static const long pad_override = [] { const char *e = getenv("GDX_SEARCH_PAD"); return e ? atol(e) : 0L; }();
    const unsigned lds_pad = static_cast<unsigned>(pad_override);
    auto group_grid = [&](uint64_t groups_per_block) {
        const uint64_t needed = (nq + groups_per_block - 1) / groups_per_block;
        uint64_t blocks = (nq + groups_per_block * 48 - 1) / (groups_per_block * 48);
        if (blocks < 1792) blocks = 1792;
        if (blocks > 65536) blocks = 65536;
        if (blocks > needed) blocks = needed;
        if (grid_override > 0) blocks = static_cast<uint64_t>(grid_override);
        return static_cast<unsigned>(blocks);
    };
    // lanes per query of the pair-line kernels: 8 (one 16-byte chunk per lane) or 4 (two chunks per lane, twice the queries
    // in flight); QueryOptions::search_lanes, else GDX_SEARCH_LANES, else 4
    static const int env_lanes = [] {
        const char *e = getenv("GDX_SEARCH_LANES");
        return (e && atoi(e) == 8) ? 8 : 4;
    }();
    const int lanes = (qo.search_lanes == 4 || qo.search_lanes == 8) ? qo.search_lanes : env_lanes;
    // cache policy of their line / entry loads (see below): QueryOptions::load_policy / GDX_LOAD_POLICY=0|1
    static const int env_policy = [] {
        const char *e = getenv("GDX_LOAD_POLICY");
        return e ? atoi(e) : 0;
    }();
    const int policy = (qo.load_policy >= 0 ? qo.load_policy : env_policy) == 1 ? 1 : 0;
    CursorArgs ca = c.cursors;
    bool leftover_list = false;  // ca.active_in is the (short) list another kernel of this call left over
    uint32_t *seed_list = nullptr;  // ... the seed kernel's, with each read's state in its record slot: search_fast_kernel4 next
    uint32_t seed_state_packed = 0;  // ... in the packed form (kStatePacked)
    bool compact_by_seed = false;   // c.d_compact has been filled by the seed kernel
    uint32_t *fold_left = nullptr, *fold_long = nullptr;  // the lane kernel's lists when it counts the hit totals (d_tile_sums)
    if (c.tile_sums_done != nullptr) *c.tile_sums_done = false;
    if (c.d_compact != nullptr && (c.mode != 1 || c.d_rec == nullptr))
        fail(GDX_ERR_INVALID_ARGUMENT, "internal: compact results go with the records of a count / locate search");
    // Count / locate searches on an index with text units and no jump table: top table, then the rest of the query against
    // the text at SA[row] (search_verify_kernel4); what it cannot finish is listed for the general kernel of the index
    // (pair lines or rank lines) below.  QueryOptions::search_fast = 0 switches it off like the other fast path.
    {
        static const int env_fast_v = [] { const char *e = getenv("GDX_SEARCH_FAST"); return e ? atoi(e) : -1; }();
        static const int env_seed = [] { const char *e = getenv("GDX_SEARCH_SEED"); return e ? atoi(e) : -1; }();
        const bool clean_call = c.mode == 1 && ix.layout == 0 && ix.text_units != nullptr && ix.n_searchable >= 4 &&
                                c.d_step_stats == nullptr && c.d_hint == nullptr && c.d_start == nullptr &&
                                c.d_end == nullptr && ca.active_in == nullptr && nq < 0xffffffffull;
        // (a configured lookup table deeper than the seed keeps its own check of the symbols between the two depths, as
        // with the top table below)
        const bool seed = clean_call && ix.seed != nullptr && ix.seed_k >= static_cast<uint32_t>(ix.depth) &&
                          (env_seed >= 0 ? env_seed != 0 : qo.search_seed != 0);
        const bool verify = !seed && clean_call && ix.top != nullptr && ix.top_depth >= 1u && ix.jump == nullptr &&
                            ix.top_depth >= static_cast<uint32_t>(ix.depth) &&
                            (env_fast_v >= 0 ? env_fast_v != 0 : qo.search_fast != 0);
        if (seed || verify) {
            uint64_t per_block = (nq + 1791) / 1792;
            per_block = (per_block + 63) / 64 * 64;
            const uint32_t v_range = static_cast<uint32_t>(per_block > kMaxRange ? kMaxRange : per_block);
            const uint64_t v_ranges = (nq + v_range - 1) / v_range;
            const unsigned v_blocks = static_cast<unsigned>(v_ranges < (1u << 20) ? v_ranges : (1u << 20));
            uint32_t *d_left = static_cast<uint32_t *>(stream_scratch(stream, 11, (nq + 4) * sizeof(uint32_t)));
            zs.add(d_left, sizeof(uint32_t));
            // rows a verify round takes: four when SA[row] is one fetch away, one when it costs a locate walk
            const bool entry_sa = ix.jump != nullptr && ix.jump_bytes == 32;
            const uint32_t max_rows = (ix.sa_full != nullptr || entry_sa) ? 4u : 1u;
            const VerifyView vv{ix.top, ix.text_units, ix.sa_full, entry_sa ? ix.jump : nullptr, ix.lines, ix.sb_offsets, ix.count,
                                ix.sa_samples, ix.border_keys, ix.border_vals, ix.io_to_dense, ix.top_depth, ix.n, ix.n_texts,
                                ix.sa_inv, ix.sa_rot, ix.sa_limit, max_rows, ix.perm_code_lo, ix.perm_code_hi, ix.perm_exp_lo,
                                ix.perm_exp_hi, ix.perm_mask, ix.seed, ix.seed_buckets, ix.seed_k, ix.seed_tag_bits, ix.seed_pairs, ix.seed_quads};
            static const bool env_no_perm_v = getenv("GDX_SEARCH_NO_PERM") != nullptr;
            // how the kernels get 2-bit codes: 2 = the buffer holds them (packed queries), 1 = v_perm tables, 0 = the table in LDS
            const int xlate = c.packed ? 2 : ((ix.perm_ok && !env_no_perm_v) ? 1 : 0);
#define GDX_VERIFY_LAUNCH_X(XLATE, SEED, BLOCKS, RANGE, LEFT, LIST)                                                           \
    hipLaunchKernelGGL((search_verify_kernel4<XLATE, SEED>), dim3(BLOCKS), dim3(kBlock), 0, stream, vv, c.d_qbuf, c.d_qbeg,    \
                       c.d_qend, nq, c.d_count, c.d_status, c.d_rec, RANGE, (LEFT) + 4, LEFT, (LIST) ? (LIST) + 4 : nullptr, LIST, ulen, \
                       verify_state, verify_wide, verify_state != nullptr ? seed_state_packed : 0u)
#define GDX_VERIFY_LAUNCH(SEED, BLOCKS, RANGE, LEFT, LIST)                             \
    do {                                                                               \
        if (xlate == 2) GDX_VERIFY_LAUNCH_X(2, SEED, BLOCKS, RANGE, LEFT, LIST);       \
        else if (xlate == 1) GDX_VERIFY_LAUNCH_X(1, SEED, BLOCKS, RANGE, LEFT, LIST);  \
        else GDX_VERIFY_LAUNCH_X(0, SEED, BLOCKS, RANGE, LEFT, LIST);                  \
    } while (0)
            uint32_t *const no_list = nullptr;
            const uint4 *verify_state = nullptr;  // (set below: the seed kernel's states for the verify kernel over its list)
            uint32_t verify_wide = 0;
            if (seed) {
                // the seed table's own kernel first (absent k-mers and k-mers that occur once); what it lists -- k-mers on
                // several rows, reads shorter than the seed, other symbols -- goes on in search_fast_kernel4 from the entry's
                // interval when the index has top and jump tables (one jump round for the rest of a read from a repeat),
                // else through the seed-aware verify kernel; their leftovers go to the general kernel.  The lists' lengths
                // are only known on the device: small ranges, a capped grid that strides over whatever there is.
                static const int env_lean = [] { const char *e = getenv("GDX_SEARCH_SEED_LEAN"); return e ? atoi(e) : 1; }();
                static const int env_chain = [] { const char *e = getenv("GDX_SEARCH_SEED_CHAIN"); return e ? atoi(e) : 1; }();
                // (to_fast implies every condition of `fast` below)
                const bool to_fast = env_lean != 0 && env_chain != 0 && variant == 2 && ix.pair_lines != nullptr && ix.top != nullptr &&
                                     ix.top_depth >= 1u && ix.jump != nullptr && ix.top_depth >= static_cast<uint32_t>(ix.depth) &&
                                     lanes == 4 && policy == 0 && (env_fast_v >= 0 ? env_fast_v != 0 : qo.search_fast != 0);
                static const int env_packed = [] { const char *e = getenv("GDX_SEARCH_SEED_PACKED"); return e ? atoi(e) : 1; }();
                // (... or the verify kernel's, which hands the plain form on to the general kernel: to_verify_with_state below)
                const bool to_verify_packed = !to_fast && variant == 2 && ix.pair_lines != nullptr && env_lean != 0;
                seed_state_packed = (to_fast || to_verify_packed) && c.d_rec != nullptr && env_packed != 0 ? 1u : 0u;
                // (2: states of two-copy repeats name their record in IndexView::seed_pairs -- only the verify kernel reads those)
                const char *env_pairs = getenv("GDX_SEARCH_SEED_PAIRS");  // (0: the A/B; read per call, a test switches it)
                if (seed_state_packed != 0u && !to_fast && (ix.seed_pairs != nullptr || ix.seed_quads != nullptr) && !(env_pairs != nullptr && atoi(env_pairs) == 0))
                    seed_state_packed = 2u;
                if (env_lean != 0) {
                    uint32_t *d_first = static_cast<uint32_t *>(stream_scratch(stream, 12, (nq + 4) * sizeof(uint32_t)));
                    zs.add(d_first, sizeof(uint32_t));
                    const SeedView sv{ix.seed, ix.text_units, nullptr, ix.io_to_dense, ix.seed_buckets, ix.seed_k, ix.seed_tag_bits,
                                      ix.perm_code_lo, ix.perm_code_hi, ix.perm_exp_lo, ix.perm_exp_hi, ix.perm_mask};
                    // (the pair-line general kernel resumes listed reads from their k-mer's interval as well: the verify kernel in
                    // between takes the narrow ones and leaves the wide ones to it)
                    const bool to_verify_with_state = !to_fast && variant == 2 && ix.pair_lines != nullptr && c.d_rec != nullptr;
                    uint4 *d_seed_state = (to_fast || to_verify_with_state) ? c.d_rec : nullptr;
                    uint32_t *const none = nullptr;
                    // reads longer than a seed entry covers: listed with {position, symbols in front} in the first half of their
                    // record slot (an array of its own when the call has no records) for seed_text_kernel4, whose own
                    // leftovers (a symbol outside A C G T further front) join the seed kernel's list
                    uint32_t *d_long = static_cast<uint32_t *>(stream_scratch(stream, 13, (nq + 4) * sizeof(uint32_t)));
                    zs.add(d_long, sizeof(uint32_t));
                    uint2 *d_long_state = c.d_rec != nullptr ? reinterpret_cast<uint2 *>(c.d_rec)
                                                             : static_cast<uint2 *>(stream_scratch(stream, 14, nq * sizeof(uint2)));
                    const uint32_t long_stride = c.d_rec != nullptr ? 2u : 1u;
                    const uint64_t t_groups = (nq + kBlock / 4 - 1) / (kBlock / 4);
                    const unsigned t_blocks = static_cast<unsigned>(t_groups < 8192 ? t_groups : 8192);
                    // (experiments: GDX_SEED_PAD = dynamic LDS bytes per block, which caps the resident blocks per CU)
                    static const unsigned seed_pad = [] { const char *e = getenv("GDX_SEED_PAD"); return e ? static_cast<unsigned>(atol(e)) : 0u; }();
#define GDX_SEED_LAUNCH(XLATE)                                                                                                  \
    do {                                                                                                                       \
        hipLaunchKernelGGL((search_seed_kernel4<XLATE, false>), dim3(v_blocks), dim3(kBlock), seed_pad, stream, sv, c.d_qbuf,  \
                           c.d_qbeg, c.d_qend, nq, c.d_count, c.d_status, c.d_rec, none, none, v_range, d_first + 4,          \
                           d_first, d_seed_state, c.d_compact, d_long + 4, d_long, d_long_state, long_stride,                 \
                           seed_state_packed, ulen, CursorArgs());                                                             \
        hipLaunchKernelGGL((seed_text_kernel4<XLATE, false>), dim3(t_blocks), dim3(kBlock), 0, stream, sv, c.d_qbuf, c.d_qbeg, \
                           d_long + 4, d_long, d_long_state, long_stride, c.d_count, c.d_status, c.d_rec, none, none,         \
                           c.d_compact, d_seed_state, d_first + 4, d_first, ulen);                                            \
    } while (0)
                    // one lane per read (search_seed_lane_kernel) where its loads apply: 2-bit codes or v_perm tables, k <= 24, a
                    // dword-aligned buffer; GDX_SEED_LANE=0: the four-lane kernel
                    static const int env_lane = [] { const char *e = getenv("GDX_SEED_LANE"); return e ? atoi(e) : 1; }();
                    const bool lane_kernel = env_lane != 0 && xlate != 0 && ix.seed_k <= 24u && ix.seed_k >= 8u &&
                                             (reinterpret_cast<uintptr_t>(c.d_qbuf) & 3u) == 0;
                    // the locate's hit totals folded into this call (SearchCall::d_tile_sums): the lane kernel counts what it
                    // answers, tile_sums_lists_kernel adds its two lists once the whole chain has run (finish_fold below)
                    unsigned long long *fold = nullptr;
                    if (lane_kernel && c.d_tile_sums != nullptr && c.d_tile_rest != nullptr && c.d_compact != nullptr) {
                        fold = c.d_tile_sums;
                        zs.add(fold, ((nq + kSumTile - 1) / kSumTile) * sizeof(unsigned long long));
                        fold_left = d_first;
                        fold_long = d_long;
                    }
                    // (its ranges: at most kLaneRange reads, the capacity of its lists in LDS)
                    // (experiments: GDX_SEED_LANE_RANGE = reads per range, a multiple of 64 up to kLaneRange)
                    static const uint32_t env_lane_range = [] { const char *e = getenv("GDX_SEED_LANE_RANGE"); return e ? static_cast<uint32_t>(atol(e)) : 0u; }();
                    const uint32_t lane_range = env_lane_range >= 64u && env_lane_range <= kLaneRange ? env_lane_range / 64u * 64u
                                                : (v_range < kLaneRange ? v_range : kLaneRange);
                    if (lane_range % 64u != 0u) fail(GDX_ERR_INVALID_ARGUMENT, "internal: the lane kernel's ranges are multiples of 64");
                    const uint64_t lane_ranges = (nq + lane_range - 1) / lane_range;
                    unsigned lane_blocks = static_cast<unsigned>(lane_ranges < (1u << 20) ? lane_ranges : (1u << 20));
                    // (tests: GDX_SEED_LANE_BLOCKS caps the grid, so that a block takes several ranges -- its parked queue and
                    // its tile sums then cross range borders, which a grid of one block per range never shows; read per call)
                    if (const char *e = getenv("GDX_SEED_LANE_BLOCKS")) {
                        const long cap = atol(e);
                        if (cap > 0 && static_cast<unsigned long>(cap) < lane_blocks) lane_blocks = static_cast<unsigned>(cap);
                    }
#define GDX_SEED_LANE_LAUNCH(XLATE, UNIFORM)                                                                                    \
    do {                                                                                                                       \
        hipLaunchKernelGGL((search_seed_lane_kernel<XLATE, UNIFORM>), dim3(lane_blocks), dim3(kBlock), seed_pad, stream, sv,   \
                           c.d_qbuf, c.d_qbeg, c.d_qend, nq, c.d_count, c.d_status, c.d_rec, lane_range, d_first + 4, d_first, \
                           d_seed_state, c.d_compact, d_long + 4, d_long, d_long_state, long_stride, seed_state_packed, ulen,   \
                           fold);                                                                                              \
        hipLaunchKernelGGL((seed_text_kernel4<XLATE, false>), dim3(t_blocks), dim3(kBlock), 0, stream, sv, c.d_qbuf, c.d_qbeg, \
                           d_long + 4, d_long, d_long_state, long_stride, c.d_count, c.d_status, c.d_rec, none, none,         \
                           c.d_compact, d_seed_state, d_first + 4, d_first, ulen);                                            \
    } while (0)
                    zs.flush(stream);
                    if (lane_kernel && xlate == 2 && ulen != 0u) GDX_SEED_LANE_LAUNCH(2, true);
                    else if (lane_kernel && xlate == 2) GDX_SEED_LANE_LAUNCH(2, false);
                    else if (lane_kernel && ulen != 0u) GDX_SEED_LANE_LAUNCH(1, true);
                    else if (lane_kernel) GDX_SEED_LANE_LAUNCH(1, false);
                    else if (xlate == 2) GDX_SEED_LAUNCH(2);
                    else if (xlate == 1) GDX_SEED_LAUNCH(1);
                    else GDX_SEED_LAUNCH(0);
#undef GDX_SEED_LANE_LAUNCH
#undef GDX_SEED_LAUNCH
                    compact_by_seed = true;
                    if (to_fast) {
                        seed_list = d_first;
                    } else {
                        // (the list's length is only known on the device: a grid the chip holds at once strides over whatever
                        // there is; 8192 blocks of this kernel's 12 KB of LDS took 13 us to dispatch and leave when the list
                        // was empty -- on a text without repeats it nearly is)
                        const uint32_t l_range = 256;
                        const uint64_t l_ranges = (nq + l_range - 1) / l_range;
                        static const unsigned l_cap = [] { const char *e = getenv("GDX_SEED_LIST_BLOCKS"); return e ? static_cast<unsigned>(atol(e)) : 1792u; }();
                        const unsigned l_blocks = static_cast<unsigned>(l_ranges < l_cap ? l_ranges : l_cap);
                        if (to_verify_with_state) {
                            verify_state = d_seed_state;
                            verify_wide = 1u;
                            ca.resume_state = d_seed_state;
                        }
                        GDX_VERIFY_LAUNCH(true, l_blocks, l_range, d_left, d_first);
                    }
                } else {
                    zs.flush(stream);
                    GDX_VERIFY_LAUNCH(true, v_blocks, v_range, d_left, no_list);
                }
            } else {
                zs.flush(stream);
                GDX_VERIFY_LAUNCH(false, v_blocks, v_range, d_left, no_list);
            }
#undef GDX_VERIFY_LAUNCH
#undef GDX_VERIFY_LAUNCH_X
            static const bool env_stats_v = getenv("GDX_SEARCH_FAST_STATS") != nullptr;  // debug: size of the leftover list
            if (env_stats_v) {
                uint32_t n_left = 0;
                GDX_HIP(hipMemcpyAsync(&n_left, seed_list ? seed_list : d_left, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                GDX_HIP(hipStreamSynchronize(stream));
                fprintf(stderr, "gdx: %s path left %u of %llu queries to the next kernel\n", seed ? "seed" : "verify", n_left,
                        static_cast<unsigned long long>(nq));
            }
            ca.active_in = (seed_list ? seed_list : d_left) + 4;
            ca.n_active_in = seed_list ? seed_list : d_left;
            leftover_list = true;
        }
    }
    zs.flush(stream);  // (a call that took none of the paths above: what the caller wanted zeroed)
    // a uniform batch in front of the pair-line kernels, which read offsets: those of the whole batch (8 bytes per query, one
    // streaming pass), or -- behind the seed chain -- of the listed queries only, written before each kernel that takes a list
    bool sparse_offsets = false;
    if (pair_offsets && c.mode == 1) {
        if (leftover_list) {
            d_uniform_off = static_cast<uint64_t *>(stream_scratch(stream, 15, (nq + 1) * sizeof(uint64_t)));
            c.d_qbeg = d_uniform_off;
            c.d_qend = d_uniform_off + 1;
            c.uniform_len = 0;
            sparse_offsets = true;
        } else {
            dense_uniform_offsets();
        }
    }
    auto offsets_for_list = [&](const uint32_t *list, const uint32_t *n_list) {
        if (sparse_offsets && list != nullptr)
            hipLaunchKernelGGL(fill_uniform_offsets_list_kernel, dim3(1024), dim3(kBlock), 0, stream, d_uniform_off, list, n_list, ulen);
    };
    auto finish_fold = [&] {
        if (fold_left == nullptr) return;
        // (the lists' lengths are only known on the device: a capped grid that strides over whatever there is)
        hipLaunchKernelGGL(tile_sums_lists_kernel, dim3(4096), dim3(kBlock), 0, stream, c.d_rec, c.d_compact, c.tile_max_hits,
                           fold_left + 4, fold_left, fold_long + 4, fold_long, c.d_tile_sums, c.d_tile_rest);
        if (c.tile_sums_done != nullptr) *c.tile_sums_done = true;
    };
    // compact results without the seed kernel: every query says "see the record"
    if (c.d_compact != nullptr && !compact_by_seed) GDX_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c.d_compact), static_cast<int>(kCompactSee), nq, stream));
    if (ix.layout == 0 && variant == 2 && ix.pair_lines != nullptr) {
        // Every block searches contiguous ranges of `range` queries (a multiple of 64, at most kMaxRange): about
        // 48 rounds per group at large batches, 1792+ blocks at small ones; GDX_SEARCH_GRID = number of blocks.
        uint64_t per_block = (nq + 1791) / 1792;
        per_block = (per_block + 63) / 64 * 64;
        const uint32_t range_cap = c.mode == 2 ? kCursorRange : kMaxRange;
        const uint32_t range = static_cast<uint32_t>(per_block > range_cap ? range_cap : per_block);
        const uint64_t n_ranges = (nq + range - 1) / range;
        const unsigned blocks = grid_override > 0 ? static_cast<unsigned>(grid_override)
                                                  : static_cast<unsigned>(n_ranges < (1u << 20) ? n_ranges : (1u << 20));
        // Cache policy of the line / entry loads: plain by default.  sc1 (no allocation in the CU's L1) was worth
        // +5 % while the first levels of the search were cache-resident pair lines; with the top table every load
        // is a DRAM miss and plain loads measure 3 % faster.  QueryOptions::load_policy / GDX_LOAD_POLICY=0|1.
        // (resolved above the seed block)
        // Ranges whose query lengths are spread out are searched in length order (order_range_by_length);
        // QueryOptions::length_schedule / GDX_SEARCH_SCHEDULE=0 keeps the query order.
        static const int env_schedule = [] {
            const char *e = getenv("GDX_SEARCH_SCHEDULE");
            return (e && e[0] == '0') ? 0 : 1;
        }();
        const int schedule = qo.length_schedule >= 0 ? (qo.length_schedule != 0) : env_schedule;
        // Stragglers are parked after their allowance + defer_after load rounds and finished together
        // (search_pair_body); QueryOptions::search_defer_after (resolved per index by FmIndex::query_options: on when
        // the text is repetitive) / GDX_SEARCH_DEFER; 0 = never.  The accounting kernels never park (their per-query
        // round counts describe the plain lock-step schedule).
        static const int env_defer = [] {
            const char *e = getenv("GDX_SEARCH_DEFER");
            return e ? atoi(e) : 0;
        }();
        const uint32_t defer_after = c.d_step_stats != nullptr ? 0u
                                     : static_cast<uint32_t>(qo.search_defer_after >= 0 ? qo.search_defer_after : env_defer);
#define GDX_PAIR_LAUNCH(KERNEL)                                                                                     \
    hipLaunchKernelGGL(KERNEL, dim3(g_blocks), dim3(kBlock), lds_pad, stream, ix, c.d_qbuf, c.d_qbeg, c.d_qend, nq, \
                       c.d_start, c.d_end, c.d_count, c.d_status, c.d_step_stats, g_range, schedule, c.d_hint,      \
                       c.d_rec, ca_general, defer_after)
#define GDX_PAIR_LAUNCH_W(KERNEL, P, M)                                    \
    do {                                                                   \
        if (ix.jump_bytes == 32) GDX_PAIR_LAUNCH((KERNEL<P, 32, M>));      \
        else if (ix.jump_bytes == 16) GDX_PAIR_LAUNCH((KERNEL<P, 16, M>)); \
        else GDX_PAIR_LAUNCH((KERNEL<P, 8, M>));                           \
    } while (0)
#define GDX_PAIR_LAUNCH_M(KERNEL, P)                      \
    do {                                                  \
        if (c.mode == 0) GDX_PAIR_LAUNCH_W(KERNEL, P, 0); \
        else if (c.mode == 1) GDX_PAIR_LAUNCH_W(KERNEL, P, 1); \
        else GDX_PAIR_LAUNCH_W(KERNEL, P, 2);             \
    } while (0)
        // Fast path (search_fast_kernel4): count / locate mode on an index with top and jump tables.  It finishes
        // what needs no pair line (intervals of up to sixteen rows) and lists the rest, which the general kernel --
        // with the straggler pass when the index asks for it -- takes up from the list where the fast path stopped.
        // QueryOptions::search_fast / GDX_SEARCH_FAST=0 switch it off.
        // (qo.search_fast arrives resolved by FmIndex::query_options; the environment variable is a debug override)
        static const int env_fast = [] { const char *e = getenv("GDX_SEARCH_FAST"); return e ? atoi(e) : -1; }();
        const bool fast = c.mode == 1 && c.d_step_stats == nullptr && lanes == 4 && policy == 0 &&
                          ix.top != nullptr && ix.top_depth >= 1u && ix.jump != nullptr &&
                          // the guards of the general kernel's top-table step (search_pair_body): dense 1..4 must all be
                          // searchable, and a configured lookup table deeper than the top table keeps its own check of
                          // the symbols between the two depths (a valid but unsearchable symbol there is an error, not a
                          // step) -- a leftover resumed after the top table would skip it
                          ix.n_searchable >= 4 && ix.top_depth >= static_cast<uint32_t>(ix.depth) &&
                          (ca.active_in == nullptr || seed_list != nullptr) && c.d_hint == nullptr && c.d_start == nullptr &&
                          c.d_end == nullptr && (env_fast >= 0 ? env_fast != 0 : qo.search_fast != 0) && nq < 0xffffffffull;
        CursorArgs ca_general = ca;
        if (seed_list != nullptr && !fast) fail(GDX_ERR_INVALID_ARGUMENT, "internal: the seed kernel's list without the fast kernel");
        unsigned g_blocks = blocks;  // grid and range size of the general kernel
        uint32_t g_range = range;
        if (leftover_list) {  // short, and its length is only known on the device: small ranges, a capped grid
            g_range = 256;
            const uint64_t g_ranges = (nq + g_range - 1) / g_range;
            g_blocks = static_cast<unsigned>(g_ranges < 8192 ? g_ranges : 8192);
        }
        // Exact intervals and cursor extension on clean input (search_exact_kernel4): what it cannot finish is listed for
        // the general kernel below.  QueryOptions::search_exact / GDX_SEARCH_EXACT=0 switch it off.
        static const int env_exact = [] { const char *e = getenv("GDX_SEARCH_EXACT"); return e ? atoi(e) : -1; }();
        // (2-bit reads: exact intervals through the text route's kernel and the seed kernel in front of it -- round 6; the
        // general packed kernel steps every symbol on the pair lines of an index without jump table: 4.5 ms per 10 M reads
        // against 0.8)
        // the text route in the jump table's place (search_exact_kernel4<0, ., ., true>); GDX_SEARCH_TEXT=0: pair lines only
        static const int env_text = [] { const char *e = getenv("GDX_SEARCH_TEXT"); return e ? atoi(e) : 1; }();
        const bool text_route = ix.jump == nullptr && ix.sa_full != nullptr && ix.isa != nullptr && ix.text_units != nullptr &&
                                env_text != 0;
        const bool exact = (c.mode == 0 || c.mode == 2) && c.d_step_stats == nullptr && lanes == 4 && policy == 0 &&
                           (!c.packed || (c.mode == 0 && text_route)) &&
                           c.d_hint == nullptr && c.d_rec == nullptr && c.d_start != nullptr && c.d_end != nullptr &&
                           ix.n_searchable >= 4 && ix.sigma >= 5 && ca.resume_state == nullptr &&
                           (ix.top == nullptr || ix.top_depth >= static_cast<uint32_t>(ix.depth)) &&
                           (env_exact >= 0 ? env_exact != 0 : qo.search_exact != 0) && (defer_after == 0u || c.mode == 2) &&
                           nq < 0xffffffffull;
        // Exact intervals on an index with seed table and inverse suffix array: the seed kernel answers the reads that occur
        // exactly once through their seed (interval = the row ISA[position]); the exact kernel then goes over its list.
        static const int env_seed_x = [] { const char *e = getenv("GDX_SEARCH_SEED"); return e ? atoi(e) : -1; }();
        const bool seed_exact = exact && c.mode == 0 && ix.seed != nullptr && ix.isa != nullptr && ix.text_units != nullptr &&
                                ix.seed_k >= static_cast<uint32_t>(ix.depth) && ca.active_in == nullptr &&
                                (env_seed_x >= 0 ? env_seed_x != 0 : qo.search_seed != 0);
        CursorArgs ca_exact = ca;
        unsigned x_blocks = blocks;
        uint32_t x_range = range;
        if (seed_exact) {
            uint32_t *d_first = static_cast<uint32_t *>(stream_scratch(stream, 12, (nq + 4) * sizeof(uint32_t)));
            GDX_HIP(hipMemsetAsync(d_first, 0, sizeof(uint32_t), stream));
            const SeedView sv{ix.seed, ix.text_units, ix.isa, ix.io_to_dense, ix.seed_buckets, ix.seed_k, ix.seed_tag_bits,
                              ix.perm_code_lo, ix.perm_code_hi, ix.perm_exp_lo, ix.perm_exp_hi, ix.perm_mask};
            static const bool env_no_perm_s = getenv("GDX_SEARCH_NO_PERM") != nullptr;
            uint4 *const no_rec = nullptr;
            uint32_t *const none_u32 = nullptr;
            uint32_t *d_long = static_cast<uint32_t *>(stream_scratch(stream, 13, (nq + 4) * sizeof(uint32_t)));
            GDX_HIP(hipMemsetAsync(d_long, 0, sizeof(uint32_t), stream));
            uint2 *const no_state = nullptr;
            const uint64_t t_groups = (nq + kBlock / 4 - 1) / (kBlock / 4);
            const unsigned t_blocks = static_cast<unsigned>(t_groups < 8192 ? t_groups : 8192);
            if (c.packed) {
                hipLaunchKernelGGL((search_seed_kernel4<2, true>), dim3(blocks), dim3(kBlock), 0, stream, sv, c.d_qbuf, c.d_qbeg,
                                   c.d_qend, nq, c.d_count, c.d_status, no_rec, c.d_start, c.d_end, range, d_first + 4, d_first, no_rec,
                                   none_u32, d_long + 4, d_long, no_state, 0u, 0u, ulen, CursorArgs());
                hipLaunchKernelGGL((seed_text_kernel4<2, true>), dim3(t_blocks), dim3(kBlock), 0, stream, sv, c.d_qbuf, c.d_qbeg,
                                   d_long + 4, d_long, no_state, 0u, c.d_count, c.d_status, no_rec, c.d_start, c.d_end, none_u32, no_rec,
                                   d_first + 4, d_first, ulen);
            } else if (ix.perm_ok && !env_no_perm_s) {
                hipLaunchKernelGGL((search_seed_kernel4<1, true>), dim3(blocks), dim3(kBlock), 0, stream, sv, c.d_qbuf, c.d_qbeg,
                                   c.d_qend, nq, c.d_count, c.d_status, no_rec, c.d_start, c.d_end, range, d_first + 4, d_first, no_rec,
                                   none_u32, d_long + 4, d_long, no_state, 0u, 0u, ulen, CursorArgs());
                hipLaunchKernelGGL((seed_text_kernel4<1, true>), dim3(t_blocks), dim3(kBlock), 0, stream, sv, c.d_qbuf, c.d_qbeg,
                                   d_long + 4, d_long, no_state, 0u, c.d_count, c.d_status, no_rec, c.d_start, c.d_end, none_u32, no_rec,
                                   d_first + 4, d_first, ulen);
            } else {
                hipLaunchKernelGGL((search_seed_kernel4<0, true>), dim3(blocks), dim3(kBlock), 0, stream, sv, c.d_qbuf, c.d_qbeg,
                                   c.d_qend, nq, c.d_count, c.d_status, no_rec, c.d_start, c.d_end, range, d_first + 4, d_first, no_rec,
                                   none_u32, d_long + 4, d_long, no_state, 0u, 0u, ulen, CursorArgs());
                hipLaunchKernelGGL((seed_text_kernel4<0, true>), dim3(t_blocks), dim3(kBlock), 0, stream, sv, c.d_qbuf, c.d_qbeg,
                                   d_long + 4, d_long, no_state, 0u, c.d_count, c.d_status, no_rec, c.d_start, c.d_end, none_u32, no_rec,
                                   d_first + 4, d_first, ulen);
            }
            ca_exact.active_in = d_first + 4;
            ca_exact.n_active_in = d_first;
            x_range = 256;
            const uint64_t x_ranges = (nq + x_range - 1) / x_range;
            x_blocks = static_cast<unsigned>(x_ranges < 8192 ? x_ranges : 8192);
        }
        const bool seed_usable = ix.seed != nullptr && ix.seed_k >= static_cast<uint32_t>(ix.depth) && ix.seed_k <= 24u;
        // The first chunk of a batch of cursors on such an index: the seed kernel's pipeline serves the cursors that are still
        // cursor_empty and lists the others for the exact kernel (search_seed_kernel4<., true, true>).  "First" is a guess -- a
        // chunk call with index 0, or a call without a live list: a wrong guess costs a pass, never a result.
        // GDX_SEARCH_SEED_CURSOR=0: the exact kernel alone.
        static const int env_seed_cursor = [] { const char *e = getenv("GDX_SEARCH_SEED_CURSOR"); return e ? atoi(e) : 1; }();
        const bool seed_cursor = exact && c.mode == 2 && text_route && seed_usable && env_seed_cursor != 0 &&
                                 (env_seed_x >= 0 ? env_seed_x != 0 : qo.search_seed != 0) &&
                                 // (chunks longer than an entry covers all go the exact kernel's way: its seed route, then the text)
                                 (ca.chunk_symbols != 0u ? (ca.chunk_index == 0u && ca.chunk_symbols >= ix.seed_k && ca.chunk_symbols <= ix.seed_k + 32u)
                                                         : ca.active_in == nullptr);
        if (seed_cursor) {
            uint32_t *d_first = static_cast<uint32_t *>(stream_scratch(stream, 12, (nq + 4) * sizeof(uint32_t)));
            GDX_HIP(hipMemsetAsync(d_first, 0, sizeof(uint32_t), stream));
            SeedView sv{ix.seed, ix.text_units, ix.isa, ix.io_to_dense, ix.seed_buckets, ix.seed_k, ix.seed_tag_bits,
                        ix.perm_code_lo, ix.perm_code_hi, ix.perm_exp_lo, ix.perm_exp_hi, ix.perm_mask};
            sv.n = ix.n;
            static const bool env_no_perm_c = getenv("GDX_SEARCH_NO_PERM") != nullptr;
            uint4 *const no_rec = nullptr;
            uint32_t *const none_u32 = nullptr;
            uint2 *const no_state = nullptr;
            if (ix.perm_ok && !env_no_perm_c)
                hipLaunchKernelGGL((search_seed_kernel4<1, true, true>), dim3(blocks), dim3(kBlock), 0, stream, sv, c.d_qbuf, c.d_qbeg,
                                   c.d_qend, nq, c.d_count, c.d_status, no_rec, c.d_start, c.d_end, range, d_first + 4, d_first, no_rec,
                                   none_u32, none_u32, none_u32, no_state, 0u, 0u, 0u, ca);
            else
                hipLaunchKernelGGL((search_seed_kernel4<0, true, true>), dim3(blocks), dim3(kBlock), 0, stream, sv, c.d_qbuf, c.d_qbeg,
                                   c.d_qend, nq, c.d_count, c.d_status, no_rec, c.d_start, c.d_end, range, d_first + 4, d_first, no_rec,
                                   none_u32, none_u32, none_u32, no_state, 0u, 0u, 0u, ca);
            ca_exact.active_in = d_first + 4;
            ca_exact.n_active_in = d_first;
            x_range = 256;
            const uint64_t x_ranges = (nq + x_range - 1) / x_range;
            x_blocks = static_cast<unsigned>(x_ranges < 8192 ? x_ranges : 8192);
        }
        if (exact) {
            uint32_t *d_left = static_cast<uint32_t *>(stream_scratch(stream, 11, (nq + 4) * sizeof(uint32_t)));
            GDX_HIP(hipMemsetAsync(d_left, 0, sizeof(uint32_t), stream));
            // (behind the seed kernel's cursor pass the exact kernel does not look into the seed table again)
            const ExactView ev{ix.top, ix.jump, ix.pair_lines, ix.io_to_dense, ix.top_depth, ix.n, static_cast<uint32_t>(ix.depth),
                               ix.perm_code_lo, ix.perm_code_hi, ix.perm_exp_lo, ix.perm_exp_hi, ix.perm_mask,
                               ix.sa_full, ix.isa, ix.text_units, (seed_usable && !seed_cursor) ? ix.seed : nullptr, ix.seed_buckets,
                               ix.seed_k, ix.seed_tag_bits};
            static const bool env_no_perm = getenv("GDX_SEARCH_NO_PERM") != nullptr;  // debug: translate through LDS
            const bool perm = ix.perm_ok && !env_no_perm;
#define GDX_EXACT_LAUNCH(J, XLATE, CURSOR)                                                                                  \
    hipLaunchKernelGGL((search_exact_kernel4<J, XLATE, CURSOR>), dim3(x_blocks), dim3(kBlock), 0, stream, ev, c.d_qbuf,     \
                       c.d_qbeg, c.d_qend, nq, c.d_start, c.d_end, c.d_count, c.d_status, x_range, schedule, d_left + 4, d_left, ca_exact)
#define GDX_EXACT_LAUNCH_X(J, CURSOR)                    \
    do {                                                 \
        if (perm) GDX_EXACT_LAUNCH(J, 1, CURSOR);        \
        else GDX_EXACT_LAUNCH(J, 0, CURSOR);             \
    } while (0)
#define GDX_EXACT_LAUNCH_T(XLATE, CURSOR)                                                                                    \
    hipLaunchKernelGGL((search_exact_kernel4<0, XLATE, CURSOR, true>), dim3(x_blocks), dim3(kBlock), 0, stream, ev, c.d_qbuf, \
                       c.d_qbeg, c.d_qend, nq, c.d_start, c.d_end, c.d_count, c.d_status, x_range, schedule, d_left + 4, d_left, ca_exact)
#define GDX_EXACT_LAUNCH_J(CURSOR)                                                 \
    do {                                                                           \
        if (text_route && c.packed) GDX_EXACT_LAUNCH_T(2, false);                  \
        else if (text_route && perm) GDX_EXACT_LAUNCH_T(1, CURSOR);                \
        else if (text_route) GDX_EXACT_LAUNCH_T(0, CURSOR);                        \
        else if (ix.jump == nullptr) GDX_EXACT_LAUNCH_X(0, CURSOR);                \
        else if (ix.jump_bytes == 32) GDX_EXACT_LAUNCH_X(32, CURSOR);              \
        else if (ix.jump_bytes == 16) GDX_EXACT_LAUNCH_X(16, CURSOR);              \
        else GDX_EXACT_LAUNCH_X(8, CURSOR);                                        \
    } while (0)
            if (c.mode == 2) GDX_EXACT_LAUNCH_J(true);
            else GDX_EXACT_LAUNCH_J(false);
#undef GDX_EXACT_LAUNCH_J
#undef GDX_EXACT_LAUNCH_T
#undef GDX_EXACT_LAUNCH_X
#undef GDX_EXACT_LAUNCH
            g_range = 256;
            const uint64_t g_ranges = (nq + g_range - 1) / g_range;
            g_blocks = static_cast<unsigned>(g_ranges < 8192 ? g_ranges : 8192);
            ca_general.active_in = d_left + 4;  // the general kernel below goes over the leftover list
            ca_general.n_active_in = d_left;
        }
        if (fast) {
            // the leftover list is short (0.3 % of the reads of a non-repetitive text) and its length is only known on
            // the device: small ranges spread it over the chip, a capped grid strides over whatever there is
            g_range = 256;
            const uint64_t g_ranges = (nq + g_range - 1) / g_range;
            g_blocks = static_cast<unsigned>(g_ranges < 8192 ? g_ranges : 8192);
            uint32_t *d_left = static_cast<uint32_t *>(stream_scratch(stream, 11, (nq + 4) * sizeof(uint32_t)));
            GDX_HIP(hipMemsetAsync(d_left, 0, sizeof(uint32_t), stream));
            // where a leftover query stands is kept in its record slot; a call without records (counts only) lets the
            // general kernel start its few leftovers over rather than allocate 16 bytes per query for them
            uint4 *d_state = c.d_rec;
            const FastView fv{ix.top, ix.jump, ix.io_to_dense, ix.top_depth, ix.sa_inv, ix.sa_rot, ix.sa_limit,
                              ix.perm_code_lo, ix.perm_code_hi, ix.perm_exp_lo, ix.perm_exp_hi, ix.perm_mask};
            static const bool env_no_perm = getenv("GDX_SEARCH_NO_PERM") != nullptr;  // debug: translate through LDS
            // sixteen-row rounds where reads from repeats are common (QueryOptions::search_fast == 2)
            const bool wide_rounds = (env_fast >= 0 ? env_fast : qo.search_fast) == 2;
            // (after the seed kernel: over its list -- small ranges, a capped grid, as for the general kernel's lists)
            const unsigned f_blocks = seed_list ? g_blocks : blocks;
            const uint32_t f_range = seed_list ? g_range : range;
            const uint32_t *f_list = seed_list ? seed_list + 4 : nullptr;
#define GDX_FAST_LAUNCH(J, XLATE)                                                                                             \
    do {                                                                                                                      \
        if (wide_rounds)                                                                                                      \
            hipLaunchKernelGGL((search_fast_kernel4<J, XLATE, true>), dim3(f_blocks), dim3(kBlock), 0, stream, fv, c.d_qbuf,  \
                               c.d_qbeg, c.d_qend, nq, c.d_count, c.d_status, c.d_rec, f_range, d_left + 4, d_left, d_state,  \
                               f_list, seed_list, seed_list ? seed_state_packed : 0u); \
        else                                                                                                                  \
            hipLaunchKernelGGL((search_fast_kernel4<J, XLATE, false>), dim3(f_blocks), dim3(kBlock), 0, stream, fv, c.d_qbuf, \
                               c.d_qbeg, c.d_qend, nq, c.d_count, c.d_status, c.d_rec, f_range, d_left + 4, d_left, d_state,  \
                               f_list, seed_list, seed_list ? seed_state_packed : 0u); \
    } while (0)
#define GDX_FAST_LAUNCH_P(XLATE)                                  \
    do {                                                          \
        if (ix.jump_bytes == 32) GDX_FAST_LAUNCH(32, XLATE);      \
        else if (ix.jump_bytes == 16) GDX_FAST_LAUNCH(16, XLATE); \
        else GDX_FAST_LAUNCH(8, XLATE);                           \
    } while (0)
            if (seed_list != nullptr) offsets_for_list(seed_list + 4, seed_list);
            if (c.packed) GDX_FAST_LAUNCH_P(2);
            else if (ix.perm_ok && !env_no_perm) GDX_FAST_LAUNCH_P(1);
            else GDX_FAST_LAUNCH_P(0);
#undef GDX_FAST_LAUNCH_P
#undef GDX_FAST_LAUNCH
            static const bool env_stats = getenv("GDX_SEARCH_FAST_STATS") != nullptr;  // debug: size of the leftover list
            if (env_stats) {
                uint32_t n_left = 0;
                GDX_HIP(hipMemcpyAsync(&n_left, d_left, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                GDX_HIP(hipStreamSynchronize(stream));
                fprintf(stderr, "gdx: fast path left %u of %llu queries to the general kernel\n", n_left,
                        static_cast<unsigned long long>(nq));
            }
            ca_general.active_in = d_left + 4;  // the general kernel below searches the leftover list
            ca_general.n_active_in = d_left;
            ca_general.resume_state = d_state;
        }
        offsets_for_list(ca_general.active_in, ca_general.n_active_in);
        if (c.packed) {
#define GDX_PACKED_W(M)                                                                    \
    do {                                                                                   \
        if (ix.jump_bytes == 32) GDX_PAIR_LAUNCH((search_pair_packed_kernel4<32, M>));     \
        else if (ix.jump_bytes == 16) GDX_PAIR_LAUNCH((search_pair_packed_kernel4<16, M>)); \
        else GDX_PAIR_LAUNCH((search_pair_packed_kernel4<8, M>));                          \
    } while (0)
            if (c.mode == 0) GDX_PACKED_W(0);
            else GDX_PACKED_W(1);
#undef GDX_PACKED_W
        } else if (c.d_step_stats != nullptr) {  // accounting: always the exact mode (the reference's LF steps)
            if (lanes == 8) GDX_PAIR_LAUNCH_W(search_pair_stats_kernel8, 0, 0);
            else GDX_PAIR_LAUNCH_W(search_pair_stats_kernel4, 0, 0);
        } else if (defer_after != 0u && c.mode != 2) {
#define GDX_PAIR_LAUNCH_D(KERNEL)                          \
    do {                                                  \
        if (c.mode == 0) GDX_PAIR_LAUNCH_W(KERNEL, 0, 0); \
        else GDX_PAIR_LAUNCH_W(KERNEL, 0, 1);             \
    } while (0)
            // (6 waves per SIMD / 80 VGPRs halve its 14 spills and change nothing: 14.25 vs 14.36 ms)
            if (lanes == 8) GDX_PAIR_LAUNCH_D(search_pair_defer_kernel8);
            else GDX_PAIR_LAUNCH_D(search_pair_defer_kernel4);
#undef GDX_PAIR_LAUNCH_D
        } else if (lanes == 8) {
            if (policy == 1) GDX_PAIR_LAUNCH_M(search_pair_kernel8, 1);
            else GDX_PAIR_LAUNCH_M(search_pair_kernel8, 0);
        } else {
            if (policy == 0) GDX_PAIR_LAUNCH_M(search_pair_kernel4, 0);
            else GDX_PAIR_LAUNCH_M(search_pair_kernel4, 1);
        }
#undef GDX_PAIR_LAUNCH_M
#undef GDX_PAIR_LAUNCH_W
#undef GDX_PAIR_LAUNCH
        finish_fold();
        return;
    }
    // rank-line and generic kernels: no hints (locate then walks from the interval itself)
    if (c.d_hint) GDX_HIP(hipMemsetAsync(c.d_hint, 0xff, nq * sizeof(uint2), stream));
#define GDX_PLAIN_LAUNCH(TABLE, GROUP, GRID)                                                                        \
    do {                                                                                                            \
        if (c.mode == 2)                                                                                            \
            hipLaunchKernelGGL((search_kernel<TABLE, GROUP, true>), dim3(GRID), dim3(kBlock), lds_pad, stream, ix,  \
                               c.d_qbuf, c.d_qbeg, c.d_qend, nq, c.d_start, c.d_end, c.d_count, c.d_status,         \
                               c.d_step_stats, c.d_rec, ca, ulen);                                                  \
        else if (c.packed)                                                                                          \
            hipLaunchKernelGGL((search_kernel<TABLE, GROUP, false, true>), dim3(GRID), dim3(kBlock), lds_pad, stream, ix, \
                               c.d_qbuf, c.d_qbeg, c.d_qend, nq, c.d_start, c.d_end, c.d_count, c.d_status,         \
                               c.d_step_stats, c.d_rec, ca, ulen);                                                  \
        else                                                                                                        \
            hipLaunchKernelGGL((search_kernel<TABLE, GROUP, false>), dim3(GRID), dim3(kBlock), lds_pad, stream, ix, \
                               c.d_qbuf, c.d_qbeg, c.d_qend, nq, c.d_start, c.d_end, c.d_count, c.d_status,         \
                               c.d_step_stats, c.d_rec, ca, ulen);                                                  \
    } while (0)
    if (ix.layout == 0 && variant != 1) GDX_PLAIN_LAUNCH(QuadLineTable, 4, group_grid(kBlock / 4));
    else if (ix.layout == 0) GDX_PLAIN_LAUNCH(LineTable, 1, grid_for_items(nq));
    else GDX_PLAIN_LAUNCH(GenericTable, 1, grid_for_items(nq));
#undef GDX_PLAIN_LAUNCH
    finish_fold();
}

// ASCII -> 2-bit: packed byte b holds the codes of bytes 4 b .. 4 b + 3 of the query buffer; *bad_symbols counts the
// symbols that are not one of the dense codes 1..4 (their code is written as 0), bad_flags (optional, one byte per
// 64 symbols) marks where they are so that the caller can find the queries they belong to
__global__ __launch_bounds__(kBlock) void pack_queries_kernel(const uint8_t *__restrict__ io_to_dense,
                                                              const uint8_t *__restrict__ qbuf, uint64_t n_symbols,
                                                              uint8_t *__restrict__ packed, uint8_t *__restrict__ bad_flags,
                                                              unsigned long long *__restrict__ bad_symbols)
{
    __shared__ uint8_t s_dense[256];
    for (int i = threadIdx.x; i < 256; i += kBlock) s_dense[i] = io_to_dense[i];
    __syncthreads();
    const uint64_t n_bytes = (n_symbols + 3) / 4;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    uint32_t bad = 0;
    for (uint64_t b = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; b < n_bytes; b += stride) {
        uint32_t out = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint64_t j = 4 * b + k;
            if (j < n_symbols) {
                const uint32_t d = s_dense[qbuf[j]];
                if (d - 1u < 4u) {
                    out |= (d - 1u) << (2u * k);
                } else {
                    bad++;
                    if (bad_flags) bad_flags[j >> 6] = 1;
                }
            }
        }
        packed[b] = static_cast<uint8_t>(out);
    }
    if (bad && bad_symbols) atomicAdd(bad_symbols, static_cast<unsigned long long>(bad));
}

void launch_pack_queries(const IndexView &ix, const uint8_t *d_qbuf, uint64_t n_symbols, uint8_t *d_packed,
                         uint8_t *d_bad_flags, unsigned long long *d_bad_symbols, hipStream_t stream)
{
    if (n_symbols == 0) return;
    hipLaunchKernelGGL(pack_queries_kernel, dim3(grid_for_items((n_symbols + 3) / 4)), dim3(kBlock), 0, stream,
                       ix.io_to_dense, d_qbuf, n_symbols, d_packed, d_bad_flags, d_bad_symbols);
}

void launch_extend_front(const IndexView &ix, uint32_t *d_start, uint32_t *d_end, const uint8_t *d_io_symbols,
                         uint64_t m, uint8_t *d_out_status, hipStream_t stream)
{
    if (m == 0) return;
    if (ix.layout == 0 && ix.pair_lines != nullptr) {
        hipLaunchKernelGGL((extend_front_kernel<QuadLineTable, 8, true>), dim3(grid_for_items(m * 8)), dim3(kBlock), 0,
                           stream, ix, d_start, d_end, d_io_symbols, m, d_out_status);
    } else if (ix.layout == 0) {
        hipLaunchKernelGGL((extend_front_kernel<QuadLineTable, 4, false>), dim3(grid_for_items(m * 4)), dim3(kBlock), 0,
                           stream, ix, d_start, d_end, d_io_symbols, m, d_out_status);
    } else {
        hipLaunchKernelGGL((extend_front_kernel<GenericTable, 1, false>), dim3(grid_for_items(m)), dim3(kBlock), 0,
                           stream, ix, d_start, d_end, d_io_symbols, m, d_out_status);
    }
}

void launch_rank_many(const IndexView &ix, const uint8_t *d_symbols, const uint32_t *d_idx, uint64_t m,
                      uint32_t *d_out, uint32_t *d_error, hipStream_t stream)
{
    if (m == 0) return;
    GDX_DISPATCH_TABLE(ix, rank_many_kernel, grid_for_items(m), stream, ix, d_symbols, d_idx, m, d_out, d_error);
}

void launch_symbol_at_many(const IndexView &ix, const uint32_t *d_idx, uint64_t m, uint8_t *d_out,
                           uint32_t *d_error, hipStream_t stream)
{
    if (m == 0) return;
    GDX_DISPATCH_TABLE(ix, symbol_at_kernel, grid_for_items(m), stream, ix, d_idx, m, d_out, d_error);
}

void launch_lf_walk(const IndexView &ix, const uint32_t *d_rows, uint64_t m, uint32_t steps, uint8_t *d_symbols,
                    uint32_t *d_end_rows, hipStream_t stream)
{
    if (m == 0 || steps == 0) return;
    GDX_DISPATCH_TABLE(ix, lf_walk_kernel, grid_for_items(m), stream, ix, d_rows, m, steps, d_symbols, d_end_rows);
}

void launch_fill_lookup(const IndexView &ix, uint2 *d_lookup, int depth, hipStream_t stream)
{
    uint64_t entries = 1;
    for (int j = 0; j < depth; j++) entries *= static_cast<uint64_t>(ix.n_searchable);
    GDX_DISPATCH_TABLE(ix, fill_lookup_kernel, grid_for_items(entries), stream, ix, d_lookup, depth, entries);
}

// sum of the interval widths of the top-table entries wider than `rows` rows = the number of text positions whose
// D-mer stays wider than a jump can take after the top table (a measure of how repetitive the text is)
__global__ __launch_bounds__(kBlock) void top_wide_kernel(const uint2 *__restrict__ top, uint64_t entries, uint32_t rows,
                                                          unsigned long long *__restrict__ sum)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    unsigned long long mine = 0;
    for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; e < entries; e += stride) {
        const uint2 v = top[e];
        const uint32_t w = v.y - v.x;
        if (w > rows) mine += w;
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(sum, mine);
}

void launch_top_wide(const uint2 *d_top, uint32_t depth, uint32_t rows, unsigned long long *d_sum, hipStream_t stream)
{
    const uint64_t entries = 1ull << (2u * depth);
    hipLaunchKernelGGL(top_wide_kernel, dim3(grid_for_items(entries)), dim3(kBlock), 0, stream, d_top, entries, rows, d_sum);
}

void launch_fill_top(const IndexView &ix, uint2 *d_top, uint32_t depth, hipStream_t stream)
{
    const uint64_t entries = 1ull << (2u * depth);
    hipLaunchKernelGGL(fill_top_kernel, dim3(grid_for_items(entries)), dim3(kBlock), 0, stream, ix, d_top, depth,
                       entries);
}

}  // namespace gdx
