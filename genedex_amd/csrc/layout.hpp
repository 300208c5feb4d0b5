// layout.hpp -- device-side view of the FM index and the rank / LF primitives.
//
// Data layout in HBM:
//
//  * rank lines (sigma <= 8, the DNA case): one 64-byte line per 128 BWT positions.  A line is
//    four 16-byte chunks; chunk j holds, for positions [32j, 32j+32) of the line, the three
//    bit planes as 32-bit words (x, y, z = plane 0, 1, 2; bit t <-> position 32j+t) and in w
//    two u16 block offsets: symbol 2j in the low half, symbol 2j+1 in the high half
//    (= number of occurrences of that symbol in the enclosing 65536-position superblock before
//    this line).  So one rank touches exactly one aligned 64-byte line plus one u32 of the
//    small superblock table (32 B per superblock, L2 resident).
//    Logical content is the reference's CondensedTextWithRankSupport<I, Block64>
//    (condensed.rs:24-30) with two 64-bit blocks fused per line.
//
//  * generic planes (sigma > 8): the reference's logical layout itself: interleaved 64-bit
//    plane words, u16 block offsets per 64 positions, u32 superblock offsets.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace gdx {

// native clang vector: HIP's uint4 is a struct-with-union that defeats SROA and sends the
// 64-byte line to scratch / LDS
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));  // four dwords at a dword-aligned address

// lookup tables of all depths 0..=d are built (lookup_table.rs:163-181); the reference computes indices with
// const-curried code up to depth 15 and a dynamic loop beyond (lookup_table.rs:68-113) -- here every depth is the loop.
// The limit is memory: the deepest table alone has n_searchable^d entries of 8 bytes (4^16 = 34 GB for DNA).
constexpr int kMaxLookupDepth = 24;
constexpr uint32_t kSuperblockShift = 16;  // 65536 positions (condensed.rs:34)
constexpr uint32_t kLineShift = 7;         // 128 positions per rank line
constexpr uint32_t kLinesPerSuperblock = 1u << (kSuperblockShift - kLineShift);

struct IndexView {
    // --- occurrence table ------------------------------------------------------------
    const u32x4 *lines;           // [n_lines][4]               (layout 0)
    const uint32_t *sb_offsets;   // [n_superblocks][sb_stride] absolute counts before the superblock
    const uint64_t *g_planes;     // generic: [n_blocks64][nbits]       (layout 1)
    const uint16_t *g_block_off;  // generic: [n_blocks64][sigma]
    uint32_t sb_stride;           // 8 for layout 0, sigma for layout 1
    // layout 1 is the reference's own occurrence table, any of its four variants (gdx_build_options_t.reference_table_layout;
    // the default for sigma > 8 is Condensed / Block64):
    //   g_kind 0 = condensed (condensed.rs:24-47): plane b of block k = words [(k * nbits + b) * g_wpb, + g_wpb) of g_planes,
    //              u16 block offsets g_block_off[k * sigma + c], u32 superblock offsets every 65536 positions
    //   g_kind 1 = flat (flat.rs:30-52): block of symbol c = words [(k * sigma + c) * g_wpb, + g_wpb): 16 bits of block offset,
    //              then g_used = 64 g_wpb - 16 indicator bits; superblocks of g_sb = (65536 / g_used) * g_used positions
    uint32_t g_kind, g_wpb;       // g_wpb: 64-bit words per block, 1 (Block64) or 8 (Block512)
    uint32_t g_used, g_sb;        // positions per block / per superblock
    // --- pair lines: one or two LF steps per 128-byte fetch (rank-line layout only) -------------
    const u32x4 *pair_lines;      // [ceil((n+1)/64)][8], null when absent
    // --- jump table: 8 .. 32 LF steps of a narrow interval per fetch -----------------------------------
    // [n] entries of jump_bytes.  Level j of entry i = {t_j = row after 8j LF steps from row i, c_j = 2-bit codes
    // (dense symbol - 1, bits 15:14 = the first of them) of the symbols of steps 8j-7 .. 8j, valid bit j-1 = these
    // and all earlier symbols are in 1..4}; a code without its target is a lookahead: it tells whether a row
    // survives further steps without saying where it goes.  As 32-bit words:
    //    8 bytes: {t1, c1 | valid << 16}                                   1 level
    //   16 bytes: {t1, t2, c1 | c2 << 16, c3 | valid << 16}                 2 levels + lookahead c3
    //   32 bytes: the same + {t3, t4, SA[i], c4 | c5 << 16}                 4 levels + lookahead c5 (valid bit 4 = c5
    //             is usable) and the suffix-array value of the row itself: a search that narrows to one row while
    //             it holds that row's entry knows the hit's text position (SA[i] - symbols still to match) without
    //             any locate walk, and locate resolves any other row with one fetch instead of a walk to a sample
    //             (round 2 stored a fifth target there; the sample read was then one DRAM request per hit and two
    //             thirds of locate's traffic)
    // Bits 8..11 of the valid field: how many leading symbols of c1 are real (8 when valid bit 0 is set); an entry
    // whose first level is cut short by a sentinel or an N still tells whether its row survives that many steps.
    const void *jump;             // null when absent
    uint32_t jump_bytes;          // 0, 8, 16 or 32
    // --- top table: interval after the first top_depth symbols of a DNA query, one cache-resident fetch ----
    const uint2 *top;             // [4^top_depth], index = 2-bit codes (first consumed symbol highest); the entry of
                                  // an absent D-mer is its frozen empty interval; null when absent
    uint32_t top_depth;           // 1..16
    // --- full suffix array and the text itself (optional; the low-memory alternative to the jump table) ----------------
    const uint32_t *sa_full;      // [n] SA[row], null when absent (32-byte jump entries carry the same value in word 6)
    // Text units: the concatenated text (sentinels included), 32 symbols per 16-byte unit {codes lo, codes hi, mask, 0}:
    // 2-bit code (dense - 1) of symbol 32 u + i in bits 2 i + 1 : 2 i of the 64-bit code string, mask bit i set when that
    // symbol is not one of the dense codes 1..4 (sentinel, N, ...; its code is 0).  kTextPadUnits zero units (mask bits
    // set) precede unit 0, so that a window that starts before the text reads as "no match".  A search that is down to a
    // few rows compares the rest of the query with the text at SA[row] -- 32 symbols per 64-bit compare, one or two
    // fetches per row whatever the length -- instead of walking LF steps (search_verify_kernel4).
    const u32x4 *text_units;      // null when absent
    // --- seed table: the LAST seed_k symbols of a count / locate query in one 128-byte fetch, and for most reads the
    // whole answer (optional, needs the text units) -------------------------------------------------------------------------
    // A bucketed hash table over the distinct seed_k-mers (A C G T only) of the text: seed_buckets buckets of eight 16-byte
    // entries.  key = the k-mer in text order, 2 bits per symbol, first symbol lowest; tag = its low seed_tag_bits bits,
    // a = key >> seed_tag_bits (< seed_buckets by construction), home bucket = (a + umulhi(tag * 0x9E3779B1, buckets))
    // mod buckets -- for a fixed tag the map a -> bucket is one to one, so (bucket, tag) names the k-mer exactly and
    // an absent k-mer is known to be absent.  An entry that found its home bucket full sits `disp` buckets further
    // (linear probing over buckets); bit 31 of every entry of a bucket says that some entry was turned away there.
    //   word 0: tag [20:0] | disp [25:21] (31 = empty slot) | kind [26] | partial [27] | v code [30:29] | overflow [31]
    //   kind 0 (the k-mer occurs once):
    //           {w0, SA of its row, codes of the 32 symbols in front of that position (as a text unit: first lowest)}
    //           -- a read of up to seed_k + 32 symbols is decided by this entry alone, position included; a longer one
    //           goes on comparing with the text units.  `partial`: only the v < 32 symbols right in front are A C G T
    //           of the same text (then comes a sentinel, an N, ...); v sits in the low six bits of the codes (bits 30:29
    //           of word 0: 1 = v is 30, 2 = 31, which leave no room there), and a read that reaches further back than v
    //           symbols does not occur
    //   kind 1 (several rows):
    //           {w0, lo, hi, 0} = the k-mer's suffix-array interval; the search goes on from there as after a top table.
    //           `partial` set (kSeedPairInfo): the interval has exactly two rows, both with 32 symbols A C G T of their text in
    //           front, and word 3 is the index of their 32-byte record in seed_pairs: {SA[lo], SA[lo + 1], codes of the 32
    //           symbols in front of the first, of the second (as in a kind-0 entry), 0, 0} -- a count / locate read of up to
    //           seed_k + 32 symbols from a two-copy repeat is decided by that record: no suffix-array line, no text lines
    const u32x4 *seed;            // null when absent
    const u32x4 *seed_pairs;      // two u32x4 per record; null when absent (no room in the budget, or no such k-mers)
    // the same for k-mers on THREE or FOUR rows (kSeedQuadInfo, word 3 = the index): 64-byte records {SA[lo .. lo + 3]} {contexts of
    // rows 0, 1} {contexts of rows 2, 3} {lo, rows, 0, 0} -- every row with 32 symbols A C G T in front; a read that ends on three
    // or four of them leaves as the masked record of those rows, which is why lo is in the record
    const u32x4 *seed_quads;      // four u32x4 per record; null when absent
    // inverse suffix array (optional): isa[p] = the row whose suffix starts at text position p.  With it the seed table
    // also answers EXACT intervals: a read that occurs once, at position p, has the interval [isa[p], isa[p] + 1)
    const uint32_t *isa;          // [n], null when absent
    uint32_t seed_buckets;
    uint32_t seed_k;              // 8..32
    uint32_t seed_tag_bits;       // 0..21
    // --- C array, alphabet -------------------------------------------------------------
    const uint32_t *count;        // [sigma+1]  (lib.rs:95)
    const uint8_t *io_to_dense;   // [256]      (alphabet.rs:24-28)
    // the same table for the four searchable symbols 1..4 in a form v_perm_b32 can apply to four bytes at once
    // (make_perm_translation, fm_index.hip): byte c is one of them exactly when (c & perm_mask) == exp[c & 7], and its
    // dense code is then code[c & 7] + 1; exp / code are 8-byte tables held in two registers each.  perm_ok = 0: the
    // alphabet has no such table (two searchable symbols share their low three bits) and kernels use io_to_dense.
    uint32_t perm_code_lo, perm_code_hi, perm_exp_lo, perm_exp_hi, perm_mask;
    int32_t perm_ok;
    // --- sampled suffix array ------------------------------------------------------------
    const uint32_t *sa_samples;   // SA[i] for i % sa_rate == 0
    const uint32_t *border_keys;  // sorted SA indices whose BWT symbol is the sentinel
    const uint32_t *border_vals;  // SA value there (= start of a text)
    const uint32_t *sentinels;    // sentinel_indices (text_id_search_tree.rs:8)
    // --- lookup tables -------------------------------------------------------------------
    const uint2 *lookup;          // all depths 0..depth concatenated; table t starts at lookup_offset(k, t)
    // --- scalars ---------------------------------------------------------------------------
    uint32_t n;                   // total text length incl. sentinels
    uint32_t n_texts;
    uint32_t sa_rate;
    // division-free "is row i sampled, and which sample": sa_rate = 2^sa_rot * d with d odd, sa_inv = d^-1 mod 2^32,
    // sa_limit = (2^32 - 1) / sa_rate; see sampled_slot()
    uint32_t sa_inv, sa_rot, sa_limit;
    int32_t sigma;
    int32_t nbits;
    int32_t n_searchable;
    int32_t depth;
    int32_t layout;
};

// seed table entries (IndexView::seed)
constexpr uint32_t kSeedTagBitsMax = 21;
constexpr uint32_t kSeedDispShift = 21;
constexpr uint32_t kSeedMaxDisp = 30;                       // 31 marks an empty slot
constexpr uint32_t kSeedEmpty = 31u << kSeedDispShift;
constexpr uint32_t kSeedMatchMask = 0x03ffffffu;            // tag and disp
constexpr uint32_t kSeedKind = 1u << 26;
constexpr uint32_t kSeedPartial = 1u << 27;
constexpr uint32_t kSeedPairInfo = kSeedPartial;           // in an entry of kind 1: see IndexView::seed_pairs
constexpr uint32_t kSeedQuadInfo = 1u << 29;                // in an entry of kind 1 (bit 29, a kind-0 entry's v code): see IndexView::seed_quads
constexpr uint32_t kSeedPartialShift = 29;                  // bits 30:29 of a partial entry: 0 = v in the codes, 1 = 30, 2 = 31
constexpr uint32_t kSeedFound = 1u << 28;                   // never set in the table: marks a matched entry in registers
constexpr uint32_t kSeedOverflow = 1u << 31;

__device__ __forceinline__ uint32_t seed_home(uint64_t key, uint32_t tag_bits, uint32_t buckets, uint32_t &tag)
{
    tag = static_cast<uint32_t>(key) & ((1u << tag_bits) - 1u);
    const uint32_t a = static_cast<uint32_t>(key >> tag_bits);
    uint32_t b = a + __umulhi(tag * 0x9E3779B1u, buckets);  // both < buckets
    if (b >= buckets) b -= buckets;
    return b;
}

// first entry of lookup table t among the concatenated tables: sum of k^j for j < t (k = number of searchable symbols)
__host__ __device__ __forceinline__ uint64_t lookup_offset(uint32_t k, uint32_t t)
{
    uint64_t off = 0, pw = 1;
    for (uint32_t j = 0; j < t; j++) {
        off += pw;
        pw *= k;
    }
    return off;
}

// ---------------------------------------------------------------------------------------
// rank lines (layout 0)

struct RankLine {
    u32x4 c0, c1, c2, c3;
};

__device__ __forceinline__ RankLine load_line(const IndexView &ix, uint32_t idx)
{
    const u32x4 *p = ix.lines + (static_cast<uint64_t>(idx >> kLineShift) << 2);
    RankLine l;
    l.c0 = p[0];
    l.c1 = p[1];
    l.c2 = p[2];
    l.c3 = p[3];
    return l;
}

// matches of the symbol among the first `bits` (may be <= 0 or >= 32) positions of one chunk
__device__ __forceinline__ uint32_t chunk_popcount(const u32x4 c, uint32_t n0, uint32_t n1, uint32_t n2,
                                                   int32_t bits)
{
    // block.rs:152-178 negate / set_to_self_and / count_ones_before, on 32 positions
    const uint32_t m = (c.x ^ n0) & (c.y ^ n1) & (c.z ^ n2);
    const uint32_t mask = bits >= 32 ? 0xffffffffu : (bits <= 0 ? 0u : ((1u << bits) - 1u));
    return __popc(m & mask);
}

// occurrences of `symbol` among the first `within` (0..127) positions of the line
__device__ __forceinline__ uint32_t line_popcount(const RankLine &l, uint32_t symbol, uint32_t within)
{
    const uint32_t n0 = (symbol & 1u) ? 0u : 0xffffffffu;
    const uint32_t n1 = (symbol & 2u) ? 0u : 0xffffffffu;
    const uint32_t n2 = (symbol & 4u) ? 0u : 0xffffffffu;
    const int32_t w = static_cast<int32_t>(within);
    return chunk_popcount(l.c0, n0, n1, n2, w) + chunk_popcount(l.c1, n0, n1, n2, w - 32) +
           chunk_popcount(l.c2, n0, n1, n2, w - 64) + chunk_popcount(l.c3, n0, n1, n2, w - 96);
}

__device__ __forceinline__ uint32_t line_block_offset(const RankLine &l, uint32_t symbol)
{
    const uint32_t j = symbol >> 1;
    uint32_t w = l.c0.w;
    w = (j == 1) ? l.c1.w : w;
    w = (j == 2) ? l.c2.w : w;
    w = (j == 3) ? l.c3.w : w;
    return (symbol & 1u) ? (w >> 16) : (w & 0xffffu);
}

__device__ __forceinline__ uint32_t line_symbol_at(const RankLine &l, uint32_t within)
{
    const uint32_t j = within >> 5, t = within & 31u;
    uint32_t x = l.c0.x, y = l.c0.y, z = l.c0.z;
    x = (j == 1) ? l.c1.x : x;
    y = (j == 1) ? l.c1.y : y;
    z = (j == 1) ? l.c1.z : z;
    x = (j == 2) ? l.c2.x : x;
    y = (j == 2) ? l.c2.y : y;
    z = (j == 2) ? l.c2.z : z;
    x = (j == 3) ? l.c3.x : x;
    y = (j == 3) ? l.c3.y : y;
    z = (j == 3) ? l.c3.z : z;
    return ((x >> t) & 1u) | (((y >> t) & 1u) << 1) | (((z >> t) & 1u) << 2);
}

struct LineTable {
    // rank(symbol, idx) = #symbol in bwt[0..idx)   (condensed.rs:291-341)
    static __device__ __forceinline__ uint32_t rank(const IndexView &ix, uint32_t symbol, uint32_t idx)
    {
        const uint32_t sb = ix.sb_offsets[(idx >> kSuperblockShift) * 8u + symbol];
        const RankLine l = load_line(ix, idx);
        return sb + line_block_offset(l, symbol) + line_popcount(l, symbol, idx & 127u);
    }
    // both borders of one interval; loads are issued before any use
    static __device__ __forceinline__ void rank2(const IndexView &ix, uint32_t symbol, uint32_t lo, uint32_t hi,
                                                 uint32_t &rlo, uint32_t &rhi)
    {
        const uint32_t sb_lo = ix.sb_offsets[(lo >> kSuperblockShift) * 8u + symbol];
        const uint32_t sb_hi = ix.sb_offsets[(hi >> kSuperblockShift) * 8u + symbol];
        const RankLine l_lo = load_line(ix, lo);
        const RankLine l_hi = load_line(ix, hi);
        rlo = sb_lo + line_block_offset(l_lo, symbol) + line_popcount(l_lo, symbol, lo & 127u);
        rhi = sb_hi + line_block_offset(l_hi, symbol) + line_popcount(l_hi, symbol, hi & 127u);
    }
    static __device__ __forceinline__ uint32_t symbol_at(const IndexView &ix, uint32_t idx)
    {
        return line_symbol_at(load_line(ix, idx), idx & 127u);
    }
    // symbol_at(idx) and, unless it is the sentinel, rank(symbol, idx) from the same line
    static __device__ __forceinline__ uint32_t symbol_and_rank(const IndexView &ix, uint32_t idx, uint32_t &rank_out)
    {
        const RankLine l = load_line(ix, idx);
        const uint32_t c = line_symbol_at(l, idx & 127u);
        const uint32_t sb = ix.sb_offsets[(idx >> kSuperblockShift) * 8u + c];
        rank_out = sb + line_block_offset(l, c) + line_popcount(l, c, idx & 127u);
        return c;
    }
};

// ---------------------------------------------------------------------------------------
// rank lines, four lanes per query: the quad fetches one 64-byte line with ONE coalesced 16-byte load
// per lane (lane j owns chunk j), counts its own 32 positions and the quad sums with two DPP adds.
// One L2 request per line instead of four, and no duplicate DRAM fetches of a line whose first miss
// is still in flight.

__device__ __forceinline__ uint32_t quad_sum(uint32_t v)
{
    // quad_perm [1,0,3,2] then [2,3,0,1]: every lane of the quad ends up with the quad's total
    v += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0xB1, 0xF, 0xF, true));
    v += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x4E, 0xF, 0xF, true));
    return v;
}

struct QuadLineTable {
    // contribution of this lane's chunk to rank-within-superblock of (symbol, idx)
    static __device__ __forceinline__ uint32_t partial(const u32x4 c, uint32_t sub, uint32_t symbol, uint32_t idx)
    {
        const uint32_t n0 = (symbol & 1u) ? 0u : 0xffffffffu;
        const uint32_t n1 = (symbol & 2u) ? 0u : 0xffffffffu;
        const uint32_t n2 = (symbol & 4u) ? 0u : 0xffffffffu;
        const uint32_t pop = chunk_popcount(c, n0, n1, n2, static_cast<int32_t>(idx & 127u) - 32 * static_cast<int32_t>(sub));
        const uint32_t off = (symbol & 1u) ? (c.w >> 16) : (c.w & 0xffffu);
        return pop + ((symbol >> 1) == sub ? off : 0u);
    }
    static __device__ __forceinline__ void rank2(const IndexView &ix, uint32_t symbol, uint32_t lo, uint32_t hi,
                                                 uint32_t &rlo, uint32_t &rhi)
    {
        const uint32_t sub = threadIdx.x & 3u;
        const uint32_t sb_lo = ix.sb_offsets[(lo >> kSuperblockShift) * 8u + symbol];
        const uint32_t sb_hi = ix.sb_offsets[(hi >> kSuperblockShift) * 8u + symbol];
        const uint32_t line_lo = lo >> kLineShift, line_hi = hi >> kLineShift;
        const u32x4 a = ix.lines[(static_cast<uint64_t>(line_lo) << 2) + sub];
        u32x4 b = a;
        if (line_hi != line_lo) b = ix.lines[(static_cast<uint64_t>(line_hi) << 2) + sub];  // quad-uniform
        rlo = sb_lo + quad_sum(partial(a, sub, symbol, lo));
        rhi = sb_hi + quad_sum(partial(b, sub, symbol, hi));
    }
};

// ---------------------------------------------------------------------------------------
// pair lines: one 128-byte line per 64 BWT positions holds the bit planes of BOTH preceding symbols
// (bwt1[i] = text[SA[i]-1], bwt0[i] = text[SA[i]-2]), the ABSOLUTE number of occurrences of each of the
// 16 pairs of symbols 1..4 before the line -- with C2 already added -- and the same for the 4 single
// symbols, so that
//     two LF steps  LF(c2, LF(c1, i)) = pair[c2 c1] + #{ j in line, j < i : bwt0[j] = c2, bwt1[j] = c1 }
//     one LF step   LF(c1, i)         = single[c1]  + #{ j in line, j < i : bwt1[j] = c1 }
// each cost exactly one 128-byte fetch and nothing else (no superblock table).  Every DRAM request of
// this GPU is 128 bytes wide whatever the load asked for (measured: TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ
// for 64-byte gathers), so the line is the natural unit, and the search of a read becomes ~len/2 + 6
// line fetches.  8 bits per symbol; symbols outside 1..4 (sentinel, N) go through the rank lines.
// Chunk j (16 B, one lane of an 8-lane group) covers positions [8j, 8j+8) of the line:
//   x = b1p0 | b1p1 << 8 | b1p2 << 16 | b0p0 << 24      (b1pK / b0pK: plane K of bwt1 / bwt0, 8 positions)
//   y = b0p1 | b0p2 << 8 | (16 bits of single[1 + j/2], low half if j is even) << 16
//   z = pair[2j], w = pair[2j+1]                          (pair index = (c2-1)*4 + (c1-1))

constexpr uint32_t kTextPadUnits = 2;   // zero units in front of the text units (IndexView::text_units)
constexpr uint32_t kPairLineShift = 6;  // 64 positions per pair line
constexpr uint32_t kJumpSymbols = 8;    // LF steps folded into one jump-table entry

__device__ __forceinline__ uint32_t oct_sum(uint32_t v)
{
    v = quad_sum(v);
    // row_half_mirror: lane i <-> lane 7-i inside each group of 8, i.e. the other quad's total
    v += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x141, 0xF, 0xF, true));
    return v;
}

// minimum / maximum over the kLanes (4 or 8) lanes of a group; every lane gets the result
template <int kLanes>
__device__ __forceinline__ uint32_t group_min(uint32_t v)
{
    uint32_t o = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0xB1, 0xF, 0xF, true));
    v = o < v ? o : v;
    o = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x4E, 0xF, 0xF, true));
    v = o < v ? o : v;
    if (kLanes == 8) {
        o = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x141, 0xF, 0xF, true));
        v = o < v ? o : v;
    }
    return v;
}
template <int kLanes>
__device__ __forceinline__ uint32_t group_max(uint32_t v)
{
    uint32_t o = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0xB1, 0xF, 0xF, true));
    v = o > v ? o : v;
    o = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x4E, 0xF, 0xF, true));
    v = o > v ? o : v;
    if (kLanes == 8) {
        o = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x141, 0xF, 0xF, true));
        v = o > v ? o : v;
    }
    return v;
}

// One round of line / entry loads as ONE asm statement: every load, the exec masking of the optional ones and the
// single wait sit between the same pair of braces, so no compiler-generated instruction (a register copy, a scratch
// spill of a destination that is still in flight) can land between issue and wait -- the compiler's own waitcnt
// bookkeeping does not see loads issued from inline asm.  The destinations are early-clobber outputs: they are
// defined by the statement and complete when it ends; lanes outside a mask get undefined values.
// kPolicy: 0 = plain loads, 1 = sc1 (served by L2, no allocation in the CU's L1).
// two lines of one chunk per lane (8 lanes per line): a always, b only in the lanes of m_b
template <int kPolicy>
__device__ __forceinline__ void load_round2(const u32x4 *pa, const u32x4 *pb, unsigned long long m_b, u32x4 &a, u32x4 &b)
{
    unsigned long long saved;
    if (kPolicy == 1)
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
                     "s_and_saveexec_b64 %2, %5\n\t"
                     "global_load_dwordx4 %1, %4, off sc1\n\t"
                     "s_mov_b64 exec, %2\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&s"(saved)
                     : "v"(pa), "v"(pb), "s"(m_b)
                     : "memory", "scc");
    else
        asm volatile("global_load_dwordx4 %0, %3, off\n\t"
                     "s_and_saveexec_b64 %2, %5\n\t"
                     "global_load_dwordx4 %1, %4, off\n\t"
                     "s_mov_b64 exec, %2\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&s"(saved)
                     : "v"(pa), "v"(pb), "s"(m_b)
                     : "memory", "scc");
}

// two lines of two chunks per lane (4 lanes per line), or a jump entry in a0 (+ a1): a0 always, a1 in the lanes of
// m_a1, b0 and b1 in the lanes of m_b
template <int kPolicy>
__device__ __forceinline__ void load_round4(const u32x4 *pa0, const u32x4 *pa1, const u32x4 *pb0, const u32x4 *pb1,
                                            unsigned long long m_a1, unsigned long long m_b, u32x4 &a0, u32x4 &a1,
                                            u32x4 &b0, u32x4 &b1)
{
    unsigned long long saved;
    if (kPolicy == 1)
        asm volatile("global_load_dwordx4 %0, %5, off sc1\n\t"
                     "s_mov_b64 %4, exec\n\t"
                     "s_and_b64 exec, %4, %9\n\t"
                     "global_load_dwordx4 %1, %6, off sc1\n\t"
                     "s_and_b64 exec, %4, %10\n\t"
                     "global_load_dwordx4 %2, %7, off sc1\n\t"
                     "global_load_dwordx4 %3, %8, off sc1\n\t"
                     "s_mov_b64 exec, %4\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&s"(saved)
                     : "v"(pa0), "v"(pa1), "v"(pb0), "v"(pb1), "s"(m_a1), "s"(m_b)
                     : "memory", "scc");
    else
        asm volatile("global_load_dwordx4 %0, %5, off\n\t"
                     "s_mov_b64 %4, exec\n\t"
                     "s_and_b64 exec, %4, %9\n\t"
                     "global_load_dwordx4 %1, %6, off\n\t"
                     "s_and_b64 exec, %4, %10\n\t"
                     "global_load_dwordx4 %2, %7, off\n\t"
                     "global_load_dwordx4 %3, %8, off\n\t"
                     "s_mov_b64 exec, %4\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&s"(saved)
                     : "v"(pa0), "v"(pa1), "v"(pb0), "v"(pb1), "s"(m_a1), "s"(m_b)
                     : "memory", "scc");
}

// both border lines of an interval for a group of 8 / kChunks lanes; b = a when the borders share a line
template <int kPolicy, int kChunks>
__device__ __forceinline__ void load_two_lines(const u32x4 *pa, const u32x4 *pb, bool second, int step,
                                               u32x4 (&a)[kChunks], u32x4 (&b)[kChunks])
{
    const unsigned long long m_b = __ballot(second);
    if (kChunks == 1) {
        load_round2<kPolicy>(pa, pb, m_b, a[0], b[0]);
    } else {
        const unsigned long long all = __ballot(true);
        load_round4<kPolicy>(pa, pa + step, pb, pb + step, all, m_b, a[0], a[kChunks - 1], b[0], b[kChunks - 1]);
    }
    if (!second) {
#pragma unroll
        for (int k = 0; k < kChunks; k++) b[k] = a[k];
    }
}

// kLanes = 8: one chunk per lane; kLanes = 4: two chunks per lane (sub and sub + 4), twice the queries per wave
template <int kLanes>
__device__ __forceinline__ uint32_t group_sum(uint32_t v)
{
    return kLanes == 8 ? oct_sum(v) : quad_sum(v);
}

struct PairTable {
    static __device__ __forceinline__ uint32_t low_mask(uint32_t idx, uint32_t chunk)
    {
        int32_t bits = static_cast<int32_t>(idx & 63u) - 8 * static_cast<int32_t>(chunk);
        bits = bits < 0 ? 0 : (bits > 8 ? 8 : bits);  // v_med3_i32
        return (1u << bits) - 1u;
    }
    static __device__ __forceinline__ uint32_t pair_partial(const u32x4 c, uint32_t chunk, uint32_t pair, uint32_t nx,
                                                            uint32_t ny, uint32_t idx)
    {
        const uint32_t tx = c.x ^ nx, ty = c.y ^ ny;
        const uint32_t m = tx & (tx >> 8) & (tx >> 16) & (tx >> 24) & ty & (ty >> 8) & 0xffu;
        const uint32_t cnt = (pair & 1u) ? c.w : c.z;
        return __popc(m & low_mask(idx, chunk)) + ((pair >> 1) == chunk ? cnt : 0u);
    }
    static __device__ __forceinline__ uint32_t single_partial(const u32x4 c, uint32_t chunk, uint32_t c1, uint32_t nx,
                                                              uint32_t idx)
    {
        const uint32_t tx = c.x ^ nx;
        const uint32_t m = tx & (tx >> 8) & (tx >> 16) & 0xffu;
        const bool owner = (chunk >> 1) == (c1 - 1u);
        return __popc(m & low_mask(idx, chunk)) + (owner ? ((c.y >> 16) << ((chunk & 1u) * 16u)) : 0u);
    }
    // one LF step with a symbol in 1..4
    template <int kPolicy, int kLanes>
    static __device__ __forceinline__ void lf1(const IndexView &ix, uint32_t c1, uint32_t lo, uint32_t hi,
                                               uint32_t &nlo, uint32_t &nhi)
    {
        constexpr int kChunks = 8 / kLanes;
        const uint32_t sub = threadIdx.x & (kLanes - 1u);
        const uint32_t bits_x = (c1 & 1u) | ((c1 & 2u) << 7) | ((c1 & 4u) << 14);
        const uint32_t nx = ~(bits_x * 0xffu) & 0xffffffu;
        const uint32_t line_lo = lo >> kPairLineShift, line_hi = hi >> kPairLineShift;
        u32x4 a[kChunks], b[kChunks];
        load_two_lines<kPolicy, kChunks>(ix.pair_lines + (static_cast<uint64_t>(line_lo) << 3) + sub,
                                         ix.pair_lines + (static_cast<uint64_t>(line_hi) << 3) + sub,
                                         line_hi != line_lo, kLanes, a, b);
        uint32_t plo = 0, phi = 0;
#pragma unroll
        for (int k = 0; k < kChunks; k++) {
            plo += single_partial(a[k], sub + k * kLanes, c1, nx, lo);
            phi += single_partial(b[k], sub + k * kLanes, c1, nx, hi);
        }
        nlo = group_sum<kLanes>(plo);
        nhi = group_sum<kLanes>(phi);
    }
};

// ---------------------------------------------------------------------------------------
// generic planes (layout 1): the reference's own three arrays

struct GenericTable {
    // rank(symbol, idx) over the reference's own arrays (condensed.rs:291-341, flat.rs:221-246); Condensed / Block64 -- what
    // alphabets beyond eight symbols get by default -- keeps its own short path
    static __device__ __forceinline__ uint32_t rank(const IndexView &ix, uint32_t symbol, uint32_t idx)
    {
        if (ix.g_kind == 0u && ix.g_wpb == 1u) {
            const uint32_t blk = idx >> 6;
            const uint32_t sb = ix.sb_offsets[(idx >> kSuperblockShift) * ix.sb_stride + symbol];
            const uint32_t bo = ix.g_block_off[static_cast<uint64_t>(blk) * ix.sigma + symbol];
            const uint64_t *planes = ix.g_planes + static_cast<uint64_t>(blk) * ix.nbits;
            uint64_t acc = ~0ull;
            uint32_t s = symbol;
            for (int b = 0; b < ix.nbits; b++) {
                uint64_t p = planes[b];
                acc &= (s & 1u) ? p : ~p;
                s >>= 1;
            }
            const uint32_t t = idx & 63u;
            const uint64_t mask = t ? (~0ull >> (64u - t)) : 0ull;
            return sb + bo + __popcll(acc & mask);
        }
        const uint32_t blk = idx / ix.g_used, t = idx - blk * ix.g_used;  // t positions of the block count
        uint32_t r = ix.sb_offsets[(idx / ix.g_sb) * ix.sb_stride + symbol];
        if (ix.g_kind == 0u) {
            r += ix.g_block_off[static_cast<uint64_t>(blk) * ix.sigma + symbol];
            const uint64_t *planes = ix.g_planes + static_cast<uint64_t>(blk) * ix.nbits * ix.g_wpb;
            for (uint32_t w = 0; w * 64u < t; w++) {
                uint64_t acc = ~0ull;
                uint32_t s = symbol;
                for (int b = 0; b < ix.nbits; b++) {
                    const uint64_t p = planes[static_cast<uint32_t>(b) * ix.g_wpb + w];
                    acc &= (s & 1u) ? p : ~p;
                    s >>= 1;
                }
                const uint32_t left = t - w * 64u;
                r += __popcll(left >= 64u ? acc : acc & (~0ull >> (64u - left)));
            }
        } else {
            const uint64_t *blkw = ix.g_planes + (static_cast<uint64_t>(blk) * ix.sigma + symbol) * ix.g_wpb;
            r += static_cast<uint32_t>(blkw[0] & 0xffffull);  // the block offset lives in the block's first 16 bits
            const uint32_t end = t + 16u;                      // indicator bits 16 .. end - 1 count
            for (uint32_t w = 0; w * 64u < end; w++) {
                uint64_t v = blkw[w];
                if (w == 0u) v &= ~0xffffull;
                const uint32_t left = end - w * 64u;
                r += __popcll(left >= 64u ? v : v & (~0ull >> (64u - left)));
            }
        }
        return r;
    }
    static __device__ __forceinline__ void rank2(const IndexView &ix, uint32_t symbol, uint32_t lo, uint32_t hi,
                                                 uint32_t &rlo, uint32_t &rhi)
    {
        rlo = rank(ix, symbol, lo);
        rhi = rank(ix, symbol, hi);
    }
    static __device__ __forceinline__ uint32_t symbol_at(const IndexView &ix, uint32_t idx)
    {
        if (ix.g_kind == 0u) {
            const uint32_t blk = idx / ix.g_used, t = idx - blk * ix.g_used;
            const uint64_t *planes = ix.g_planes + static_cast<uint64_t>(blk) * ix.nbits * ix.g_wpb + (t >> 6);
            uint32_t c = 0;
            for (int b = 0; b < ix.nbits; b++)
                c |= static_cast<uint32_t>((planes[static_cast<uint32_t>(b) * ix.g_wpb] >> (t & 63u)) & 1ull) << b;
            return c;
        }
        const uint32_t blk = idx / ix.g_used, t = idx - blk * ix.g_used + 16u;
        const uint64_t *blkw = ix.g_planes + static_cast<uint64_t>(blk) * ix.sigma * ix.g_wpb + (t >> 6);
        uint32_t c = 0;
        for (int s = 0; s < ix.sigma; s++)  // flat.rs:248-266: exactly one indicator bit is set
            c = ((blkw[static_cast<uint32_t>(s) * ix.g_wpb] >> (t & 63u)) & 1ull) ? static_cast<uint32_t>(s) : c;
        return c;
    }
    static __device__ __forceinline__ uint32_t symbol_and_rank(const IndexView &ix, uint32_t idx, uint32_t &rank_out)
    {
        const uint32_t c = symbol_at(ix, idx);
        rank_out = rank(ix, c, idx);
        return c;
    }
};

// sampled_suffix_array.rs:118 `i % sampling_rate == 0` and :133 `i / sampling_rate` in three instructions for any
// rate: q = rotr(i * sa_inv, sa_rot) equals i / sa_rate when sa_rate divides i and exceeds sa_limit otherwise (the
// divisibility test by multiplication with the modular inverse of the odd part; the rotation checks the power of two)
__device__ __forceinline__ bool sampled_slot(const IndexView &ix, uint32_t i, uint32_t &slot)
{
    const uint32_t m = i * ix.sa_inv;
    slot = __builtin_amdgcn_alignbit(m, m, ix.sa_rot);
    return slot <= ix.sa_limit;
}
__device__ __forceinline__ bool is_sampled(const IndexView &ix, uint32_t i)
{
    uint32_t slot;
    return sampled_slot(ix, i, slot);
}

// lower_bound over a small sorted u32 array (text ids, border keys)
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint32_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

}  // namespace gdx
