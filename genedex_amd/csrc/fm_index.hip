// fm_index.hip -- index construction kernels (everything but the suffix sorter) and the host-side
// FmIndex object.  Construction follows the reference's data flow (construction/mod.rs:25-57):
// concatenate + densely encode -> count -> suffix array -> BWT + text borders -> SA sampling ->
// occurrence table -> lookup tables; every stage is a HIP kernel, nothing is computed on the host
// except O(#texts) bookkeeping.
#include "fm_index.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "index_file.hpp"
#include "kernels.hpp"

namespace gdx {

// IndexView::perm_*: the alphabet table restricted to the dense symbols 1..4 as two 8-entry byte tables indexed by the
// low three bits of the IO byte.  Exists when the searchable bytes that share their low three bits are one symbol in
// at most two spellings that differ in bit 5 only (A / a), which holds for every DNA alphabet of the reference
// (alphabet.rs:264-345); tried with the exact mask first, then case-insensitively.
static void make_perm_translation(const uint8_t *io_to_dense, IndexView &view)
{
    view.perm_ok = 0;
    view.perm_code_lo = view.perm_code_hi = view.perm_exp_lo = view.perm_exp_hi = view.perm_mask = 0;
    for (uint32_t mask : {0xffu, 0xdfu}) {
        uint8_t code[8], expect[8];
        bool ok = true;
        for (uint32_t k = 0; k < 8 && ok; k++) {
            int e = -1, d = -1;
            for (uint32_t c = k; c < 256 && ok; c += 8) {
                const uint32_t dense = io_to_dense[c];
                if (dense < 1 || dense > 4) continue;
                if (e < 0) {
                    e = static_cast<int>(c & mask);
                    d = static_cast<int>(dense);
                } else if (e != static_cast<int>(c & mask) || d != static_cast<int>(dense)) {
                    ok = false;
                }
            }
            if (e < 0) {  // no searchable byte ends in these bits: an entry no byte with these low bits can equal
                expect[k] = static_cast<uint8_t>(k ^ 1u);
                code[k] = 0;
                continue;
            }
            for (uint32_t c = k; c < 256 && ok; c += 8) {  // no other byte may pass the test
                const uint32_t dense = io_to_dense[c];
                if ((dense < 1 || dense > 4) && static_cast<int>(c & mask) == e) ok = false;
            }
            expect[k] = static_cast<uint8_t>(e);
            code[k] = static_cast<uint8_t>(d - 1);
        }
        if (!ok) continue;
        auto word = [](const uint8_t *b) {
            return static_cast<uint32_t>(b[0]) | (static_cast<uint32_t>(b[1]) << 8) | (static_cast<uint32_t>(b[2]) << 16) |
                   (static_cast<uint32_t>(b[3]) << 24);
        };
        view.perm_code_lo = word(code);
        view.perm_code_hi = word(code + 4);
        view.perm_exp_lo = word(expect);
        view.perm_exp_hi = word(expect + 4);
        view.perm_mask = mask * 0x01010101u;
        view.perm_ok = 1;
        return;
    }
}

namespace {

constexpr int kBlock = 256;

unsigned grid_for_items(uint64_t items, uint64_t cap = 256u * 16u)
{
    const uint64_t blocks = (items + kBlock - 1) / kBlock;
    return static_cast<unsigned>(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

double now_seconds()
{
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double>(clk::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------
// construction/mod.rs:255-308: text t occupies [sentinel[t-1]+1, sentinel[t]) of the concatenation and
// is followed by one sentinel (dense 0).  hist[c] += occurrences; *error = 1 on a symbol outside the
// alphabet (alphabet.rs:195-198 panics).
__global__ __launch_bounds__(kBlock) void encode_concat_kernel(const uint8_t *__restrict__ io_text,
                                                               const uint32_t *__restrict__ sentinels,
                                                               uint32_t n_texts, uint64_t n,
                                                               const uint8_t *__restrict__ io_to_dense,
                                                               uint8_t *__restrict__ dense,
                                                               unsigned long long *__restrict__ hist,
                                                               uint32_t *__restrict__ error)
{
    __shared__ uint32_t s_hist[256];
    __shared__ uint8_t s_dense[256];
    for (int i = threadIdx.x; i < 256; i += kBlock) {
        s_hist[i] = 0;
        s_dense[i] = io_to_dense[i];
    }
    __syncthreads();
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; p < n; p += stride) {
        const uint32_t t = lower_bound_u32(sentinels, n_texts, static_cast<uint32_t>(p));
        uint8_t d = 0;
        if (sentinels[t] != static_cast<uint32_t>(p)) {
            d = s_dense[io_text[p - t]];  // t sentinels precede position p
            if (d == 0) *error = 1;
        }
        dense[p] = d;
        atomicAdd(&s_hist[d], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += kBlock)
        if (s_hist[i]) atomicAdd(&hist[i], static_cast<unsigned long long>(s_hist[i]));
}

__global__ __launch_bounds__(kBlock) void histogram_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                           unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t s_hist[256];
    for (int i = threadIdx.x; i < 256; i += kBlock) s_hist[i] = 0;
    __syncthreads();
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; p < n; p += stride)
        atomicAdd(&s_hist[text[p]], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += kBlock)
        if (s_hist[i]) atomicAdd(&hist[i], static_cast<unsigned long long>(s_hist[i]));
}

// bwt.rs:93-116 (bwt[i] = text[SA[i]-1], SA[i]==0 wraps to the last symbol; the border map collects
// {i -> SA[i]} for every BWT sentinel) fused with sampled_suffix_array.rs:37-43 (keep SA[i], i%rate==0)
__global__ __launch_bounds__(kBlock) void bwt_samples_borders_kernel(const uint8_t *__restrict__ text,
                                                                     const uint32_t *__restrict__ sa, uint64_t n,
                                                                     uint32_t rate, uint8_t *__restrict__ bwt,
                                                                     uint32_t *__restrict__ samples,
                                                                     uint32_t *__restrict__ border_keys,
                                                                     uint32_t *__restrict__ border_vals,
                                                                     uint32_t *__restrict__ n_borders)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; j < n; j += stride) {
        const uint32_t s = sa[j];
        const uint64_t ti = s > 0 ? s : n;
        const uint8_t b = text[ti - 1];
        bwt[j] = b;
        if (j % rate == 0) samples[j / rate] = s;
        if (b == 0) {
            const uint32_t at = atomicAdd(n_borders, 1u);
            border_keys[at] = static_cast<uint32_t>(j);
            border_vals[at] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// rank lines (layout 0).  One block per superblock of 65536 positions = 512 lines; thread t builds
// lines 2t and 2t+1, then a block-wide exclusive scan of the per-line symbol counts yields the u16
// block offsets (condensed.rs:365-415 fill_superblock, two Block64 fused per line).
// bwt must be readable (zero padded) up to n_lines * 128 bytes.

__device__ __forceinline__ void build_line_planes(const uint8_t *__restrict__ src, uint32_t (&x)[4],
                                                  uint32_t (&y)[4], uint32_t (&z)[4])
{
    const uint4 *v = reinterpret_cast<const uint4 *>(src);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t xx = 0, yy = 0, zz = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint4 w = v[2 * j + h];
            const uint32_t words[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t s = (words[k] >> (8 * b)) & 0xffu;
                    const int t = h * 16 + k * 4 + b;
                    xx |= (s & 1u) << t;
                    yy |= ((s >> 1) & 1u) << t;
                    zz |= ((s >> 2) & 1u) << t;
                }
            }
        }
        x[j] = xx;
        y[j] = yy;
        z[j] = zz;
    }
}

__device__ __forceinline__ uint32_t planes_count(const uint32_t (&x)[4], const uint32_t (&y)[4],
                                                 const uint32_t (&z)[4], uint32_t c)
{
    const uint32_t n0 = (c & 1u) ? 0u : ~0u, n1 = (c & 2u) ? 0u : ~0u, n2 = (c & 4u) ? 0u : ~0u;
    uint32_t pop = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) pop += __popc((x[j] ^ n0) & (y[j] ^ n1) & (z[j] ^ n2));
    return pop;
}

__global__ __launch_bounds__(kBlock) void build_lines_kernel(const uint8_t *__restrict__ bwt, uint64_t n,
                                                             uint64_t n_lines, u32x4 *__restrict__ lines,
                                                             uint32_t *__restrict__ sb_totals)
{
    __shared__ uint32_t s_scan[8][kBlock];
    const uint64_t sb = blockIdx.x;
    const uint32_t t = threadIdx.x;
    uint32_t x[2][4], y[2][4], z[2][4];
    uint32_t cnt[2][8];
    uint32_t pair_sum[8];
#pragma unroll
    for (int c = 0; c < 8; c++) pair_sum[c] = 0;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const uint64_t line = sb * kLinesPerSuperblock + 2 * t + l;
        if (line < n_lines) {
            build_line_planes(bwt + line * 128, x[l], y[l], z[l]);
            const uint64_t first = line * 128;
            const uint32_t valid = first >= n ? 0u : (n - first >= 128 ? 128u : static_cast<uint32_t>(n - first));
            uint32_t others = 0;
#pragma unroll
            for (int c = 1; c < 8; c++) {
                cnt[l][c] = planes_count(x[l], y[l], z[l], c);
                others += cnt[l][c];
            }
            cnt[l][0] = valid - others;  // padding decodes as 0 but is not text
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) x[l][j] = y[l][j] = z[l][j] = 0;
#pragma unroll
            for (int c = 0; c < 8; c++) cnt[l][c] = 0;
        }
#pragma unroll
        for (int c = 0; c < 8; c++) pair_sum[c] += cnt[l][c];
    }
    // inclusive Hillis-Steele scan over the 256 thread pairs, 8 symbols at once
#pragma unroll
    for (int c = 0; c < 8; c++) s_scan[c][t] = pair_sum[c];
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        uint32_t add[8];
#pragma unroll
        for (int c = 0; c < 8; c++) add[c] = t >= static_cast<uint32_t>(off) ? s_scan[c][t - off] : 0u;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; c++) s_scan[c][t] += add[c];
        __syncthreads();
    }
    uint32_t before[8];
#pragma unroll
    for (int c = 0; c < 8; c++) before[c] = s_scan[c][t] - pair_sum[c];
    if (t == kBlock - 1) {
#pragma unroll
        for (int c = 0; c < 8; c++) sb_totals[sb * 8 + c] = s_scan[c][t];
    }
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const uint64_t line = sb * kLinesPerSuperblock + 2 * t + l;
        if (line < n_lines) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                u32x4 chunk;
                chunk.x = x[l][j];
                chunk.y = y[l][j];
                chunk.z = z[l][j];
                chunk.w = (before[2 * j] & 0xffffu) | (before[2 * j + 1] << 16);
                lines[line * 4 + j] = chunk;
            }
        }
#pragma unroll
        for (int c = 0; c < 8; c++) before[c] += cnt[l][c];
    }
}

// exclusive prefix sum over superblocks, one thread per symbol column (condensed.rs:104-115)
__global__ void superblock_prefix_kernel(uint32_t *sb, uint64_t n_sb, uint32_t stride)
{
    const uint32_t c = threadIdx.x;
    if (c >= stride) return;
    uint32_t sum = 0;
    for (uint64_t s = 0; s < n_sb; s++) {
        const uint32_t v = sb[s * stride + c];
        sb[s * stride + c] = sum;
        sum += v;
    }
}

// ---------------------------------------------------------------------------------------------
// pair lines (layout.hpp PairTable).  bwt0[i] = text[SA[i]-2] is recovered from the one-step table
// itself, bwt0[i] = bwt1[LF(i)], so that imported indexes (from_parts) get pair lines too.  Rows whose
// bwt1 is the sentinel keep bwt0 = 0: a pair with a sentinel is never searched.

__global__ __launch_bounds__(kBlock) void derive_bwt0_kernel(IndexView ix, uint8_t *__restrict__ bwt0)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; p < ix.n; p += stride) {
        uint32_t r;
        const uint32_t c = LineTable::symbol_and_rank(ix, static_cast<uint32_t>(p), r);
        uint32_t prev = 0;
        if (c != 0) prev = LineTable::symbol_at(ix, ix.count[c] + r);
        bwt0[p] = static_cast<uint8_t>(prev);
    }
}

// Jump table (search.hip, IndexView::jump; format in layout.hpp): entry i has levels j = 1 .. L, level j = {row after
// 8j LF steps from row i, 2-bit codes (dense symbol - 1) of the symbols of steps 8j-7 .. 8j, a valid bit}.  A level
// is valid when its eight symbols are all in 1..4 (no sentinel, no N) and every earlier level is valid.  Level 1
// comes from eight LF steps on the rank lines; later levels are copied from the entries of earlier targets:
//   pass 2: level 2 of i = level 1 of entry t1(i)
//   pass 3: levels 3, 4 of i = levels 1, 2 of entry t2(i)   (16-byte entries keep only the codes of level 3)
//   pass 4: codes of level 5 of i = codes of level 1 of entry t4(i)   (a lookahead: the fifth target gave way to SA[i])
//   pass 5 (32-byte entries): SA[i] by the locate walk from row i (sampled_suffix_array.rs:110-138)
// Every pass reads only fields that earlier passes completed and writes only its own entry, whole words at a
// time, so the passes run in place.
__device__ __forceinline__ uint32_t jump_valid(const uint32_t *e, uint32_t words)
{
    return words == 2 ? (e[1] >> 16) : (e[3] >> 16);
}
__device__ __forceinline__ uint32_t jump_code1(const uint32_t *e, uint32_t words)
{
    return words == 2 ? (e[1] & 0xffffu) : (e[2] & 0xffffu);
}

__global__ __launch_bounds__(kBlock) void derive_jump_level1_kernel(IndexView ix, uint32_t *__restrict__ jump,
                                                                    uint32_t words)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; p < ix.n; p += stride) {
        uint32_t row = static_cast<uint32_t>(p), code = 0, steps = 0;
        for (; steps < kJumpSymbols; steps++) {
            uint32_t r;
            const uint32_t c = LineTable::symbol_and_rank(ix, row, r);
            if (c - 1u >= 4u) break;  // sentinel or a symbol outside 1..4
            code |= (c - 1u) << (2u * (kJumpSymbols - 1u - steps));
            row = ix.count[c] + r;
        }
        // a level-1 code that stops short keeps its leading symbols and their number (bits 8..11 of the valid field):
        // enough to decide a query with fewer symbols left than that (search_fast_kernel4)
        const bool ok = steps == kJumpSymbols;
        const uint32_t flags = ((ok ? 1u : 0u) | (steps << 8)) << 16;
        uint32_t *e = jump + p * words;
        for (uint32_t w = 0; w < words; w++) e[w] = 0u;
        if (ok) e[0] = row;
        if (words == 2) e[1] = code | flags;
        else {
            e[2] = code;
            e[3] = flags;
        }
    }
}

// pass = 2, 3 or 4 (see above); words = 4 (16-byte entries) or 8 (32-byte entries)
__global__ __launch_bounds__(kBlock) void derive_jump_levels_kernel(uint64_t n, uint32_t *__restrict__ jump, uint32_t words,
                                                                    uint32_t pass)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; p < n; p += stride) {
        uint32_t *e = jump + p * words;
        const uint32_t valid = e[3] >> 16;
        if (pass == 2) {
            if (!(valid & 1u)) continue;
            const uint32_t *s = jump + static_cast<uint64_t>(e[0]) * words;
            if (!(jump_valid(s, words) & 1u)) continue;
            e[1] = s[0];
            e[2] = (e[2] & 0xffffu) | (jump_code1(s, words) << 16);
            e[3] = (e[3] & 0xffffu) | ((valid | 2u) << 16);
        } else if (pass == 3) {
            if (!(valid & 2u)) continue;
            const uint32_t *s = jump + static_cast<uint64_t>(e[1]) * words;
            const uint32_t sv = jump_valid(s, words);
            if (!(sv & 1u)) continue;
            uint32_t v = valid | 4u;
            const uint32_t c3 = jump_code1(s, words);
            if (words == 8) {
                e[4] = s[0];
                if (sv & 2u) {
                    e[5] = s[1];
                    e[7] = s[2] >> 16;  // level 4 codes; level 5 comes in pass 4
                    v |= 8u;
                }
            }
            e[3] = c3 | (v << 16);
        } else {
            if (!(valid & 8u)) continue;
            const uint32_t *s = jump + static_cast<uint64_t>(e[5]) * words;
            if (!(jump_valid(s, words) & 1u)) continue;
            e[7] = (e[7] & 0xffffu) | (jump_code1(s, words) << 16);
            e[3] = (e[3] & 0xffffu) | ((valid | 16u) << 16);
        }
    }
}

// SA[row] of one row, recovered exactly as locate would (walk to a sampled row or to a text start); the 32-byte jump
// entries hold it already when they exist
__device__ __forceinline__ uint32_t sa_of_row(const IndexView &ix, uint32_t row)
{
    if (ix.jump != nullptr && ix.jump_bytes == 32) return static_cast<const uint32_t *>(ix.jump)[static_cast<uint64_t>(row) * 8u + 6u];
    uint32_t steps = 0;
    for (;;) {
        uint32_t slot;
        if (sampled_slot(ix, row, slot)) return ix.sa_samples[slot] + steps;
        uint32_t r;
        const uint32_t c = LineTable::symbol_and_rank(ix, row, r);
        if (c == 0) return ix.border_vals[lower_bound_u32(ix.border_keys, ix.n_texts, row)] + steps;
        row = ix.count[c] + r;
        steps++;
    }
}

__global__ __launch_bounds__(kBlock) void fill_sa_full_kernel(IndexView ix, uint32_t *__restrict__ sa, uint32_t stride_words,
                                                              uint32_t offset)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; p < ix.n; p += stride)
        sa[p * stride_words + offset] = sa_of_row(ix, static_cast<uint32_t>(p));
}

// SA[row] of EVERY row by pointer jumping instead of one walk per row: link[r] = {row after d LF steps, d}, or {SA[r], 0}
// once known (sampled rows and rows whose BWT symbol is the sentinel know it from the start).  A round replaces
// {x, d} by {link[x].row, d + link[x].d} -- or resolves r when x is resolved -- so a chain of length L is done after
// log2(L) rounds.  The plain walk is quadratic on long runs of one symbol (an assembly gap of 30 M N: the rows of
// N^j X lie a constant number of rows apart, and where that number is a multiple of the sampling rate a walk only ends
// when the run does: 93 s for the genome-like text of the bench); this is ~25 rounds over the table whatever the text.
// Updates race, harmlessly: every 8-byte value a round can read is a valid {row, distance} pair of its row.
__global__ __launch_bounds__(kBlock) void sa_links_init_kernel(IndexView ix, unsigned long long *__restrict__ link)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; p < ix.n; p += stride) {
        const uint32_t row = static_cast<uint32_t>(p);
        uint32_t slot, r;
        unsigned long long v;
        if (sampled_slot(ix, row, slot)) {
            v = ix.sa_samples[slot];
        } else {
            const uint32_t c = LineTable::symbol_and_rank(ix, row, r);
            if (c == 0) v = ix.border_vals[lower_bound_u32(ix.border_keys, ix.n_texts, row)];
            else v = static_cast<unsigned long long>(ix.count[c] + r) | (1ull << 32);
        }
        link[p] = v;
    }
}

__global__ __launch_bounds__(kBlock) void sa_links_round_kernel(uint64_t n, unsigned long long *__restrict__ link,
                                                                unsigned long long *__restrict__ n_open)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    unsigned long long open = 0;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; p < n; p += stride) {
        const unsigned long long v = link[p];
        const uint32_t d = static_cast<uint32_t>(v >> 32);
        if (d == 0u) continue;
        const unsigned long long m = __atomic_load_n(link + static_cast<uint32_t>(v), __ATOMIC_RELAXED);
        const uint32_t md = static_cast<uint32_t>(m >> 32);
        const unsigned long long nv = md == 0u ? static_cast<unsigned long long>(static_cast<uint32_t>(m) + d)
                                               : (m & 0xffffffffull) | (static_cast<unsigned long long>(d + md) << 32);
        __atomic_store_n(link + p, nv, __ATOMIC_RELAXED);
        open += md != 0u ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) open += __shfl_xor(open, off);
    if ((threadIdx.x & 63u) == 0 && open) atomicAdd(n_open, open);
}

__global__ __launch_bounds__(kBlock) void sa_links_store_kernel(uint64_t n, const unsigned long long *__restrict__ link,
                                                                uint32_t *__restrict__ sa, uint32_t stride_words, uint32_t offset)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; p < n; p += stride)
        sa[p * stride_words + offset] = static_cast<uint32_t>(link[p]);
}

// out[row * stride_words + offset] = SA[row] for every row
void compute_full_sa(const IndexView &ix, uint64_t n, uint32_t *d_out, uint32_t stride_words, uint32_t offset, hipStream_t stream)
{
    const unsigned grid = grid_for_items(n);
    unsigned long long *d_link = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d_link), n * sizeof(unsigned long long) + 8) != hipSuccess) {
        (void)hipGetLastError();  // not enough room for the links: one walk per row
        hipLaunchKernelGGL(fill_sa_full_kernel, dim3(grid), dim3(kBlock), 0, stream, ix, d_out, stride_words, offset);
        return;
    }
    unsigned long long *d_open = d_link + n;
    hipLaunchKernelGGL(sa_links_init_kernel, dim3(grid), dim3(kBlock), 0, stream, ix, d_link);
    unsigned long long open = 1;
    for (int round = 0; round < 64 && open != 0; round++) {
        open = 0;
        GDX_HIP(hipMemsetAsync(d_open, 0, sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(sa_links_round_kernel, dim3(grid), dim3(kBlock), 0, stream, n, d_link, d_open);
        GDX_HIP(hipMemcpyAsync(&open, d_open, sizeof(open), hipMemcpyDeviceToHost, stream));
        GDX_HIP(hipStreamSynchronize(stream));
    }
    if (open != 0) {
        (void)hipFree(d_link);
        fail(GDX_ERR_INVALID_ARGUMENT, "the suffix-array samples and the BWT are inconsistent (an LF chain never reaches a sample)");
    }
    hipLaunchKernelGGL(sa_links_store_kernel, dim3(grid), dim3(kBlock), 0, stream, n, d_link, d_out, stride_words, offset);
    GDX_HIP(hipStreamSynchronize(stream));
    (void)hipFree(d_link);
}

// Text units (layout.hpp): all mask bits set and codes zero to begin with -- the pad units in front, the tail behind the
// text -- then every row r puts its BWT symbol where it stands in the text: text[SA[r] - 1] = bwt[r] (bwt.rs:93-116 read
// backwards; the row with SA = 0 holds the last sentinel).  Works for built, imported and loaded indexes alike.
__global__ __launch_bounds__(kBlock) void init_text_units_kernel(u32x4 *__restrict__ units, uint64_t n_units, uint64_t n)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t u = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; u < n_units; u += stride) {
        // positions of the text proper start clean (mask 0): the scatter sets the mask of every non-A C G T symbol
        uint32_t mask = 0xffffffffu;
        if (u >= kTextPadUnits) {
            const uint64_t first = (u - kTextPadUnits) * 32u;
            if (first + 32u <= n) mask = 0u;
            else if (first < n) mask = 0xffffffffu << static_cast<uint32_t>(n - first);
        }
        u32x4 v = {0u, 0u, mask, 0u};
        units[u] = v;
    }
}

__global__ __launch_bounds__(kBlock) void scatter_text_units_kernel(IndexView ix, const uint32_t *__restrict__ sa,
                                                                    uint32_t *__restrict__ words)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t r = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; r < ix.n; r += stride) {
        const uint32_t c = LineTable::symbol_at(ix, static_cast<uint32_t>(r));
        const uint32_t p = sa[r];
        const uint64_t pos = p == 0u ? static_cast<uint64_t>(ix.n) - 1u : static_cast<uint64_t>(p) - 1u;
        uint32_t *unit = words + ((pos >> 5) + kTextPadUnits) * 4u;
        const uint32_t i = static_cast<uint32_t>(pos & 31u);
        if (c - 1u < 4u) {
            if (c != 1u) atomicOr(unit + (i >> 4), (c - 1u) << (2u * (i & 15u)));
        } else {
            atomicOr(unit + 2, 1u << i);
        }
    }
}

// inverse suffix array: isa[SA[r]] = r
__global__ __launch_bounds__(kBlock) void scatter_isa_kernel(const uint32_t *__restrict__ sa, uint64_t n, uint32_t *__restrict__ isa)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t r = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; r < n; r += stride) isa[sa[r]] = static_cast<uint32_t>(r);
}

// ---- seed table (IndexView::seed) -------------------------------------------------------------------------------------
// The rows of the suffix array whose suffix starts with the same k-mer are one interval, so the distinct k-mers of the
// text are the rows whose k-mer differs from the row before ("heads"), and everything about them -- interval, position,
// the symbols in front -- is read off the full suffix array and the text units.

// 32 text symbols from position s (a unit-array position: text position + 32 * kTextPadUnits): codes and "not A C G T" mask
__device__ __forceinline__ void text_window32(const u32x4 *__restrict__ units, uint64_t s, uint64_t &code, uint32_t &mask)
{
    const uint32_t b = static_cast<uint32_t>(s & 31u);
    const u32x4 u0 = units[s >> 5];
    const u32x4 u1 = units[(s >> 5) + 1];  // (two spare units follow the text)
    const uint64_t c0 = static_cast<uint64_t>(u0.x) | (static_cast<uint64_t>(u0.y) << 32);
    const uint64_t c1 = static_cast<uint64_t>(u1.x) | (static_cast<uint64_t>(u1.y) << 32);
    code = b ? (c0 >> (2u * b)) | (c1 << (64u - 2u * b)) : c0;
    mask = b ? (u0.z >> b) | (u1.z << (32u - b)) : u0.z;
}

// the k-mer (k <= 24) at text position p as a key, false when it holds a sentinel / N or runs off the text
__device__ __forceinline__ bool seed_key_at(const u32x4 *__restrict__ units, uint32_t p, uint32_t k, uint64_t &key)
{
    uint64_t code;
    uint32_t mask;
    text_window32(units, static_cast<uint64_t>(p) + 32u * kTextPadUnits, code, mask);
    key = code & ((1ull << (2u * k)) - 1ull);
    return (mask & ((1u << k) - 1u)) == 0u;
}

// the 32 symbols in front of two text positions, when all of them are A C G T of the positions' own texts (IndexView::seed_pairs)
__device__ __forceinline__ bool seed_pair_contexts(const u32x4 *__restrict__ units, uint32_t p1, uint32_t p2, uint64_t &c1, uint64_t &c2)
{
    uint32_t m1, m2;
    text_window32(units, static_cast<uint64_t>(p1) + 32u * kTextPadUnits - 32u, c1, m1);
    text_window32(units, static_cast<uint64_t>(p2) + 32u * kTextPadUnits - 32u, c2, m2);
    return (m1 | m2) == 0u;
}

// n_heads[0] += the distinct k-mers; [1] += those on exactly two rows whose contexts are whole (they get a record in seed_pairs);
// [2] += those on three or four such rows (seed_quads)
__global__ __launch_bounds__(kBlock) void seed_count_heads_kernel(const u32x4 *__restrict__ units, const uint32_t *__restrict__ sa,
                                                                  uint64_t n, uint32_t k, unsigned long long *__restrict__ n_heads)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    unsigned long long mine = 0, pairs = 0, quads = 0;
    for (uint64_t r = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; r < n; r += stride) {
        uint64_t key, prev = 0;
        if (!seed_key_at(units, sa[r], k, key)) continue;
        const bool same = r > 0 && seed_key_at(units, sa[r - 1], k, prev) && prev == key;
        mine += same ? 0u : 1u;
        if (same) continue;
        // rows of the k-mer, as far as it matters here: 1 .. 5
        uint32_t rows = 1;
        while (rows < 5u && r + rows < n && seed_key_at(units, sa[r + rows], k, prev) && prev == key) rows++;
        uint64_t c1, c2;
        if (rows == 2u) pairs += seed_pair_contexts(units, sa[r], sa[r + 1], c1, c2) ? 1u : 0u;
        if (rows == 3u || rows == 4u)
            quads += (seed_pair_contexts(units, sa[r], sa[r + 1], c1, c2) && seed_pair_contexts(units, sa[r + 2], sa[r + rows - 1], c1, c2)) ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) {
        mine += __shfl_xor(mine, off);
        pairs += __shfl_xor(pairs, off);
        quads += __shfl_xor(quads, off);
    }
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(n_heads, mine);
    if ((threadIdx.x & 63u) == 0 && pairs) atomicAdd(n_heads + 1, pairs);
    if ((threadIdx.x & 63u) == 0 && quads) atomicAdd(n_heads + 2, quads);
}

__global__ __launch_bounds__(kBlock) void seed_clear_kernel(u32x4 *__restrict__ table, uint64_t entries)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    const u32x4 empty = {kSeedEmpty, 0u, 0u, 0u};
    for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; e < entries; e += stride) table[e] = empty;
}

// stats: [0] entries of kind 0, [1] entries of kind 1, [2] entries that found no slot within kSeedMaxDisp buckets (the
// build then starts over with more buckets), [3] largest displacement
__global__ __launch_bounds__(kBlock) void seed_insert_kernel(const u32x4 *__restrict__ units, const uint32_t *__restrict__ sa,
                                                             uint64_t n, uint32_t k, uint32_t tag_bits, uint32_t buckets,
                                                             u32x4 *__restrict__ table, uint32_t *__restrict__ fill,
                                                             unsigned long long *__restrict__ stats,
                                                             u32x4 *__restrict__ pair_records, uint32_t pair_capacity,
                                                             uint32_t *__restrict__ n_pair_records,  // [0] pairs, [1] quads
                                                             u32x4 *__restrict__ quad_records, uint32_t quad_capacity)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    unsigned long long n_kind[2] = {0, 0}, n_failed = 0, max_d = 0;
    for (uint64_t r = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; r < n; r += stride) {
        uint64_t key, other = 0;
        const uint32_t p = sa[r];
        if (!seed_key_at(units, p, k, key)) continue;
        if (r > 0 && seed_key_at(units, sa[r - 1], k, other) && other == key) continue;  // not the first row of its k-mer
        // end of the interval: rows r .. hi - 1 start with the k-mer (galloping, then bisection)
        auto same = [&](uint64_t x) { return seed_key_at(units, sa[x], k, other) && other == key; };
        uint64_t hi = r + 1;
        if (hi < n && same(hi)) {
            uint64_t step = 2;
            while (r + step < n && same(r + step)) step *= 2;
            uint64_t good = r + step / 2, bad = r + step < n ? r + step : n;  // same(good), !same(bad) (or bad == n)
            while (bad - good > 1) {
                const uint64_t mid = good + (bad - good) / 2;
                if (same(mid)) good = mid;
                else bad = mid;
            }
            hi = bad;
        }
        u32x4 e = {kSeedKind, static_cast<uint32_t>(r), static_cast<uint32_t>(hi), 0u};
        if (hi - r == 2 && pair_records != nullptr) {  // a two-copy repeat: both positions and what stands in front of them
            const uint32_t p2 = sa[r + 1];
            uint64_t c1, c2;
            if (seed_pair_contexts(units, p, p2, c1, c2)) {
                const uint32_t at = atomicAdd(n_pair_records, 1u);
                if (at < pair_capacity) {
                    pair_records[2ull * at] = u32x4{p, p2, static_cast<uint32_t>(c1), static_cast<uint32_t>(c1 >> 32)};
                    pair_records[2ull * at + 1] = u32x4{static_cast<uint32_t>(c2), static_cast<uint32_t>(c2 >> 32), 0u, 0u};
                    e.x |= kSeedPairInfo;
                    e.w = at;
                }
            }
        }
        if ((hi - r == 3 || hi - r == 4) && quad_records != nullptr) {  // three or four copies: the same, 64 bytes
            const uint32_t rows = static_cast<uint32_t>(hi - r);
            const uint32_t p1 = sa[r + 1], p2 = sa[r + 2], p3 = sa[r + rows - 1];  // (three rows: the third twice)
            uint64_t c0, c1, c2, c3;
            if (seed_pair_contexts(units, p, p1, c0, c1) && seed_pair_contexts(units, p2, p3, c2, c3)) {
                const uint32_t at = atomicAdd(n_pair_records + 1, 1u);
                if (at < quad_capacity) {
                    quad_records[4ull * at] = u32x4{p, p1, p2, p3};
                    quad_records[4ull * at + 1] = u32x4{static_cast<uint32_t>(c0), static_cast<uint32_t>(c0 >> 32), static_cast<uint32_t>(c1), static_cast<uint32_t>(c1 >> 32)};
                    quad_records[4ull * at + 2] = u32x4{static_cast<uint32_t>(c2), static_cast<uint32_t>(c2 >> 32), static_cast<uint32_t>(c3), static_cast<uint32_t>(c3 >> 32)};
                    quad_records[4ull * at + 3] = u32x4{static_cast<uint32_t>(r), rows, 0u, 0u};
                    e.x |= kSeedQuadInfo;
                    e.w = at;
                }
            }
        }
        if (hi - r == 1) {
            uint64_t code;
            uint32_t mask;
            text_window32(units, static_cast<uint64_t>(p) + 32u * kTextPadUnits - 32u, code, mask);
            // symbols A C G T right in front of p (bit 31 of the mask is the symbol at p - 1)
            const uint32_t v = mask == 0u ? 32u : static_cast<uint32_t>(__builtin_clz(mask));
            {
                if (v <= 29u) code = (code & ~63ull) | v;  // (the low three symbols are not among the v)
                e.x = v == 32u ? 0u : (kSeedPartial | (v >= 30u ? (v - 29u) << kSeedPartialShift : 0u));
                e.y = p;
                e.z = static_cast<uint32_t>(code);
                e.w = static_cast<uint32_t>(code >> 32);
            }
        }
        uint32_t tag;
        uint32_t b = seed_home(key, tag_bits, buckets, tag);
        uint32_t d = 0;
        for (;;) {
            const uint32_t slot = atomicAdd(fill + b, 1u);
            if (slot < 8u) {
                e.x |= tag | (d << kSeedDispShift);
                table[(static_cast<uint64_t>(b) << 3) + slot] = e;
                n_kind[(e.x & kSeedKind) ? 1 : 0]++;
                max_d = d > max_d ? d : max_d;
                break;
            }
            if (++d > kSeedMaxDisp) {
                n_failed++;
                break;
            }
            b = b + 1u == buckets ? 0u : b + 1u;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        n_kind[0] += __shfl_xor(n_kind[0], off);
        n_kind[1] += __shfl_xor(n_kind[1], off);
        n_failed += __shfl_xor(n_failed, off);
        const unsigned long long o = __shfl_xor(max_d, off);
        max_d = o > max_d ? o : max_d;
    }
    if ((threadIdx.x & 63u) == 0) {
        if (n_kind[0]) atomicAdd(stats + 0, n_kind[0]);
        if (n_kind[1]) atomicAdd(stats + 1, n_kind[1]);
        if (n_failed) atomicAdd(stats + 2, n_failed);
        if (max_d) atomicMax(stats + 3, max_d);
    }
}

// every entry of a bucket that turned an entry away says so (the search then looks into the next bucket as well);
// *n_flagged counts those buckets
__global__ __launch_bounds__(kBlock) void seed_flag_kernel(u32x4 *__restrict__ table, const uint32_t *__restrict__ fill,
                                                           uint64_t entries, unsigned long long *__restrict__ n_flagged)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    unsigned long long mine = 0;
    for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; e < entries; e += stride) {
        if (fill[e >> 3] > 8u) {
            reinterpret_cast<uint32_t *>(table + e)[0] |= kSeedOverflow;
            mine += (e & 7u) == 0u ? 1u : 0u;
        }
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(n_flagged, mine);
}

// Bit planes of one 64-position block of bwt1 / bwt0 (zero padded input) and its 16 pair + 4 single counts.
struct PairBlock {
    uint64_t p1[3], p0[3];
};

__device__ __forceinline__ PairBlock load_pair_block(const uint8_t *__restrict__ bwt1,
                                                     const uint8_t *__restrict__ bwt0, uint64_t block)
{
    PairBlock pb;
#pragma unroll
    for (int k = 0; k < 3; k++) pb.p1[k] = pb.p0[k] = 0;
    const uint4 *v1 = reinterpret_cast<const uint4 *>(bwt1 + block * 64);
    const uint4 *v0 = reinterpret_cast<const uint4 *>(bwt0 + block * 64);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint4 a = v1[j], b = v0[j];
        const uint32_t wa[4] = {a.x, a.y, a.z, a.w}, wb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int w = 0; w < 4; w++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint64_t s1 = (wa[w] >> (8 * k)) & 0xffu, s0 = (wb[w] >> (8 * k)) & 0xffu;
                const int bit = j * 16 + w * 4 + k;
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    pb.p1[pl] |= ((s1 >> pl) & 1ull) << bit;
                    pb.p0[pl] |= ((s0 >> pl) & 1ull) << bit;
                }
            }
        }
    }
    return pb;
}

__device__ __forceinline__ uint64_t symbol_mask(const uint64_t (&p)[3], int c)
{
    return ((c & 1) ? p[0] : ~p[0]) & ((c & 2) ? p[1] : ~p[1]) & ((c & 4) ? p[2] : ~p[2]);
}

// counts[0..15] = pairs (c2-1)*4 + (c1-1), counts[16..19] = singles c1 = 1..4
__device__ __forceinline__ void count_pair_block(const PairBlock &pb, uint32_t (&counts)[20])
{
#pragma unroll
    for (int c1 = 1; c1 <= 4; c1++) {
        const uint64_t m1 = symbol_mask(pb.p1, c1);
        counts[16 + c1 - 1] = __popcll(m1);
#pragma unroll
        for (int c2 = 1; c2 <= 4; c2++) counts[(c2 - 1) * 4 + (c1 - 1)] = __popcll(m1 & symbol_mask(pb.p0, c2));
    }
}

// One block per superblock of 65536 positions = 1024 pair lines; thread t owns lines 4t .. 4t+3.
// Pass 1 counts, a block-wide scan turns the per-thread totals into offsets inside the superblock,
// pass 2 writes the lines with superblock-relative counts; add_pair_bases_kernel makes them absolute.
__global__ __launch_bounds__(kBlock) void build_pair_lines_kernel(const uint8_t *__restrict__ bwt1,
                                                                  const uint8_t *__restrict__ bwt0, uint64_t n_lines,
                                                                  u32x4 *__restrict__ pair_lines,
                                                                  uint32_t *__restrict__ sb_totals)
{
    __shared__ uint32_t s_scan[20][kBlock];
    const uint64_t sb = blockIdx.x;
    const uint32_t t = threadIdx.x;
    const uint64_t first = sb * 1024 + 4ull * t;
    uint32_t total[20];
#pragma unroll
    for (int k = 0; k < 20; k++) total[k] = 0;
    for (int l = 0; l < 4; l++) {
        if (first + l < n_lines) {
            uint32_t c[20];
            count_pair_block(load_pair_block(bwt1, bwt0, first + l), c);
#pragma unroll
            for (int k = 0; k < 20; k++) total[k] += c[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 20; k++) s_scan[k][t] = total[k];
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        uint32_t add[20];
#pragma unroll
        for (int k = 0; k < 20; k++) add[k] = t >= static_cast<uint32_t>(off) ? s_scan[k][t - off] : 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 20; k++) s_scan[k][t] += add[k];
        __syncthreads();
    }
    uint32_t before[20];
#pragma unroll
    for (int k = 0; k < 20; k++) before[k] = s_scan[k][t] - total[k];
    if (t == kBlock - 1) {
#pragma unroll
        for (int k = 0; k < 20; k++) sb_totals[sb * 20 + k] = s_scan[k][t];
    }
    for (int l = 0; l < 4; l++) {
        const uint64_t line = first + l;
        if (line >= n_lines) break;
        const PairBlock pb = load_pair_block(bwt1, bwt0, line);
        uint32_t c[20];
        count_pair_block(pb, c);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t sh = 8 * j;
            const uint32_t single = before[16 + (j >> 1)];
            u32x4 chunk;
            chunk.x = static_cast<uint32_t>((pb.p1[0] >> sh) & 0xff) | (static_cast<uint32_t>((pb.p1[1] >> sh) & 0xff) << 8) |
                      (static_cast<uint32_t>((pb.p1[2] >> sh) & 0xff) << 16) |
                      (static_cast<uint32_t>((pb.p0[0] >> sh) & 0xff) << 24);
            chunk.y = static_cast<uint32_t>((pb.p0[1] >> sh) & 0xff) | (static_cast<uint32_t>((pb.p0[2] >> sh) & 0xff) << 8) |
                      ((((j & 1) ? (single >> 16) : single) & 0xffffu) << 16);
            chunk.z = before[2 * j];
            chunk.w = before[2 * j + 1];
            pair_lines[line * 8 + j] = chunk;
        }
#pragma unroll
        for (int k = 0; k < 20; k++) before[k] += c[k];
    }
}

// counts in the lines are relative to their superblock: add base[k] + (prefix of the superblock totals)
// so that a line directly yields LF values (pairs: base = C2, singles: base = C).
__global__ __launch_bounds__(kBlock) void add_pair_bases_kernel(u32x4 *__restrict__ pair_lines, uint64_t n_lines,
                                                                const uint32_t *__restrict__ sb_prefix,
                                                                const uint32_t *__restrict__ base)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t line = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; line < n_lines; line += stride) {
        const uint32_t *pre = sb_prefix + (line >> 10) * 20;
        u32x4 c[8];
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = pair_lines[line * 8 + j];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c[j].z += pre[2 * j] + base[2 * j];
            c[j].w += pre[2 * j + 1] + base[2 * j + 1];
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {  // singles: 16-bit halves in chunks 2s (low) and 2s+1 (high)
            const uint32_t rel = (c[2 * s].y >> 16) | ((c[2 * s + 1].y >> 16) << 16);
            const uint32_t abs = rel + pre[16 + s] + base[16 + s];
            c[2 * s].y = (c[2 * s].y & 0xffffu) | ((abs & 0xffffu) << 16);
            c[2 * s + 1].y = (c[2 * s + 1].y & 0xffffu) | ((abs >> 16) << 16);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) pair_lines[line * 8 + j] = c[j];
    }
}

// ---------------------------------------------------------------------------------------------
// generic planes (layout 1) = the reference's own arrays (condensed.rs:24-30), also used to export
// the table of a line-layout index in the reference's logical form.

__global__ __launch_bounds__(kBlock) void build_planes_kernel(const uint8_t *__restrict__ bwt, uint64_t n,
                                                              uint64_t n_blocks, int nbits,
                                                              uint64_t *__restrict__ planes)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t b = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; b < n_blocks; b += stride) {
        uint64_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t t = 0; t < 64; t++) {
            const uint64_t p = b * 64 + t;
            const uint32_t s = p < n ? bwt[p] : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] |= static_cast<uint64_t>((s >> k) & 1u) << t;
        }
        for (int k = 0; k < nbits; k++) planes[b * nbits + k] = w[k];
    }
}

// one thread per (superblock, symbol): walks the superblock once and writes the u16 offset of the
// symbol at every 64-block start, plus the superblock total.  O(n * sigma) work; the generic layout
// serves alphabets with more than 8 dense symbols on small and medium inputs.
__global__ __launch_bounds__(kBlock) void build_generic_offsets_kernel(const uint8_t *__restrict__ bwt, uint64_t n,
                                                                       uint64_t n_blocks, uint64_t n_sb, int sigma,
                                                                       uint16_t *__restrict__ block_off,
                                                                       uint32_t *__restrict__ sb_totals)
{
    const uint64_t total = n_sb * static_cast<uint64_t>(sigma);
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; w < total; w += stride) {
        const uint64_t sb = w / sigma;
        const uint32_t c = static_cast<uint32_t>(w % sigma);
        uint32_t sum = 0;
        const uint64_t b0 = sb * 1024, b1 = (b0 + 1024 < n_blocks) ? b0 + 1024 : n_blocks;
        for (uint64_t b = b0; b < b1; b++) {
            block_off[b * sigma + c] = static_cast<uint16_t>(sum);
            const uint64_t p0 = b * 64, p1 = (p0 + 64 < n) ? p0 + 64 : n;
            for (uint64_t p = p0; p < p1; p++) sum += (bwt[p] == c);
        }
        sb_totals[sb * sigma + c] = sum;
    }
}

// The reference's occurrence table in any of its four variants (IndexView::g_kind ...), from the BWT: one thread per block
// and unit (condensed: bit plane, flat: symbol) packs the block's words ...
__global__ __launch_bounds__(kBlock) void build_ref_blocks_kernel(const uint8_t *__restrict__ bwt, uint64_t n, uint64_t n_blocks,
                                                                  uint32_t kind, uint32_t wpb, uint32_t used, uint32_t units,
                                                                  uint64_t *__restrict__ blocks)
{
    const uint64_t total = n_blocks * units;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; w < total; w += stride) {
        const uint64_t k = w / units;
        const uint32_t u = static_cast<uint32_t>(w % units);
        uint64_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint64_t p0 = k * used;
        for (uint32_t j = 0; j < used && p0 + j < n; j++) {
            const uint32_t s = bwt[p0 + j];
            const uint32_t bit = kind == 0u ? (s >> u) & 1u : (s == u ? 1u : 0u);
            const uint32_t pos = kind == 0u ? j : j + 16u;  // (flat: the first 16 bits of a block are its offset)
#pragma unroll
            for (uint32_t t = 0; t < 8; t++)
                if (t == (pos >> 6)) words[t] |= static_cast<uint64_t>(bit) << (pos & 63u);
        }
        for (uint32_t t = 0; t < wpb; t++) blocks[w * wpb + t] = words[t];
    }
}

// ... and one thread per (superblock, symbol) walks the superblock once: the symbol's offset at every block start (u16, into
// block_off or -- flat -- into the first 16 bits of the block itself; blocks behind the text's last position included:
// idx == n must be addressable) and the superblock total.  O(n * sigma) work, as build_generic_offsets_kernel.
__global__ __launch_bounds__(kBlock) void build_ref_offsets_kernel(const uint8_t *__restrict__ bwt, uint64_t n, uint64_t n_blocks,
                                                                   uint64_t n_sb, uint32_t kind, uint32_t wpb, uint32_t used,
                                                                   uint32_t sb_size, int sigma, uint64_t *__restrict__ blocks,
                                                                   uint16_t *__restrict__ block_off, uint32_t *__restrict__ sb_totals)
{
    const uint64_t total = n_sb * static_cast<uint64_t>(sigma);
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    const uint64_t per_sb = sb_size / used;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; w < total; w += stride) {
        const uint64_t sb = w / sigma;
        const uint32_t c = static_cast<uint32_t>(w % sigma);
        uint32_t sum = 0;
        const uint64_t b0 = sb * per_sb, b1 = b0 + per_sb < n_blocks ? b0 + per_sb : n_blocks;
        const uint64_t sb_end = (sb + 1) * sb_size < n ? (sb + 1) * sb_size : n;
        for (uint64_t b = b0; b < b1; b++) {
            if (kind == 0u) block_off[b * sigma + c] = static_cast<uint16_t>(sum);
            else blocks[(b * sigma + c) * wpb] |= static_cast<uint64_t>(sum & 0xffffu);
            const uint64_t p0 = b * used, p1 = p0 + used < sb_end ? p0 + used : sb_end;
            for (uint64_t p = p0; p < p1; p++) sum += (bwt[p] == c);
        }
        sb_totals[sb * sigma + c] = sum;
    }
}

template <class Table>
__global__ __launch_bounds__(kBlock) void decode_bwt_kernel(IndexView ix, uint8_t *__restrict__ bwt)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; p < ix.n; p += stride)
        bwt[p] = static_cast<uint8_t>(Table::symbol_at(ix, static_cast<uint32_t>(p)));
}

// symbol_at over the reference's own arrays, all four variants:
// condensed (condensed.rs:343-362): plane b of block k = words [(k * nbits + b) * wpb, +wpb), bit j of it;
// flat (flat.rs:248-266): block of symbol c = words [(k * sigma + c) * wpb, +wpb), text bit j + 16.
__global__ __launch_bounds__(kBlock) void decode_reference_blocks_kernel(const uint64_t *__restrict__ blocks,
                                                                         int table_kind, int block_bits, int nbits,
                                                                         int sigma, uint64_t n,
                                                                         uint8_t *__restrict__ bwt,
                                                                         uint32_t *__restrict__ error)
{
    const uint64_t wpb = static_cast<uint64_t>(block_bits) / 64;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; p < n; p += stride) {
        uint32_t s = 0;
        if (table_kind == 0) {
            const uint64_t k = p / block_bits, j = p % block_bits;
            const uint64_t *w = blocks + k * nbits * wpb;
            for (int b = 0; b < nbits; b++) s |= static_cast<uint32_t>((w[b * wpb + j / 64] >> (j % 64)) & 1ull) << b;
        } else {
            const uint64_t used = static_cast<uint64_t>(block_bits) - 16;
            const uint64_t k = p / used, j = p % used + 16;
            const uint64_t *w = blocks + k * sigma * wpb;
            uint32_t found = 0;
            for (int c = 0; c < sigma; c++)
                if ((w[c * wpb + j / 64] >> (j % 64)) & 1ull) {
                    s = static_cast<uint32_t>(c);
                    found++;
                }
            if (found != 1) *error = 1;  // flat.rs:265 unreachable!(): exactly one indicator bit per position
        }
        bwt[p] = static_cast<uint8_t>(s);
    }
}

__global__ __launch_bounds__(kBlock) void widen_u32_kernel(const uint32_t *in, uint64_t *out, uint64_t m)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < m; i += stride) out[i] = in[i];
}

__global__ __launch_bounds__(kBlock) void narrow_u64_kernel(const uint64_t *in, uint32_t *out, uint64_t m,
                                                            uint64_t limit, uint32_t *error)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < m; i += stride) {
        const uint64_t v = in[i];
        if (v > limit) *error = 1;
        out[i] = static_cast<uint32_t>(v);
    }
}

// from_parts: flag |= 1 if a suffix-array sample is not a text position, |= 2 if a border key is not a BWT sentinel row
__global__ __launch_bounds__(kBlock) void check_parts_kernel(const uint32_t *__restrict__ samples, uint64_t n_samples,
                                                             uint32_t n, const uint32_t *__restrict__ border_keys,
                                                             uint32_t n_texts, const uint8_t *__restrict__ bwt,
                                                             uint32_t *__restrict__ flag)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_samples; i += stride)
        if (samples[i] >= n) atomicOr(flag, 1u);
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_texts; i += stride)
        if (bwt[border_keys[i]] != 0) atomicOr(flag, 2u);
}

}  // namespace

// rank lines + per-superblock symbol totals of a zero padded BWT (used by the wide index too, wide.hip)
void launch_build_lines(const uint8_t *d_bwt_padded, uint64_t n, uint64_t n_lines, u32x4 *d_lines, uint32_t *d_sb_totals,
                        uint64_t n_sb, hipStream_t stream)
{
    hipLaunchKernelGGL(build_lines_kernel, dim3(static_cast<unsigned>(n_sb)), dim3(kBlock), 0, stream, d_bwt_padded, n, n_lines,
                       d_lines, d_sb_totals);
}

namespace {

int ilog2_ceil(uint64_t v)  // condensed.rs:417-419
{
    int bits = 0;
    while ((1ull << bits) < v) bits++;
    return bits;
}

void validate_config(const IndexConfig &cfg)
{
    if (cfg.sigma < 2 || cfg.sigma > 256) fail(GDX_ERR_INVALID_ARGUMENT, "sigma must be in 2..=256 (condensed.rs:64)");
    if (cfg.n_searchable < 1 || cfg.n_searchable > cfg.sigma - 1)
        fail(GDX_ERR_INVALID_ARGUMENT, "n_searchable must be in 1..=sigma-1 (alphabet.rs:183-186)");
    if (cfg.sa_rate == 0 || cfg.sa_rate > 0xffffffffull)
        fail(GDX_ERR_INVALID_ARGUMENT, "suffix_array_sampling_rate must be > 0 (config.rs:28)");
    if (cfg.index_width != 32 && cfg.index_width != -32 && cfg.index_width != 64)
        fail(GDX_ERR_INVALID_ARGUMENT, "index_width must be 32 (u32), -32 (i32) or 64 (i64)");
    if (cfg.lookup_depth < 0 || cfg.lookup_depth > kMaxLookupDepth)
        fail(GDX_ERR_INVALID_ARGUMENT, "lookup_table_depth must be in 0..=%d", kMaxLookupDepth);
    // all tables 0..=depth (8 bytes per entry) must fit the device beside the index itself: at most a third of its
    // memory (DNA: depth 16 = 46 GB, depth 17 = 183 GB is refused on a 288 GB GPU)
    double total = 0, pw = 1;
    for (int t = 0; t <= cfg.lookup_depth; t++) {
        total += pw;
        pw *= cfg.n_searchable;
    }
    {
        size_t free_b = 0, total_b = 0;
        double limit = 2147483648.0 * 8.0;  // (no device yet: the old bound)
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b != 0) limit = static_cast<double>(total_b) / 3.0;
        else (void)hipGetLastError();
        if (total * 8.0 > limit || total >= 1.8e19)
            fail(GDX_ERR_INVALID_ARGUMENT, "lookup tables of depth %d would need %.3g bytes", cfg.lookup_depth, total * 8.0);
    }
    // every entry point (build, import, load of an untrusted file): a dense code indexes count[] and the superblock
    // offsets at query time
    for (int b = 0; b < 256; b++)
        if (cfg.io_to_dense[b] >= cfg.sigma)
            fail(GDX_ERR_INVALID_ARGUMENT, "io_to_dense[%d] = %d is not a dense symbol (sigma = %d)", b,
                 static_cast<int>(cfg.io_to_dense[b]), cfg.sigma);
    int dc = 0;
    if (hipGetDeviceCount(&dc) != hipSuccess || dc <= 0) fail(GDX_ERR_DEVICE, "no HIP device available");
    if (cfg.device_id < 0 || cfg.device_id >= dc) fail(GDX_ERR_INVALID_ARGUMENT, "device_id %d out of range", cfg.device_id);
}

void check_width(uint64_t n, int width)
{
    // construction/mod.rs:34 assert!(text.len() <= I::max_value())
    const uint64_t limit = width == -32 ? 0x7fffffffull : 0xffffffffull;
    if (n > limit)
        fail(GDX_ERR_TEXT_TOO_LONG, "total text length %llu (incl. sentinels) exceeds the index storage type",
             static_cast<unsigned long long>(n));
}

}  // namespace

FmIndex::~FmIndex() = default;

void FmIndex::make_current() const { GDX_HIP(hipSetDevice(cfg_.device_id)); }

uint64_t FmIndex::device_bytes() const
{
    return top_.bytes() + jump_.bytes() + sa_full_.bytes() + text_units_.bytes() + seed_.bytes() + isa_.bytes() + pair_lines_.bytes() + lines_.bytes() + sb_offsets_.bytes() + g_planes_.bytes() + g_block_off_.bytes() + count_.bytes() +
           io_to_dense_.bytes() + sa_samples_.bytes() + border_keys_.bytes() + border_vals_.bytes() +
           sentinels_.bytes() + lookup_.bytes();
}

// Builds the occurrence table from the (zero padded) BWT, the lookup tables, and the device view.
// Expects n_, n_texts_, count_host_, sentinels_host_, border_*_host_ and sa_samples_ to be set.
void FmIndex::finish_from_bwt(const uint8_t *d_bwt_padded, hipStream_t stream)
{
    const int sigma = cfg_.sigma;
    const int nbits = ilog2_ceil(static_cast<uint64_t>(sigma));
    const uint64_t len = n_ + 1;  // condensed.rs:69: idx == n must be addressable
    const uint64_t n_sb = div_ceil(len, 65536);
    double t0 = now_seconds();

    view_ = IndexView{};
    const int ref_layout = cfg_.build.table_layout >= 1 ? cfg_.build.table_layout : 0;  // 1..4: Condensed64/512, Flat64/512
    view_.layout = (sigma <= 8 && ref_layout == 0) ? 0 : 1;
    view_.g_kind = 0;
    view_.g_wpb = 1;
    view_.g_used = 64;
    view_.g_sb = 65536;
    if (ref_layout != 0) {
        // the reference's own table, whichever variant was asked for (block.rs, condensed.rs:59-124, flat.rs:59-126)
        const uint32_t kind = ref_layout >= 3 ? 1u : 0u, wpb = (ref_layout == 2 || ref_layout == 4) ? 8u : 1u;
        const uint32_t used = kind == 0u ? 64u * wpb : 64u * wpb - 16u;
        const uint32_t sb_size = kind == 0u ? 65536u : (65536u / used) * used;
        const uint32_t units = kind == 0u ? static_cast<uint32_t>(nbits) : static_cast<uint32_t>(sigma);
        const uint64_t n_blocks = div_ceil(len, used);
        const uint64_t n_sbr = div_ceil(len, sb_size);
        g_planes_.alloc(n_blocks * units * wpb);
        if (kind == 0u) g_block_off_.alloc(n_blocks * sigma);
        sb_offsets_.alloc(n_sbr * sigma);
        hipLaunchKernelGGL(build_ref_blocks_kernel, dim3(grid_for_items(n_blocks * units)), dim3(kBlock), 0, stream, d_bwt_padded,
                           n_, n_blocks, kind, wpb, used, units, g_planes_.get());
        hipLaunchKernelGGL(build_ref_offsets_kernel, dim3(grid_for_items(n_sbr * sigma)), dim3(kBlock), 0, stream, d_bwt_padded,
                           n_, n_blocks, n_sbr, kind, wpb, used, sb_size, sigma, g_planes_.get(), g_block_off_.get(),
                           sb_offsets_.get());
        hipLaunchKernelGGL(superblock_prefix_kernel, dim3(1), dim3(256), 0, stream, sb_offsets_.get(), n_sbr,
                           static_cast<uint32_t>(sigma));
        view_.sb_stride = static_cast<uint32_t>(sigma);
        view_.g_kind = kind;
        view_.g_wpb = wpb;
        view_.g_used = used;
        view_.g_sb = sb_size;
    } else if (view_.layout == 0) {
        const uint64_t n_lines = div_ceil(len, 128);
        lines_.alloc(n_lines * 4);
        sb_offsets_.alloc(n_sb * 8);
        hipLaunchKernelGGL(build_lines_kernel, dim3(static_cast<unsigned>(n_sb)), dim3(kBlock), 0, stream,
                           d_bwt_padded, n_, n_lines, lines_.get(), sb_offsets_.get());
        hipLaunchKernelGGL(superblock_prefix_kernel, dim3(1), dim3(64), 0, stream, sb_offsets_.get(), n_sb, 8u);
        view_.sb_stride = 8;
    } else {
        const uint64_t n_blocks = div_ceil(len, 64);
        g_planes_.alloc(n_blocks * nbits);
        g_block_off_.alloc(n_blocks * sigma);
        sb_offsets_.alloc(n_sb * sigma);
        hipLaunchKernelGGL(build_planes_kernel, dim3(grid_for_items(n_blocks)), dim3(kBlock), 0, stream,
                           d_bwt_padded, n_, n_blocks, nbits, g_planes_.get());
        hipLaunchKernelGGL(build_generic_offsets_kernel, dim3(grid_for_items(n_sb * sigma)), dim3(kBlock), 0, stream,
                           d_bwt_padded, n_, n_blocks, n_sb, sigma, g_block_off_.get(), sb_offsets_.get());
        hipLaunchKernelGGL(superblock_prefix_kernel, dim3(1), dim3(256), 0, stream, sb_offsets_.get(), n_sb,
                           static_cast<uint32_t>(sigma));
        view_.sb_stride = static_cast<uint32_t>(sigma);
    }

    // small arrays
    std::vector<uint32_t> count32(sigma + 1);
    for (int c = 0; c <= sigma; c++) count32[c] = static_cast<uint32_t>(count_host_[c]);
    count_.alloc(sigma + 1);
    GDX_HIP(hipMemcpyAsync(count_.get(), count32.data(), count32.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    io_to_dense_.alloc(256);
    GDX_HIP(hipMemcpyAsync(io_to_dense_.get(), cfg_.io_to_dense, 256, hipMemcpyHostToDevice, stream));
    std::vector<uint32_t> tmp(n_texts_);
    sentinels_.alloc(n_texts_);
    border_keys_.alloc(n_texts_);
    border_vals_.alloc(n_texts_);
    for (uint64_t t = 0; t < n_texts_; t++) tmp[t] = static_cast<uint32_t>(sentinels_host_[t]);
    GDX_HIP(hipMemcpy(sentinels_.get(), tmp.data(), n_texts_ * sizeof(uint32_t), hipMemcpyHostToDevice));
    for (uint64_t t = 0; t < n_texts_; t++) tmp[t] = static_cast<uint32_t>(border_keys_host_[t]);
    GDX_HIP(hipMemcpy(border_keys_.get(), tmp.data(), n_texts_ * sizeof(uint32_t), hipMemcpyHostToDevice));
    for (uint64_t t = 0; t < n_texts_; t++) tmp[t] = static_cast<uint32_t>(border_vals_host_[t]);
    GDX_HIP(hipMemcpy(border_vals_.get(), tmp.data(), n_texts_ * sizeof(uint32_t), hipMemcpyHostToDevice));

    // lookup tables, all depths concatenated (lookup_table.rs:163-181)
    lookup_off_host_.assign(kMaxLookupDepth + 2, 0);
    uint64_t entries = 0, pw = 1;
    // (k^depth entries of 8 bytes: 22 searchable symbols overflow 64 bits at depth 15, 95 at depth 10 -- a table that cannot
    // exist is an argument error, not a wrapped size and an out-of-bounds fill; 2^40 entries = 8 TB is beyond any device)
    constexpr uint64_t kMaxLookupEntries = 1ull << 40;
    for (int t = 0; t <= cfg_.lookup_depth; t++) {
        lookup_off_host_[t] = entries;
        entries += pw;
        if (entries > kMaxLookupEntries)
            fail(GDX_ERR_INVALID_ARGUMENT, "lookup_table_depth %d with %d searchable symbols needs more than 2^40 table entries",
                 cfg_.lookup_depth, cfg_.n_searchable);
        if (t < cfg_.lookup_depth) pw *= static_cast<uint64_t>(cfg_.n_searchable);  // (pw <= entries <= 2^40: no overflow)
    }
    for (int t = cfg_.lookup_depth + 1; t < kMaxLookupDepth + 2; t++) lookup_off_host_[t] = entries;
    lookup_.alloc(entries);

    view_.lines = lines_.get();
    view_.sb_offsets = sb_offsets_.get();
    view_.g_planes = g_planes_.get();
    view_.g_block_off = g_block_off_.get();
    view_.count = count_.get();
    view_.io_to_dense = io_to_dense_.get();
    view_.sa_samples = sa_samples_.get();
    view_.border_keys = border_keys_.get();
    view_.border_vals = border_vals_.get();
    view_.sentinels = sentinels_.get();
    view_.lookup = lookup_.get();
    view_.n = static_cast<uint32_t>(n_);
    view_.n_texts = static_cast<uint32_t>(n_texts_);
    view_.sa_rate = static_cast<uint32_t>(cfg_.sa_rate);
    {
        uint32_t d = static_cast<uint32_t>(cfg_.sa_rate), rot = 0;
        while ((d & 1u) == 0u) {
            d >>= 1;
            rot++;
        }
        uint32_t inv = d;  // Newton iteration for the inverse of an odd number modulo 2^32 (3 correct bits to start)
        for (int it = 0; it < 5; it++) inv *= 2u - d * inv;
        view_.sa_inv = inv;
        view_.sa_rot = rot;
        view_.sa_limit = static_cast<uint32_t>(0xffffffffull / cfg_.sa_rate);
    }
    make_perm_translation(cfg_.io_to_dense, view_);
    view_.sigma = sigma;
    view_.nbits = nbits;
    view_.n_searchable = cfg_.n_searchable;
    view_.depth = cfg_.lookup_depth;

    GDX_HIP(hipStreamSynchronize(stream));
    stats_.seconds_table = now_seconds() - t0;
    t0 = now_seconds();
    const uint2 root = make_uint2(0u, static_cast<uint32_t>(n_));  // lookup_table.rs:205-208
    GDX_HIP(hipMemcpyAsync(lookup_.get(), &root, sizeof(root), hipMemcpyHostToDevice, stream));
    for (int t = 1; t <= cfg_.lookup_depth; t++) launch_fill_lookup(view_, lookup_.get(), t, stream);
    GDX_HIP(hipStreamSynchronize(stream));
    GDX_HIP(hipGetLastError());
    stats_.seconds_lookup = now_seconds() - t0;

    build_aux(d_bwt_padded, stream);
}

// ---- pair lines, jump table, top table (rank-line layout only): BuildOptions decide, the budget trims ----------
void FmIndex::build_aux(const uint8_t *d_bwt_padded, hipStream_t stream)
{
    const int sigma = cfg_.sigma;
    const uint64_t len = n_ + 1;
    const uint64_t n_sb = div_ceil(len, 65536);
    const BuildOptions &bo = cfg_.build;
    aux_report_ = AuxReport{};
    view_.pair_lines = nullptr;
    view_.jump = nullptr;
    view_.jump_bytes = 0;
    view_.top = nullptr;
    view_.top_depth = 0;
    view_.sa_full = nullptr;
    view_.text_units = nullptr;
    view_.seed = nullptr;
    view_.seed_buckets = view_.seed_k = view_.seed_tag_bits = 0;
    view_.isa = nullptr;
    isa_.release();
    pair_lines_.release();
    jump_.release();
    top_.release();
    sa_full_.release();
    text_units_.release();
    seed_.release();
    seed_pairs_.release();
    seed_quads_.release();
    view_.seed_pairs = view_.seed_quads = nullptr;
    // environment variables are debug overrides of fields left at their default
    auto env_int = [](const char *name, int fallback) {
        const char *e = getenv(name);
        return e ? atoi(e) : fallback;
    };
    int want_pairs = bo.pair_lines;
    if (want_pairs < 0) want_pairs = env_int("GDX_NO_PAIR_LINES", 0) == 1 ? 0 : 1;
    // THE DEFAULT SHAPE (round 6): options left at their defaults on a DNA-like alphabet (rank-line layout, dense symbols 1..4
    // searchable) build ONE index that serves every call at its best measured speed -- seed table (k from the text length, load
    // 60 %) + text units + full suffix array + inverse suffix array + pair lines + a top table of depth <= 14, no jump table:
    // count / locate through the seed table (search_seed_lane_kernel), exact intervals and cursors through seed entry / text /
    // ISA with the pair lines for the steps that empty an interval (search_exact_kernel4<0, ., ., true>), locate with SA[row]
    // one fetch away.  3.1 G symbols: 104 GB.  It needs 8.5 bytes per symbol + the seed table inside the budget for auxiliary
    // structures; where that does not fit -- or any of the structures is asked for or switched off explicitly -- the options
    // mean what they say and the tables of rounds 1-3 (32-byte jump entries + top table, shrunk to the budget) are the default.
    // GDX_DEFAULT_SHAPE=tables: the latter regardless (debugging aid).
    bool default_shape = bo.seed_symbols < 0 && bo.jump_bytes < 0 && bo.full_sa < 0 && bo.inverse_sa < 0 && bo.text_units < 0 &&
                         want_pairs != 0 && view_.layout == 0 && view_.sigma >= 5 && cfg_.n_searchable >= 4 && n_ > 0;
    if (default_shape) {
        const char *e = getenv("GDX_DEFAULT_SHAPE");
        if (e != nullptr && strcmp(e, "tables") == 0) default_shape = false;
    }
    uint32_t seed_load_pct = bo.seed_load_percent > 0 ? static_cast<uint32_t>(bo.seed_load_percent) : 70u;
    auto auto_seed_k = [&] {
        uint32_t k = 8;
        while (k < 24 && (1ull << (2u * (k - 8u))) < n_) k++;
        return k;
    };
    if (default_shape) {  // does it fit?  (the same arithmetic as below)
        size_t free_b = 0, total_b = 0;
        GDX_HIP(hipMemGetInfo(&free_b, &total_b));
        const size_t reserve = total_b / 16 > (4ull << 30) ? total_b / 16 : (4ull << 30);
        double budget = free_b > reserve ? static_cast<double>(free_b - reserve) : 0.0;
        if (bo.aux_budget_bytes != 0) {
            budget = static_cast<double>(bo.aux_budget_bytes) < budget ? static_cast<double>(bo.aux_budget_bytes) : budget;
        } else {
            if (budget > static_cast<double>(total_b / 2)) budget = static_cast<double>(total_b / 2);
            if (const char *e = getenv("GDX_AUX_BUDGET_GB")) budget = std::min(budget, atof(e) * 1e9);
        }
        const uint32_t k = auto_seed_k();
        const uint32_t load = bo.seed_load_percent > 0 ? static_cast<uint32_t>(bo.seed_load_percent) : 60u;
        double seed_bytes = 16.0 / (load / 100.0) * static_cast<double>(n_);
        if (2u * k > kSeedTagBitsMax) seed_bytes = std::max(seed_bytes, 128.0 * static_cast<double>(1ull << (2u * k - kSeedTagBitsMax)));
        const double need = 1.25 * seed_bytes + 8.5 * static_cast<double>(n_);
        if (need > budget) default_shape = false;
        else seed_load_pct = load;
    }
    // seed table (layout.hpp): k from the text length unless given: ceil(log4 n) + 8, so that a k-mer
    // that occurs at all almost always occurs once (3.1 G symbols: 24), at most 24 (k + 32 symbols fit the search window)
    uint32_t seed_k = 0;
    if (default_shape) {
        seed_k = auto_seed_k();
    } else if (bo.seed_symbols >= 1 && view_.sigma >= 5) {
        if (bo.seed_symbols == 1) {
            seed_k = auto_seed_k();
        } else {
            seed_k = static_cast<uint32_t>(bo.seed_symbols);
        }
        if (seed_k < 8u || seed_k > 24u) fail(GDX_ERR_INVALID_ARGUMENT, "seed_symbols must be 1 (automatic) or 8..24");
    }
    const bool want_seed = seed_k != 0;
    const bool want_text = bo.text_units == 1 || want_seed, want_sa_full = bo.full_sa == 1 || default_shape,
               want_isa = bo.inverse_sa == 1 || default_shape;
    aux_report_.default_shape = default_shape;
    if (view_.layout == 0 && n_ > 0 && (want_pairs || want_text || want_sa_full || want_isa)) {
        double t0 = now_seconds();
        if (want_pairs) {
        const uint64_t n_lines = div_ceil(len, 128);
        const uint64_t padded = n_lines * 128;
        DeviceBuffer<uint8_t> d_bwt0(padded);
        GDX_HIP(hipMemsetAsync(d_bwt0.get(), 0, padded, stream));
        hipLaunchKernelGGL(derive_bwt0_kernel, dim3(grid_for_items(n_)), dim3(kBlock), 0, stream, view_, d_bwt0.get());
        const uint64_t n_plines = div_ceil(len, 64);
        pair_lines_.alloc(n_plines * 8);
        DeviceBuffer<uint32_t> d_sbp(n_sb * 20);
        hipLaunchKernelGGL(build_pair_lines_kernel, dim3(static_cast<unsigned>(n_sb)), dim3(kBlock), 0, stream,
                           d_bwt_padded, d_bwt0.get(), n_plines, pair_lines_.get(), d_sbp.get());
        hipLaunchKernelGGL(superblock_prefix_kernel, dim3(1), dim3(64), 0, stream, d_sbp.get(), n_sb, 20u);
        // bases: C2[c2 c1] = C[c2] + rank(c2, C[c1]) (first SA slot of the 2-mer c2 c1) and C[c1]
        uint8_t sym[16];
        uint32_t at[16], r[16], base[20];
        for (int c2 = 1; c2 <= 4; c2++)
            for (int c1 = 1; c1 <= 4; c1++) {
                const int p = (c2 - 1) * 4 + (c1 - 1);
                const bool ok = c1 < sigma && c2 < sigma;
                sym[p] = static_cast<uint8_t>(ok ? c2 : 0);
                at[p] = static_cast<uint32_t>(ok ? count_host_[c1] : 0);
            }
        DeviceBuffer<uint8_t> d_sym(16);
        DeviceBuffer<uint32_t> d_at(16), d_r(16), d_err(1), d_base(20);
        GDX_HIP(hipMemcpyAsync(d_sym.get(), sym, 16, hipMemcpyHostToDevice, stream));
        GDX_HIP(hipMemcpyAsync(d_at.get(), at, sizeof(at), hipMemcpyHostToDevice, stream));
        GDX_HIP(hipMemsetAsync(d_err.get(), 0, sizeof(uint32_t), stream));
        launch_rank_many(view_, d_sym.get(), d_at.get(), 16, d_r.get(), d_err.get(), stream);
        GDX_HIP(hipMemcpyAsync(r, d_r.get(), sizeof(r), hipMemcpyDeviceToHost, stream));
        GDX_HIP(hipStreamSynchronize(stream));
        for (int p = 0; p < 16; p++) base[p] = static_cast<uint32_t>(count_host_[sym[p]]) + r[p];
        for (int c = 1; c <= 4; c++) base[16 + c - 1] = c < sigma ? static_cast<uint32_t>(count_host_[c]) : 0u;
        GDX_HIP(hipMemcpyAsync(d_base.get(), base, sizeof(base), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(add_pair_bases_kernel, dim3(grid_for_items(n_plines)), dim3(kBlock), 0, stream,
                           pair_lines_.get(), n_plines, d_sbp.get(), d_base.get());
        GDX_HIP(hipStreamSynchronize(stream));
        GDX_HIP(hipGetLastError());
        view_.pair_lines = pair_lines_.get();
        }  // pair lines
        // jump table: 32-byte entries by default (BuildOptions::jump_bytes; GDX_JUMP_BYTES / GDX_NO_JUMP_TABLE
        // override the default only)
        uint32_t jump_bytes = 32;
        if (default_shape) {
            jump_bytes = 0;  // (the text, SA and ISA stand in for it)
        } else if (bo.jump_bytes >= 0) {
            jump_bytes = static_cast<uint32_t>(bo.jump_bytes);
        } else {
            jump_bytes = static_cast<uint32_t>(env_int("GDX_JUMP_BYTES", 32));
            if (env_int("GDX_NO_JUMP_TABLE", 0) == 1) jump_bytes = 0;
        }
        if (jump_bytes != 0 && jump_bytes != 8 && jump_bytes != 16 && jump_bytes != 32)
            fail(GDX_ERR_INVALID_ARGUMENT, "jump entry bytes must be 0, 8, 16 or 32");
        if (!want_pairs) jump_bytes = 0;  // the jump table belongs to the pair-line kernels
        // top table depth wanted: even (the pair steps that follow consume two symbols each) and as deep as leaves
        // about one row per entry (4^D <= 2 n), so that most reads can jump right after it; at most 16 (34 GB).
        // Measured (search_variants.md section 25): 3.1 G symbols: 16 beats 14 by 23 %; 2^28: 14 beats 12; 2^24: 12
        // beats 10.
        uint32_t top_depth = 0;
        while (top_depth < 16 && (1ull << (2u * (top_depth + 2u))) <= 2ull * n_) top_depth += 2;
        if (bo.top_depth >= 0) top_depth = static_cast<uint32_t>(bo.top_depth);
        else top_depth = static_cast<uint32_t>(env_int("GDX_TOP_DEPTH", static_cast<int>(top_depth)));
        if (top_depth > 16u) top_depth = 16u;
        // (the default shape's top table only starts the reads the seed table does not answer -- absent k-mers, reads shorter
        // than the seed: depth 14 is 2 GB, depth 16 would be 34)
        if (default_shape && bo.top_depth < 0 && top_depth > 14u) top_depth = 14u;
        if (view_.sigma < 5) top_depth = 0;
        if (!want_pairs && !want_text) top_depth = 0;  // nothing would read it
        aux_report_.wanted_jump_bytes = jump_bytes;
        aux_report_.wanted_top_depth = top_depth;
        // Both tables are optional and have to fit into a budget: BuildOptions::aux_budget_bytes, by default what
        // the device has free minus room for query batches and results (1/16 of the memory, at least 4 GB) but
        // never more than half of the device's memory.  Otherwise they shrink in the order of what each step costs
        // on the headline workload: top 16 -> 14, jump 32 -> 16, top -> 12, jump -> 8, top -> 0, jump -> 0;
        // gdx_index_aux reports what was wanted and what was built.
        {
            size_t free_b = 0, total_b = 0;
            GDX_HIP(hipMemGetInfo(&free_b, &total_b));
            const size_t reserve = total_b / 16 > (4ull << 30) ? total_b / 16 : (4ull << 30);
            double budget = free_b > reserve ? static_cast<double>(free_b - reserve) : 0.0;
            if (bo.aux_budget_bytes != 0) {
                budget = static_cast<double>(bo.aux_budget_bytes) < budget ? static_cast<double>(bo.aux_budget_bytes) : budget;
            } else {
                if (budget > static_cast<double>(total_b / 2)) budget = static_cast<double>(total_b / 2);
                if (const char *e = getenv("GDX_AUX_BUDGET_GB")) {
                    const double b = atof(e) * 1e9;
                    budget = b < budget ? b : budget;
                }
            }
            aux_report_.budget_bytes = static_cast<uint64_t>(budget);
            // (the full suffix array and the text units are asked for explicitly: they count, but do not shrink)
            // (the seed table: 16 bytes per distinct k-mer over the load factor -- about n k-mers)
            const double seed_load = seed_load_pct / 100.0;
            // ... but never fewer than 2^(2k - 21) buckets of 128 bytes ((bucket, tag) must name a k-mer exactly with at most
            // kSeedTagBitsMax tag bits: 17 GB for k = 24 whatever the text), and a placement that fails retries with a quarter
            // more buckets: both are part of what the other tables have to leave room for
            double seed_bytes = 0.0;
            if (want_seed) {
                seed_bytes = 16.0 / seed_load * static_cast<double>(n_);
                if (2u * seed_k > kSeedTagBitsMax) seed_bytes = std::max(seed_bytes, 128.0 * static_cast<double>(1ull << (2u * seed_k - kSeedTagBitsMax)));
                seed_bytes *= 1.25;
                if (seed_bytes > budget && bo.seed_symbols > 1)  // an explicit k whose smallest table cannot fit: say so now
                    fail(GDX_ERR_INVALID_ARGUMENT,
                         "seed_symbols = %u needs a seed table of at least %.1f GB (2^(2k - 21) buckets of 128 bytes, or 16 bytes per "
                         "k-mer over the load factor), the budget for auxiliary structures is %.1f GB: choose a smaller k or "
                         "seed_symbols = 1", seed_k, seed_bytes / 1.25 / 1e9, budget / 1e9);
            }
            const double fixed = ((want_sa_full ? 4.0 : 0.0) + (want_isa ? 4.0 : 0.0)) * static_cast<double>(n_) + (want_text ? 0.5 : 0.0) * static_cast<double>(n_) +
                                 seed_bytes;
            auto need = [&] {
                return fixed + static_cast<double>(jump_bytes) * static_cast<double>(n_) +
                       (top_depth ? 8.0 * static_cast<double>(1ull << (2u * top_depth)) : 0.0);
            };
            while (need() > budget) {
                if (top_depth > 14) top_depth = 14;
                else if (jump_bytes == 32) jump_bytes = 16;
                else if (top_depth > 12) top_depth = 12;
                else if (jump_bytes == 16) jump_bytes = 8;
                else if (top_depth > 0) top_depth -= top_depth >= 2 ? 2 : 1;
                else if (jump_bytes != 0) jump_bytes = 0;
                else break;
            }
        }
        if (jump_bytes != 0) {
            const uint32_t words = jump_bytes / 4;
            // padded to whole 16-byte loads (8-byte entries are read as aligned pairs)
            jump_.alloc(div_ceil(static_cast<uint64_t>(n_) * words, 4) * 4);
            const unsigned grid = grid_for_items(n_);
            hipLaunchKernelGGL(derive_jump_level1_kernel, dim3(grid), dim3(kBlock), 0, stream, view_, jump_.get(), words);
            for (uint32_t pass = 2; jump_bytes >= 16 && pass <= (jump_bytes == 32 ? 4u : 3u); pass++)
                hipLaunchKernelGGL(derive_jump_levels_kernel, dim3(grid), dim3(kBlock), 0, stream, n_, jump_.get(), words,
                                   pass);
            if (jump_bytes == 32) {
                GDX_HIP(hipStreamSynchronize(stream));
                compute_full_sa(view_, n_, jump_.get(), 8u, 6u, stream);
            }
            GDX_HIP(hipStreamSynchronize(stream));
            GDX_HIP(hipGetLastError());
            view_.jump = jump_.get();
            view_.jump_bytes = jump_bytes;
        }
        // top table: the first symbols of a DNA query in one fetch (depth chosen above)
        if (top_depth > 0 && view_.sigma >= 5) {
            top_.alloc(1ull << (2u * top_depth));
            launch_fill_top(view_, top_.get(), top_depth, stream);
            GDX_HIP(hipStreamSynchronize(stream));
            GDX_HIP(hipGetLastError());
            view_.top = top_.get();
            view_.top_depth = top_depth;
            // how repetitive the text is, as the search sees it: the share of text positions whose D-mer interval
            // stays wider than a 4-lane jump can take (reads from there fall back to pair-line steps)
            DeviceBuffer<unsigned long long> d_sum(1);
            GDX_HIP(hipMemsetAsync(d_sum.get(), 0, sizeof(unsigned long long), stream));
            launch_top_wide(top_.get(), top_depth, 4u, d_sum.get(), stream);
            unsigned long long wide = 0;
            GDX_HIP(hipMemcpyAsync(&wide, d_sum.get(), sizeof(wide), hipMemcpyDeviceToHost, stream));
            GDX_HIP(hipStreamSynchronize(stream));
            aux_report_.wide_fraction = n_ ? static_cast<double>(wide) / static_cast<double>(n_) : 0.0;
        }
        // full suffix array: SA[row] by the locate walk (or out of the 32-byte jump entries, which hold it already)
        if (want_sa_full || want_text || want_isa) {
            DeviceBuffer<uint32_t> sa_tmp;
            uint32_t *d_sa = nullptr;
            if (want_sa_full) {
                sa_full_.alloc(n_);
                d_sa = sa_full_.get();
            } else {
                sa_tmp.alloc(n_);
                d_sa = sa_tmp.get();
            }
            if (view_.jump != nullptr && view_.jump_bytes == 32)  // (the entries hold it already)
                hipLaunchKernelGGL(fill_sa_full_kernel, dim3(grid_for_items(n_)), dim3(kBlock), 0, stream, view_, d_sa, 1u, 0u);
            else
                compute_full_sa(view_, n_, d_sa, 1u, 0u, stream);
            if (want_text) {
                const uint64_t n_units = div_ceil(n_, 32) + kTextPadUnits + 2;
                text_units_.alloc(n_units);
                hipLaunchKernelGGL(init_text_units_kernel, dim3(grid_for_items(n_units)), dim3(kBlock), 0, stream,
                                   text_units_.get(), n_units, n_);
                hipLaunchKernelGGL(scatter_text_units_kernel, dim3(grid_for_items(n_)), dim3(kBlock), 0, stream, view_, d_sa,
                                   reinterpret_cast<uint32_t *>(text_units_.get()));
            }
            GDX_HIP(hipStreamSynchronize(stream));
            GDX_HIP(hipGetLastError());
            if (want_seed) {
                // (what the budget leaves beside everything else that is or will be there: the seed table and its pair records)
                const uint64_t others = jump_.bytes() + top_.bytes() + (want_sa_full ? 4ull * n_ : 0ull) + text_units_.bytes() + (want_isa ? 4ull * n_ : 0ull);
                const uint64_t budget_now = aux_report_.budget_bytes != 0 ? aux_report_.budget_bytes : ~0ull;
                build_seed_table(d_sa, seed_k, seed_load_pct, stream, budget_now > others ? budget_now - others : 0ull);
            }
            if (want_isa) {
                isa_.alloc(n_);
                hipLaunchKernelGGL(scatter_isa_kernel, dim3(grid_for_items(n_)), dim3(kBlock), 0, stream, d_sa, n_, isa_.get());
                GDX_HIP(hipStreamSynchronize(stream));
                GDX_HIP(hipGetLastError());
                view_.isa = isa_.get();
            }
            if (want_sa_full) view_.sa_full = sa_full_.get();
            if (want_text) view_.text_units = text_units_.get();
        }
        aux_report_.aux_bytes = jump_.bytes() + top_.bytes() + sa_full_.bytes() + text_units_.bytes() + seed_.bytes() + seed_pairs_.bytes() + seed_quads_.bytes() + isa_.bytes();
        stats_.seconds_pairs = now_seconds() - t0;
    }
}

// Seed table out of the full suffix array and the text units (both on the device; layout.hpp describes the entries).
void FmIndex::build_seed_table(const uint32_t *d_sa, uint32_t k, uint32_t load, hipStream_t stream, uint64_t room_bytes)
{
    if (load < 20u || load > 100u) fail(GDX_ERR_INVALID_ARGUMENT, "seed_load_percent must be 0 (default) or 20..100");
    const unsigned grid = grid_for_items(n_);
    DeviceBuffer<unsigned long long> d_stats(8);
    GDX_HIP(hipMemsetAsync(d_stats.get(), 0, 8 * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(seed_count_heads_kernel, dim3(grid), dim3(kBlock), 0, stream, text_units_.get(), d_sa, n_, k, d_stats.get());
    unsigned long long heads_pairs[3] = {0, 0, 0};
    GDX_HIP(hipMemcpyAsync(heads_pairs, d_stats.get(), sizeof(heads_pairs), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    GDX_HIP(hipGetLastError());
    const unsigned long long heads = heads_pairs[0];
    // records of the two-copy repeats (IndexView::seed_pairs): 32 bytes each, if the budget has room for them beside the table
    // (GDX_SEED_PAIRS=0: none -- experiments)
    static const bool env_no_pairs = [] { const char *e = getenv("GDX_SEED_PAIRS"); return e != nullptr && atoi(e) == 0; }();
    unsigned long long n_pairs = env_no_pairs ? 0ull : heads_pairs[1];
    if (n_pairs > 0xfffffff0ull) n_pairs = 0;
    unsigned long long n_quads = env_no_pairs ? 0ull : heads_pairs[2];  // (k-mers on three or four rows: 64 bytes each)
    if (n_quads > 0xfffffff0ull) n_quads = 0;
    DeviceBuffer<uint32_t> d_n_pairs(2);
    // buckets: the load factor decides, but (bucket, tag) must name a k-mer exactly: 2^(2k - tag bits) <= buckets with
    // at most kSeedTagBitsMax tag bits
    uint64_t buckets = (heads * 100ull + 8ull * load - 1) / (8ull * load);
    if (buckets < 1) buckets = 1;
    if (2u * k > kSeedTagBitsMax && buckets < (1ull << (2u * k - kSeedTagBitsMax))) buckets = 1ull << (2u * k - kSeedTagBitsMax);
    for (int attempt = 0;; attempt++) {
        // (bucket numbers and their sums are 32-bit: a fuller table instead of a failure when a low load factor asks for more)
        if (buckets > (1ull << 31)) {
            if (heads > 8ull * (1ull << 31) || attempt > 0) fail(GDX_ERR_UNSUPPORTED, "seed table: too many buckets");
            buckets = 1ull << 31;
        }
        uint32_t log2b = 0;
        while ((2ull << log2b) <= buckets) log2b++;
        const uint32_t tag_bits = 2u * k > log2b ? 2u * k - log2b : 0u;
        seed_.alloc(buckets * 8);
        DeviceBuffer<uint32_t> d_fill(buckets);
        GDX_HIP(hipMemsetAsync(d_fill.get(), 0, buckets * sizeof(uint32_t), stream));
        GDX_HIP(hipMemsetAsync(d_stats.get(), 0, 8 * sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(seed_clear_kernel, dim3(grid_for_items(buckets * 8)), dim3(kBlock), 0, stream, seed_.get(), buckets * 8);
        const bool pairs_fit = n_pairs != 0 && seed_.bytes() + n_pairs * 32ull <= room_bytes;
        if (pairs_fit) seed_pairs_.alloc(n_pairs * 2);
        else seed_pairs_.release();
        const bool quads_fit = n_quads != 0 && seed_.bytes() + seed_pairs_.bytes() + n_quads * 64ull <= room_bytes;
        if (quads_fit) seed_quads_.alloc(n_quads * 4);
        else seed_quads_.release();
        GDX_HIP(hipMemsetAsync(d_n_pairs.get(), 0, 2 * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(seed_insert_kernel, dim3(grid), dim3(kBlock), 0, stream, text_units_.get(), d_sa, n_, k, tag_bits,
                           static_cast<uint32_t>(buckets), seed_.get(), d_fill.get(), d_stats.get(),
                           pairs_fit ? seed_pairs_.get() : nullptr, static_cast<uint32_t>(pairs_fit ? n_pairs : 0ull), d_n_pairs.get(),
                           quads_fit ? seed_quads_.get() : nullptr, static_cast<uint32_t>(quads_fit ? n_quads : 0ull));
        hipLaunchKernelGGL(seed_flag_kernel, dim3(grid_for_items(buckets * 8)), dim3(kBlock), 0, stream, seed_.get(), d_fill.get(),
                           buckets * 8, d_stats.get() + 4);
        unsigned long long st[5];
        GDX_HIP(hipMemcpyAsync(st, d_stats.get(), sizeof(st), hipMemcpyDeviceToHost, stream));
        GDX_HIP(hipStreamSynchronize(stream));
        GDX_HIP(hipGetLastError());
        if (st[2] != 0) {  // an entry found no slot within kSeedMaxDisp buckets of its home: more buckets
            seed_.release();
            seed_pairs_.release();
            seed_quads_.release();
            if (attempt >= 6) fail(GDX_ERR_UNSUPPORTED, "seed table: no placement found");
            buckets = buckets + buckets / 4 + 1;
            continue;
        }
        view_.seed = seed_.get();
        view_.seed_pairs = seed_pairs_.get();  // (null when none were made)
        view_.seed_quads = seed_quads_.get();
        aux_report_.seed_pair_records = pairs_fit ? n_pairs : 0ull;
        aux_report_.seed_quad_records = quads_fit ? n_quads : 0ull;
        view_.seed_buckets = static_cast<uint32_t>(buckets);
        view_.seed_k = k;
        view_.seed_tag_bits = tag_bits;
        aux_report_.seed_k = k;
        aux_report_.seed_buckets = buckets;
        aux_report_.seed_single = st[0];
        aux_report_.seed_multi = st[1];
        aux_report_.seed_max_disp = st[3];
        aux_report_.seed_overflowed = st[4];
        aux_report_.seed_bytes = seed_.bytes() + seed_pairs_.bytes() + seed_quads_.bytes();
        return;
    }
}

void FmIndex::rebuild_aux(const BuildOptions &opts)
{
    make_current();
    hipStream_t stream = hipStreamPerThread;
    GDX_HIP(hipDeviceSynchronize());
    cfg_.build = opts;
    DeviceBuffer<uint8_t> d_bwt;
    if (view_.layout == 0 && n_ > 0) {
        const uint64_t padded = div_ceil(n_ + 1, 128) * 128;
        // the rank lines stay; the pair lines are rebuilt from the BWT they encode.  The view forgets every table
        // before its buffer goes: if an allocation below throws, the handle stays usable on the rank lines
        view_.pair_lines = nullptr;
        view_.jump = nullptr;
        view_.jump_bytes = 0;
        view_.top = nullptr;
        view_.top_depth = 0;
        view_.sa_full = nullptr;
        view_.text_units = nullptr;
        view_.seed = nullptr;
        view_.seed_pairs = view_.seed_quads = nullptr;
        view_.seed_buckets = view_.seed_k = view_.seed_tag_bits = 0;
        view_.isa = nullptr;
        isa_.release();
        aux_report_ = AuxReport{};
        sa_full_.release();
        text_units_.release();
        seed_.release();
        seed_pairs_.release();
        seed_quads_.release();
        pair_lines_.release();
        jump_.release();
        top_.release();
        d_bwt.alloc(padded);
        GDX_HIP(hipMemsetAsync(d_bwt.get(), 0, padded, stream));
        hipLaunchKernelGGL(decode_bwt_kernel<LineTable>, dim3(grid_for_items(n_)), dim3(kBlock), 0, stream, view_, d_bwt.get());
    }
    build_aux(d_bwt.get(), stream);
    GDX_HIP(hipStreamSynchronize(stream));
}

void FmIndex::set_query_options(const QueryOptions &q)
{
    q_search_variant_.store(q.search_variant);
    q_search_lanes_.store(q.search_lanes);
    q_load_policy_.store(q.load_policy);
    q_schedule_.store(q.length_schedule);
    q_locate_variant_.store(q.locate_variant);
    q_locate_jump_walk_.store(q.locate_jump_walk);
    q_defer_after_.store(q.search_defer_after);
    q_fast_.store(q.search_fast);
    q_exact_.store(q.search_exact);
    q_seed_.store(q.search_seed);
    q_max_hits_.store(q.max_hits_per_query);
}

QueryOptions FmIndex::query_options() const
{
    QueryOptions q;
    q.search_variant = q_search_variant_.load();
    q.search_lanes = q_search_lanes_.load();
    q.load_policy = q_load_policy_.load();
    q.length_schedule = q_schedule_.load();
    q.locate_variant = q_locate_variant_.load();
    q.locate_jump_walk = q_locate_jump_walk_.load();
    q.search_defer_after = q_defer_after_.load();
    q.search_fast = q_fast_.load();
    q.search_exact = q_exact_.load();
    q.search_seed = q_seed_.load();
    q.max_hits_per_query = q_max_hits_.load();
    // default: park stragglers only on repetitive texts (the bookkeeping costs the plain kernel ~15 %)
    // (wide_fraction tells how repetitive the text is only next to a jump table, whose top table is as deep as the text allows;
    // without one -- the default shape, its top table capped at depth 14 -- nothing jumps and nothing is parked)
    if (q.search_defer_after < 0) q.search_defer_after = (view_.jump != nullptr && aux_report_.wide_fraction > 0.05) ? 3 : 0;
    // default: the fast-path kernel first, unless the top table is so shallow for this text that most reads leave it
    // on more rows than a jump takes -- their intervals narrow fastest on pair lines (top 12 / 8-byte jumps at hg38
    // scale: 21.8 ms without, 26.5 ms with the fast path in front)
    // -- and with jumps over up to sixteen rows when reads from repeats are common (they cost every other read 3 %)
    // (an index with text units and no jump table narrows wide intervals on the rank lines and then compares with the
    // text: search_verify_kernel4 is worth running whatever the top table leaves)
    if (q.search_fast < 0 && view_.text_units != nullptr && view_.jump == nullptr) q.search_fast = 1;
    if (q.search_fast < 0) q.search_fast = aux_report_.wide_fraction > 0.5 ? 0 : (aux_report_.wide_fraction > 0.02 ? 2 : 1);
    return q;
}

std::unique_ptr<FmIndex> FmIndex::construct_index(const uint8_t *texts_buf, bool texts_on_device,
                                                  const uint64_t *text_offsets, uint64_t n_texts,
                                                  const IndexConfig &cfg)
{
    validate_config(cfg);
    if (n_texts == 0) fail(GDX_ERR_INVALID_ARGUMENT, "There should be at least one text (construction/mod.rs:303)");
    if (!text_offsets) fail(GDX_ERR_INVALID_ARGUMENT, "text_offsets is null");
    const uint64_t io_len = text_offsets[n_texts] - text_offsets[0];
    for (uint64_t t = 0; t < n_texts; t++)
        if (text_offsets[t + 1] < text_offsets[t]) fail(GDX_ERR_INVALID_ARGUMENT, "text_offsets must be non-decreasing");
    if (io_len > 0 && !texts_buf) fail(GDX_ERR_INVALID_ARGUMENT, "texts_buf is null");
    const uint64_t n = io_len + n_texts;
    check_width(n, cfg.index_width);
    if (n_texts > 0xffffffffull) fail(GDX_ERR_TEXT_TOO_LONG, "too many texts");

    std::unique_ptr<FmIndex> ix(new FmIndex());
    ix->cfg_ = cfg;
    ix->n_ = n;
    ix->n_texts_ = n_texts;
    GDX_HIP(hipSetDevice(cfg.device_id));
    hipStream_t stream = hipStreamPerThread;

    // construction/mod.rs:266-273 sentinel_indices
    ix->sentinels_host_.resize(n_texts);
    std::vector<uint32_t> sent32(n_texts);
    for (uint64_t t = 0; t < n_texts; t++) {
        ix->sentinels_host_[t] = (text_offsets[t + 1] - text_offsets[0]) + t;
        sent32[t] = static_cast<uint32_t>(ix->sentinels_host_[t]);
    }

    double t0 = now_seconds();
    // ---- encode + concatenate + frequency table --------------------------------------------------
    DeviceBuffer<uint8_t> io_owned;
    const uint8_t *d_io = texts_buf ? texts_buf + (texts_on_device ? text_offsets[0] : 0) : nullptr;
    if (!texts_on_device) {
        io_owned.alloc(io_len ? io_len : 1);
        if (io_len) GDX_HIP(hipMemcpy(io_owned.get(), texts_buf + text_offsets[0], io_len, hipMemcpyHostToDevice));
        d_io = io_owned.get();
    }
    DeviceBuffer<uint32_t> d_sent(n_texts);
    GDX_HIP(hipMemcpy(d_sent.get(), sent32.data(), n_texts * sizeof(uint32_t), hipMemcpyHostToDevice));
    DeviceBuffer<uint8_t> d_tab(256);
    GDX_HIP(hipMemcpy(d_tab.get(), cfg.io_to_dense, 256, hipMemcpyHostToDevice));
    DeviceBuffer<uint8_t> d_text(n);
    DeviceBuffer<unsigned long long> d_hist(256);
    DeviceBuffer<uint32_t> d_flag(1);
    GDX_HIP(hipMemsetAsync(d_hist.get(), 0, 256 * sizeof(unsigned long long), stream));
    GDX_HIP(hipMemsetAsync(d_flag.get(), 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(encode_concat_kernel, dim3(grid_for_items(n)), dim3(kBlock), 0, stream, d_io, d_sent.get(),
                       static_cast<uint32_t>(n_texts), n, d_tab.get(), d_text.get(), d_hist.get(), d_flag.get());
    unsigned long long hist[256];
    uint32_t flag = 0;
    GDX_HIP(hipMemcpyAsync(hist, d_hist.get(), sizeof(hist), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipMemcpyAsync(&flag, d_flag.get(), sizeof(flag), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    GDX_HIP(hipGetLastError());
    if (flag) fail(GDX_ERR_INVALID_TEXT_SYMBOL, "symbol in io representation should be valid (alphabet.rs:195-198)");
    for (int c = cfg.sigma; c < 256; c++)
        if (hist[c]) fail(GDX_ERR_INVALID_ARGUMENT, "io_to_dense maps to dense symbol %d >= sigma", c);
    io_owned.release();
    // construction/mod.rs:318-336 frequency_table_to_count
    std::vector<uint64_t> freq(cfg.sigma);
    ix->count_host_.assign(cfg.sigma + 1, 0);
    uint64_t sum = 0;
    for (int c = 0; c < cfg.sigma; c++) {
        freq[c] = hist[c];
        ix->count_host_[c] = sum;
        sum += hist[c];
    }
    ix->count_host_[cfg.sigma] = sum;
    ix->stats_.seconds_encode = now_seconds() - t0;

    // ---- suffix array ----------------------------------------------------------------------------------
    t0 = now_seconds();
    DeviceBuffer<uint32_t> d_sa(n);
    build_suffix_array(d_text.get(), n, cfg.sigma, freq, d_sa.get(), stream, &ix->stats_);
    GDX_HIP(hipStreamSynchronize(stream));
    GDX_HIP(hipGetLastError());
    ix->stats_.seconds_sa = now_seconds() - t0;

    // ---- BWT, SA samples, text borders -------------------------------------------------------------------
    t0 = now_seconds();
    const uint64_t padded = div_ceil(n + 1, 128) * 128;
    DeviceBuffer<uint8_t> d_bwt(padded);
    GDX_HIP(hipMemsetAsync(d_bwt.get(), 0, padded, stream));
    const uint64_t n_samples = div_ceil(n, cfg.sa_rate);
    ix->sa_samples_.alloc(n_samples);
    DeviceBuffer<uint32_t> d_bk(n_texts), d_bv(n_texts), d_nb(1);
    GDX_HIP(hipMemsetAsync(d_nb.get(), 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(bwt_samples_borders_kernel, dim3(grid_for_items(n)), dim3(kBlock), 0, stream, d_text.get(),
                       d_sa.get(), n, static_cast<uint32_t>(cfg.sa_rate), d_bwt.get(), ix->sa_samples_.get(),
                       d_bk.get(), d_bv.get(), d_nb.get());
    std::vector<uint32_t> bk(n_texts), bv(n_texts);
    uint32_t nb = 0;
    GDX_HIP(hipMemcpy(&nb, d_nb.get(), sizeof(nb), hipMemcpyDeviceToHost));
    if (nb != n_texts) fail(GDX_ERR_DEVICE, "internal: %u BWT sentinels for %llu texts", nb, (unsigned long long)n_texts);
    GDX_HIP(hipMemcpy(bk.data(), d_bk.get(), n_texts * sizeof(uint32_t), hipMemcpyDeviceToHost));
    GDX_HIP(hipMemcpy(bv.data(), d_bv.get(), n_texts * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<std::pair<uint32_t, uint32_t>> borders(n_texts);
    for (uint64_t t = 0; t < n_texts; t++) borders[t] = {bk[t], bv[t]};
    std::sort(borders.begin(), borders.end());
    ix->border_keys_host_.resize(n_texts);
    ix->border_vals_host_.resize(n_texts);
    for (uint64_t t = 0; t < n_texts; t++) {
        ix->border_keys_host_[t] = borders[t].first;
        ix->border_vals_host_[t] = borders[t].second;
    }
    d_sa.release();
    d_text.release();
    ix->stats_.seconds_bwt = now_seconds() - t0;

    ix->finish_from_bwt(d_bwt.get(), stream);
    return ix;
}

std::unique_ptr<FmIndex> FmIndex::from_parts(int table_kind, int block_bits, const uint64_t *count,
                                             const uint64_t *interleaved_blocks, uint64_t n,
                                             const uint32_t *sa_samples, const uint64_t *border_keys,
                                             const uint64_t *border_vals, const uint64_t *sentinel_indices,
                                             uint64_t n_texts, const IndexConfig &cfg)
{
    validate_config(cfg);
    if (!count || !interleaved_blocks || !sa_samples || !border_keys || !border_vals || !sentinel_indices)
        fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: null array");
    if (n_texts == 0 || n < n_texts) fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: need at least one text");
    if ((table_kind != 0 && table_kind != 1) || (block_bits != 64 && block_bits != 512))
        fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: table_kind must be 0 (condensed) or 1 (flat), block_bits 64 or 512");
    check_width(n, cfg.index_width);
    std::unique_ptr<FmIndex> ix(new FmIndex());
    ix->cfg_ = cfg;
    ix->n_ = n;
    ix->n_texts_ = n_texts;
    GDX_HIP(hipSetDevice(cfg.device_id));
    hipStream_t stream = hipStreamPerThread;
    ix->count_host_.assign(count, count + cfg.sigma + 1);
    ix->sentinels_host_.assign(sentinel_indices, sentinel_indices + n_texts);
    ix->border_keys_host_.assign(border_keys, border_keys + n_texts);
    ix->border_vals_host_.assign(border_vals, border_vals + n_texts);
    if (ix->count_host_[0] != 0 || ix->count_host_[cfg.sigma] != n || ix->sentinels_host_[n_texts - 1] != n - 1)
        fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: count / sentinel_indices do not describe a text of length n");
    for (uint64_t t = 1; t < n_texts; t++)
        if (border_keys[t] <= border_keys[t - 1]) fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: border keys must be sorted");
    // the arrays below are indexed by / produce text positions on the device: everything has to stay below n
    for (uint64_t t = 0; t < n_texts; t++) {
        if (sentinel_indices[t] >= n || (t > 0 && sentinel_indices[t] <= sentinel_indices[t - 1]))
            fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: sentinel_indices must be strictly increasing and < n");
        if (border_keys[t] >= n || border_vals[t] >= n)
            fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: text border %llu is outside the text", (unsigned long long)t);
    }
    if (ix->count_host_[1] != n_texts)
        fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: count[1] must equal the number of texts (one sentinel per text)");

    const int nbits = ilog2_ceil(static_cast<uint64_t>(cfg.sigma));
    const uint64_t wpb = static_cast<uint64_t>(block_bits) / 64;
    const uint64_t n_words = table_kind == 0 ? div_ceil(n + 1, block_bits) * nbits * wpb
                                             : div_ceil(n + 1, block_bits - 16) * cfg.sigma * wpb;
    DeviceBuffer<uint64_t> d_planes(n_words);
    GDX_HIP(hipMemcpy(d_planes.get(), interleaved_blocks, n_words * sizeof(uint64_t), hipMemcpyHostToDevice));
    const uint64_t padded = div_ceil(n + 1, 128) * 128;
    DeviceBuffer<uint8_t> d_bwt(padded);
    GDX_HIP(hipMemsetAsync(d_bwt.get(), 0, padded, stream));
    DeviceBuffer<uint32_t> d_bad(1);
    GDX_HIP(hipMemsetAsync(d_bad.get(), 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(decode_reference_blocks_kernel, dim3(grid_for_items(n)), dim3(kBlock), 0, stream,
                       d_planes.get(), table_kind, block_bits, nbits, cfg.sigma, n, d_bwt.get(), d_bad.get());
    uint32_t bad = 0;
    GDX_HIP(hipMemcpyAsync(&bad, d_bad.get(), sizeof(bad), hipMemcpyDeviceToHost, stream));
    // the given count must be the prefix sums of the BWT's symbol frequencies
    DeviceBuffer<unsigned long long> d_hist(256);
    GDX_HIP(hipMemsetAsync(d_hist.get(), 0, 256 * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(histogram_kernel, dim3(grid_for_items(n)), dim3(kBlock), 0, stream, d_bwt.get(), n, d_hist.get());
    unsigned long long hist[256];
    GDX_HIP(hipMemcpyAsync(hist, d_hist.get(), sizeof(hist), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    if (bad) fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: flat blocks do not hold exactly one indicator bit per position");
    uint64_t sum = 0;
    for (int c = 0; c < 256; c++) {
        if (c < cfg.sigma && ix->count_host_[c] != sum) fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: count[%d] does not match the bit planes", c);
        if (c >= cfg.sigma && hist[c]) fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: planes hold symbol %d >= sigma", c);
        sum += hist[c];
    }
    d_planes.release();
    const uint64_t n_samples = div_ceil(n, cfg.sa_rate);
    ix->sa_samples_.alloc(n_samples);
    GDX_HIP(hipMemcpy(ix->sa_samples_.get(), sa_samples, n_samples * sizeof(uint32_t), hipMemcpyHostToDevice));
    {
        // every sample is a text position; the border keys are exactly the BWT rows holding the sentinel (the keys
        // are strictly increasing, each must hold a sentinel, and the BWT has n_texts of them: count[1] above)
        DeviceBuffer<uint32_t> d_keys(n_texts), d_flag(1);
        std::vector<uint32_t> keys32(n_texts);
        for (uint64_t t = 0; t < n_texts; t++) keys32[t] = static_cast<uint32_t>(border_keys[t]);
        GDX_HIP(hipMemcpyAsync(d_keys.get(), keys32.data(), n_texts * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        GDX_HIP(hipMemsetAsync(d_flag.get(), 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(check_parts_kernel, dim3(grid_for_items(n_samples > n_texts ? n_samples : n_texts)), dim3(kBlock),
                           0, stream, ix->sa_samples_.get(), n_samples, static_cast<uint32_t>(n), d_keys.get(),
                           static_cast<uint32_t>(n_texts), d_bwt.get(), d_flag.get());
        uint32_t flag = 0;
        GDX_HIP(hipMemcpyAsync(&flag, d_flag.get(), sizeof(flag), hipMemcpyDeviceToHost, stream));
        GDX_HIP(hipStreamSynchronize(stream));
        if (flag & 1u) fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: a suffix array sample is >= n");
        if (flag & 2u) fail(GDX_ERR_INVALID_ARGUMENT, "from_parts: border keys are not the BWT rows that hold the sentinel");
    }
    ix->finish_from_bwt(d_bwt.get(), stream);
    return ix;
}

// =============================================================================================
// host-pointer query API

namespace {

struct DeviceQueries {
    DeviceBuffer<uint8_t> qbuf;
    DeviceBuffer<uint64_t> qoff;
    DeviceQueries(const uint8_t *h_qbuf, const uint64_t *h_qoff, uint64_t nq, hipStream_t stream)
    {
        const uint64_t base = h_qoff[0];
        const uint64_t bytes = h_qoff[nq] - base;
        const uint64_t padded = div_ceil(bytes + 1, 8) * 8;  // 8-byte windows may read past the last query
        qbuf.alloc(padded);
        GDX_HIP(hipMemsetAsync(qbuf.get() + (padded - 8), 0, 8, stream));
        if (bytes) GDX_HIP(hipMemcpyAsync(qbuf.get(), h_qbuf + base, bytes, hipMemcpyHostToDevice, stream));
        qoff.alloc(nq + 1);
        if (base == 0) {
            GDX_HIP(hipMemcpyAsync(qoff.get(), h_qoff, (nq + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        } else {
            std::vector<uint64_t> rel(nq + 1);
            for (uint64_t i = 0; i <= nq; i++) rel[i] = h_qoff[i] - base;
            GDX_HIP(hipMemcpy(qoff.get(), rel.data(), (nq + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        }
    }
};

void check_queries(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq)
{
    if (!qoff) fail(GDX_ERR_INVALID_ARGUMENT, "qoff is null");
    if (nq >= 0xffffffffull) fail(GDX_ERR_UNSUPPORTED, "more than 2^32-2 queries in one call");
    for (uint64_t i = 0; i < nq; i++)
        if (qoff[i + 1] < qoff[i]) fail(GDX_ERR_INVALID_ARGUMENT, "qoff must be non-decreasing");
    if (qoff[nq] > qoff[0] && !qbuf) fail(GDX_ERR_INVALID_ARGUMENT, "qbuf is null");
}

int any_status(const uint8_t *status, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++)
        if (status[i]) return GDX_ERR_QUERY_STATUS;
    return GDX_OK;
}

void download_widened(const uint32_t *d_in, uint64_t *h_out, uint64_t m, hipStream_t stream)
{
    if (!h_out || m == 0) return;
    DeviceBuffer<uint64_t> wide(m);
    hipLaunchKernelGGL(widen_u32_kernel, dim3(grid_for_items(m)), dim3(kBlock), 0, stream, d_in, wide.get(), m);
    GDX_HIP(hipMemcpyAsync(h_out, wide.get(), m * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
}

}  // namespace

void FmIndex::locate_device(const uint32_t *d_start, const uint32_t *d_end, uint64_t m, uint64_t *out_hit_offsets,
                            gdx_hit_t *hits, uint64_t hits_capacity, uint64_t *out_total, int *rc,
                            const uint4 *d_rec) const
{
    hipStream_t stream = hipStreamPerThread;
    DeviceBuffer<uint64_t> d_off(m + 1);
    if (d_rec) {
        const size_t tb = hit_offsets_rec_temp_bytes(m);
        DeviceBuffer<uint8_t> temp(tb ? tb : 1);
        launch_hit_offsets_rec(d_rec, m, d_off.get(), temp.get(), tb, stream);
        GDX_HIP(hipStreamSynchronize(stream));
    } else {
        const size_t tb = hit_offsets_temp_bytes(m);
        DeviceBuffer<uint8_t> temp(tb ? tb : 1);
        launch_hit_offsets(d_start, d_end, m, d_off.get(), temp.get(), tb, stream);
        GDX_HIP(hipStreamSynchronize(stream));
    }
    uint64_t total = 0;
    GDX_HIP(hipMemcpy(&total, d_off.get() + m, sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (out_total) *out_total = total;
    if (out_hit_offsets)
        GDX_HIP(hipMemcpy(out_hit_offsets, d_off.get(), (m + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (total == 0) return;
    if (!hits || hits_capacity < total) {
        *rc = GDX_ERR_CAPACITY;
        return;
    }
    DeviceBuffer<gdx_hit_t> d_hits(total);
    DeviceBuffer<uint8_t> ws(locate_workspace_bytes(total));
    launch_locate(view_, d_start, d_end, m, d_off.get(), total, d_hits.get(), true, ws.get(), stream, nullptr, nullptr,
                  query_options(), d_rec);
    GDX_HIP(hipGetLastError());
    GDX_HIP(hipMemcpyAsync(hits, d_hits.get(), total * sizeof(gdx_hit_t), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
}

namespace {

void upload_narrowed(const uint64_t *h_in, uint32_t *d_out, uint64_t m, uint64_t limit, const char *what,
                     hipStream_t stream)
{
    DeviceBuffer<uint64_t> wide(m);
    DeviceBuffer<uint32_t> flag(1);
    GDX_HIP(hipMemcpyAsync(wide.get(), h_in, m * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    GDX_HIP(hipMemsetAsync(flag.get(), 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(narrow_u64_kernel, dim3(grid_for_items(m)), dim3(kBlock), 0, stream, wide.get(), d_out, m,
                       limit, flag.get());
    uint32_t f = 0;
    GDX_HIP(hipMemcpyAsync(&f, flag.get(), sizeof(f), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    if (f) fail(GDX_ERR_INVALID_ARGUMENT, "%s exceeds the text length", what);
}

}  // namespace

int FmIndex::cursor_extend_front_many(uint64_t *start, uint64_t *end, const uint8_t *io_symbols, uint64_t m,
                                      uint8_t *out_status) const
{
    if (m == 0) return GDX_OK;
    if (!start || !end || !io_symbols) fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
    make_current();
    hipStream_t stream = hipStreamPerThread;
    DeviceBuffer<uint32_t> d_start(m), d_end(m);
    DeviceBuffer<uint8_t> d_sym(m), d_status(m);
    upload_narrowed(start, d_start.get(), m, n_, "cursor start", stream);
    upload_narrowed(end, d_end.get(), m, n_, "cursor end", stream);
    GDX_HIP(hipMemcpyAsync(d_sym.get(), io_symbols, m, hipMemcpyHostToDevice, stream));
    launch_extend_front(view_, d_start.get(), d_end.get(), d_sym.get(), m, d_status.get(), stream);
    GDX_HIP(hipGetLastError());
    download_widened(d_start.get(), start, m, stream);
    download_widened(d_end.get(), end, m, stream);
    std::vector<uint8_t> status(m);
    GDX_HIP(hipMemcpy(status.data(), d_status.get(), m, hipMemcpyDeviceToHost));
    if (out_status) std::memcpy(out_status, status.data(), m);
    return any_status(status.data(), m);
}

int FmIndex::cursor_extend_front_strings(uint64_t *start, uint64_t *end, const uint8_t *qbuf, const uint64_t *qoff,
                                         uint64_t m, uint8_t *status) const
{
    check_queries(qbuf, qoff, m);
    if (m == 0) return GDX_OK;
    if (!start || !end) fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
    make_current();
    hipStream_t stream = hipStreamPerThread;
    DeviceQueries dq(qbuf, qoff, m, stream);
    DeviceBuffer<uint32_t> d_start(m), d_end(m);
    DeviceBuffer<uint8_t> d_status(m);
    upload_narrowed(start, d_start.get(), m, n_, "cursor start", stream);
    upload_narrowed(end, d_end.get(), m, n_, "cursor end", stream);
    if (status) GDX_HIP(hipMemcpyAsync(d_status.get(), status, m, hipMemcpyHostToDevice, stream));
    else GDX_HIP(hipMemsetAsync(d_status.get(), 0, m, stream));
    SearchCall c;
    c.d_qbuf = dq.qbuf.get();
    c.d_qbeg = dq.qoff.get();
    c.d_qend = dq.qoff.get() + 1;
    c.nq = m;
    c.d_start = d_start.get();
    c.d_end = d_end.get();
    c.d_status = d_status.get();
    c.mode = 2;
    launch_search_call(view_, c, stream, query_options());
    GDX_HIP(hipGetLastError());
    download_widened(d_start.get(), start, m, stream);
    download_widened(d_end.get(), end, m, stream);
    std::vector<uint8_t> st(m);
    GDX_HIP(hipMemcpy(st.data(), d_status.get(), m, hipMemcpyDeviceToHost));
    if (status) std::memcpy(status, st.data(), m);
    return any_status(st.data(), m);
}

int FmIndex::cursor_locate_many(const uint64_t *start, const uint64_t *end, uint64_t m, uint64_t *out_hit_offsets,
                                gdx_hit_t *hits, uint64_t hits_capacity, uint64_t *out_total) const
{
    if (out_total) *out_total = 0;
    if (m == 0) {
        if (out_hit_offsets) out_hit_offsets[0] = 0;
        return GDX_OK;
    }
    if (!start || !end) fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
    for (uint64_t i = 0; i < m; i++)
        if (start[i] > end[i]) fail(GDX_ERR_INVALID_ARGUMENT, "cursor %llu has start > end", (unsigned long long)i);
    make_current();
    hipStream_t stream = hipStreamPerThread;
    DeviceBuffer<uint32_t> d_start(m), d_end(m);
    upload_narrowed(start, d_start.get(), m, n_, "cursor start", stream);
    upload_narrowed(end, d_end.get(), m, n_, "cursor end", stream);
    int rc = GDX_OK;
    locate_device(d_start.get(), d_end.get(), m, out_hit_offsets, hits, hits_capacity, out_total, &rc);
    return rc;
}

int FmIndex::rank_many(const uint8_t *symbols, const uint64_t *idx, uint64_t m, uint64_t *out) const
{
    if (m == 0) return GDX_OK;
    if (!symbols || !idx || !out) fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
    make_current();
    hipStream_t stream = hipStreamPerThread;
    DeviceBuffer<uint8_t> d_sym(m);
    DeviceBuffer<uint32_t> d_idx(m), d_out(m), d_err(1);
    GDX_HIP(hipMemcpyAsync(d_sym.get(), symbols, m, hipMemcpyHostToDevice, stream));
    upload_narrowed(idx, d_idx.get(), m, n_, "rank index", stream);  // mod.rs:107-108 idx <= text_len
    GDX_HIP(hipMemsetAsync(d_err.get(), 0, sizeof(uint32_t), stream));
    launch_rank_many(view_, d_sym.get(), d_idx.get(), m, d_out.get(), d_err.get(), stream);
    GDX_HIP(hipGetLastError());
    uint32_t err = 0;
    GDX_HIP(hipMemcpyAsync(&err, d_err.get(), sizeof(err), hipMemcpyDeviceToHost, stream));
    download_widened(d_out.get(), out, m, stream);
    if (err) fail(GDX_ERR_INVALID_ARGUMENT, "rank: symbol >= alphabet size or idx > text_len (mod.rs:107-108)");
    return GDX_OK;
}

int FmIndex::symbol_at_many(const uint64_t *idx, uint64_t m, uint8_t *out) const
{
    if (m == 0) return GDX_OK;
    if (!idx || !out) fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
    if (n_ == 0) fail(GDX_ERR_INVALID_ARGUMENT, "symbol_at: idx >= text_len (condensed.rs:344)");
    make_current();
    hipStream_t stream = hipStreamPerThread;
    DeviceBuffer<uint32_t> d_idx(m), d_err(1);
    DeviceBuffer<uint8_t> d_out(m);
    upload_narrowed(idx, d_idx.get(), m, n_ - 1, "symbol_at index", stream);
    GDX_HIP(hipMemsetAsync(d_err.get(), 0, sizeof(uint32_t), stream));
    launch_symbol_at_many(view_, d_idx.get(), m, d_out.get(), d_err.get(), stream);
    GDX_HIP(hipGetLastError());
    GDX_HIP(hipMemcpyAsync(out, d_out.get(), m, hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    return GDX_OK;
}

// =============================================================================================
// persistence: header + the reference's logical arrays

void FmIndex::save(const char *path) const
{
    if (!path) fail(GDX_ERR_INVALID_ARGUMENT, "path is null");
    const int nbits = view_.nbits;
    const uint64_t len = n_ + 1, n_blocks = div_ceil(len, 64);
    FileHeader h{};
    std::memcpy(h.magic, kIndexFileMagic, 8);
    h.n = n_;
    h.n_texts = n_texts_;
    h.sa_rate = cfg_.sa_rate;
    h.n_plane_words = n_blocks * nbits;
    h.n_samples = div_ceil(n_, cfg_.sa_rate);
    h.sigma = cfg_.sigma;
    h.n_searchable = cfg_.n_searchable;
    h.lookup_depth = cfg_.lookup_depth;
    h.index_width = cfg_.index_width;
    std::memcpy(h.io_to_dense, cfg_.io_to_dense, 256);
    std::vector<uint64_t> planes(h.n_plane_words);
    std::vector<uint16_t> bo(n_blocks * cfg_.sigma);
    std::vector<uint32_t> sbo(div_ceil(len, 65536) * cfg_.sigma), samples(h.n_samples);
    export_condensed_table(planes.data(), bo.data(), sbo.data());
    export_sa_samples(samples.data());
    IndexFile out(path, "wb");
    out.write(&h, sizeof(h));
    out.write(count_host_.data(), count_host_.size() * sizeof(uint64_t));
    out.write(sentinels_host_.data(), n_texts_ * sizeof(uint64_t));
    out.write(border_keys_host_.data(), n_texts_ * sizeof(uint64_t));
    out.write(border_vals_host_.data(), n_texts_ * sizeof(uint64_t));
    out.write(samples.data(), samples.size() * sizeof(uint32_t));
    out.write(planes.data(), planes.size() * sizeof(uint64_t));
}

std::unique_ptr<FmIndex> FmIndex::load(const char *path, int device_id, const BuildOptions &build)
{
    if (!path) fail(GDX_ERR_INVALID_ARGUMENT, "path is null");
    IndexFile in(path, "rb");
    const FileHeader h = read_index_header(in, path);
    IndexConfig cfg;
    std::memcpy(cfg.io_to_dense, h.io_to_dense, 256);
    cfg.sigma = h.sigma;
    cfg.n_searchable = h.n_searchable;
    cfg.sa_rate = h.sa_rate;
    cfg.lookup_depth = h.lookup_depth;
    cfg.index_width = h.index_width;
    cfg.device_id = device_id;
    cfg.build = build;
    std::vector<uint64_t> count(h.sigma + 1), sent(h.n_texts), bk(h.n_texts), bv(h.n_texts), planes(h.n_plane_words);
    std::vector<uint32_t> samples(h.n_samples);
    in.read(count.data(), count.size() * sizeof(uint64_t));
    in.read(sent.data(), sent.size() * sizeof(uint64_t));
    in.read(bk.data(), bk.size() * sizeof(uint64_t));
    in.read(bv.data(), bv.size() * sizeof(uint64_t));
    in.read(samples.data(), samples.size() * sizeof(uint32_t));
    in.read(planes.data(), planes.size() * sizeof(uint64_t));
    return from_parts(0, 64, count.data(), planes.data(), h.n, samples.data(), bk.data(), bv.data(), sent.data(),
                      h.n_texts, cfg);
}

// =============================================================================================
// exports

void FmIndex::export_count(uint64_t *count) const { std::memcpy(count, count_host_.data(), count_host_.size() * sizeof(uint64_t)); }

void FmIndex::export_bwt(uint8_t *bwt) const
{
    if (n_ == 0) return;
    make_current();
    hipStream_t stream = hipStreamPerThread;
    DeviceBuffer<uint8_t> d(n_);
    if (view_.layout == 0)
        hipLaunchKernelGGL(decode_bwt_kernel<LineTable>, dim3(grid_for_items(n_)), dim3(kBlock), 0, stream, view_, d.get());
    else
        hipLaunchKernelGGL(decode_bwt_kernel<GenericTable>, dim3(grid_for_items(n_)), dim3(kBlock), 0, stream, view_, d.get());
    GDX_HIP(hipMemcpyAsync(bwt, d.get(), n_, hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
}

void FmIndex::export_sa_samples(uint32_t *samples) const
{
    make_current();
    GDX_HIP(hipMemcpy(samples, sa_samples_.get(), sa_samples_.bytes(), hipMemcpyDeviceToHost));
}

void FmIndex::export_borders(uint64_t *keys, uint64_t *vals) const
{
    std::memcpy(keys, border_keys_host_.data(), n_texts_ * sizeof(uint64_t));
    std::memcpy(vals, border_vals_host_.data(), n_texts_ * sizeof(uint64_t));
}

void FmIndex::export_sentinel_indices(uint64_t *out) const { std::memcpy(out, sentinels_host_.data(), n_texts_ * sizeof(uint64_t)); }

void FmIndex::export_lookup_table(int depth, uint32_t *pairs) const
{
    if (depth < 0 || depth > cfg_.lookup_depth) fail(GDX_ERR_INVALID_ARGUMENT, "lookup table depth out of range");
    make_current();
    const uint64_t entries = lookup_off_host_[depth + 1] - lookup_off_host_[depth];
    GDX_HIP(hipMemcpy(pairs, lookup_.get() + lookup_off_host_[depth], entries * sizeof(uint2), hipMemcpyDeviceToHost));
}

void FmIndex::export_condensed_table(uint64_t *blocks, uint16_t *block_offsets, uint32_t *superblock_offsets) const
{
    make_current();
    hipStream_t stream = hipStreamPerThread;
    const int sigma = cfg_.sigma, nbits = view_.nbits;
    const uint64_t len = n_ + 1, n_blocks = div_ceil(len, 64), n_sb = div_ceil(len, 65536);
    if (view_.layout == 1 && view_.g_kind == 0u && view_.g_wpb == 1u) {  // Condensed / Block64 as it is
        GDX_HIP(hipMemcpy(blocks, g_planes_.get(), g_planes_.bytes(), hipMemcpyDeviceToHost));
        GDX_HIP(hipMemcpy(block_offsets, g_block_off_.get(), g_block_off_.bytes(), hipMemcpyDeviceToHost));
        GDX_HIP(hipMemcpy(superblock_offsets, sb_offsets_.get(), sb_offsets_.bytes(), hipMemcpyDeviceToHost));
        return;
    }
    DeviceBuffer<uint8_t> d_bwt(n_ ? n_ : 1);
    if (view_.layout == 1)  // (another of the reference's variants: through the BWT)
        hipLaunchKernelGGL(decode_bwt_kernel<GenericTable>, dim3(grid_for_items(n_)), dim3(kBlock), 0, stream, view_, d_bwt.get());
    else
        hipLaunchKernelGGL(decode_bwt_kernel<LineTable>, dim3(grid_for_items(n_)), dim3(kBlock), 0, stream, view_, d_bwt.get());
    DeviceBuffer<uint64_t> d_planes(n_blocks * nbits);
    DeviceBuffer<uint16_t> d_bo(n_blocks * sigma);
    DeviceBuffer<uint32_t> d_sb(n_sb * sigma);
    hipLaunchKernelGGL(build_planes_kernel, dim3(grid_for_items(n_blocks)), dim3(kBlock), 0, stream, d_bwt.get(), n_,
                       n_blocks, nbits, d_planes.get());
    hipLaunchKernelGGL(build_generic_offsets_kernel, dim3(grid_for_items(n_sb * sigma)), dim3(kBlock), 0, stream,
                       d_bwt.get(), n_, n_blocks, n_sb, sigma, d_bo.get(), d_sb.get());
    hipLaunchKernelGGL(superblock_prefix_kernel, dim3(1), dim3(256), 0, stream, d_sb.get(), n_sb,
                       static_cast<uint32_t>(sigma));
    GDX_HIP(hipMemcpyAsync(blocks, d_planes.get(), d_planes.bytes(), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipMemcpyAsync(block_offsets, d_bo.get(), d_bo.bytes(), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipMemcpyAsync(superblock_offsets, d_sb.get(), d_sb.bytes(), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    GDX_HIP(hipGetLastError());
}

// the occurrence table of a reference_table_layout index as it sits in HBM: interleaved blocks (the flat variants carry their
// block offsets inside), u32 superblock offsets; returns the number of 64-bit words / offsets through n_words / n_sb
void FmIndex::export_reference_table(uint64_t *blocks, uint64_t capacity_words, uint64_t *n_words, uint32_t *superblock_offsets,
                                     uint64_t capacity_sb, uint64_t *n_sb) const
{
    make_current();
    if (view_.layout != 1) fail(GDX_ERR_UNSUPPORTED, "the index was not built with reference_table_layout");
    *n_words = g_planes_.bytes() / sizeof(uint64_t);
    *n_sb = sb_offsets_.bytes() / sizeof(uint32_t);
    if (blocks != nullptr) {
        if (capacity_words < *n_words) fail(GDX_ERR_CAPACITY, "export_reference_table: block buffer too small");
        GDX_HIP(hipMemcpy(blocks, g_planes_.get(), g_planes_.bytes(), hipMemcpyDeviceToHost));
    }
    if (superblock_offsets != nullptr) {
        if (capacity_sb < *n_sb) fail(GDX_ERR_CAPACITY, "export_reference_table: superblock buffer too small");
        GDX_HIP(hipMemcpy(superblock_offsets, sb_offsets_.get(), sb_offsets_.bytes(), hipMemcpyDeviceToHost));
    }
}

}  // namespace gdx
