// wide.hip -- index storage beyond 32 bits: the reference's `IndexStorage for i64` (construction/mod.rs:71-154, :225-252).
//
// A text collection of more than 2^32 - 1 symbols (sentinels included) does not fit the 32-bit rows, targets and counts of
// the engine in search.hip / locate.hip.  This file is a second, plain engine for such indexes: 64-bit rows and text
// positions throughout, the reference's own arrays only -- rank lines (the condensed table, two Block64 per line) with
// u64 superblock offsets, the sampled suffix array as u64, the border map, the sentinel positions -- and one lane per
// query / per hit.  It is the functional equivalent of genedex's i64 index (same intervals, counts, hits and hit order
// as the reference's algorithm; no lookup tables deeper than 0, DNA-sized alphabets), not a fast path: at n = 2^32 + 2^20
// it answers ~0.5 G queries/s where the 32-bit engine does 12 on a text that fits it.  A collection that splits at
// text borders is better served by the partitioned index (parts.hip).
//
// Construction: suffix array by plain prefix doubling on 64-bit ranks -- radix sort of the 21-symbol keys, then rounds that
// sort all suffixes by (rank[i], rank[i + h]) with two stable radix sorts (least significant key first) -- then BWT,
// samples, borders (bwt.rs:93-116, sampled_suffix_array.rs:37-43) and the rank lines of fm_index.hip.
#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include <sys/mman.h>

#include "fm_index.hpp"
#include "kernels.hpp"

namespace gdx {

namespace {

constexpr int kBlock = 256;

unsigned wgrid(uint64_t items, uint64_t cap = 256u * 16u)
{
    const uint64_t blocks = (items + kBlock - 1) / kBlock;
    return static_cast<unsigned>(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

struct WideView {
    const u32x4 *lines;        // rank lines (layout.hpp): [ceil((n + 1) / 128)][4]
    const uint64_t *sb;        // [n_superblocks][8] occurrences before the superblock
    const uint64_t *count;     // [sigma + 1] (lib.rs:95)
    const uint64_t *samples;   // SA[i] for i % sa_rate == 0
    const uint64_t *border_keys, *border_vals, *sentinels;
    const uint8_t *io_to_dense;
    uint64_t n, n_texts, sa_rate;
    int32_t sigma;
};

__device__ __forceinline__ uint64_t lower_bound_u64(const uint64_t *a, uint64_t n, uint64_t key)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ RankLine wide_line(const WideView &v, uint64_t idx)
{
    const u32x4 *p = v.lines + ((idx >> kLineShift) << 2);
    RankLine l;
    l.c0 = p[0];
    l.c1 = p[1];
    l.c2 = p[2];
    l.c3 = p[3];
    return l;
}

// rank(symbol, idx) = #symbol in bwt[0..idx)   (condensed.rs:291-341 with a 64-bit superblock offset)
__device__ __forceinline__ uint64_t wide_rank(const WideView &v, uint32_t symbol, uint64_t idx)
{
    const RankLine l = wide_line(v, idx);
    return v.sb[(idx >> kSuperblockShift) * 8u + symbol] + line_block_offset(l, symbol) +
           line_popcount(l, symbol, static_cast<uint32_t>(idx & 127u));
}

// ---- queries ---------------------------------------------------------------------------------------------------------

// lib.rs:217-235 cursor_for_query, one lane per query (lookup depth 0): intervals, statuses
__global__ __launch_bounds__(kBlock) void wide_search_kernel(WideView v, const uint8_t *__restrict__ qbuf,
                                                             const uint64_t *__restrict__ qoff, uint64_t nq,
                                                             uint64_t *__restrict__ out_start, uint64_t *__restrict__ out_end,
                                                             uint8_t *__restrict__ out_status)
{
    __shared__ uint8_t s_dense[256];
    for (int i = threadIdx.x; i < 256; i += kBlock) s_dense[i] = v.io_to_dense[i];
    __syncthreads();
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; q < nq; q += stride) {
        const uint64_t begin = qoff[q], end = qoff[q + 1];
        uint64_t lo = 0, hi = v.n;
        uint32_t status = GDX_Q_OK;
        for (uint64_t pos = end; pos > begin && lo != hi; pos--) {  // lib.rs:226-232: stop at the empty interval
            const uint32_t c = s_dense[qbuf[pos - 1]];
            if (c == 0) {  // alphabet.rs:195-198
                status = GDX_Q_INVALID_SYMBOL;
                lo = hi = 0;
                break;
            }
            const uint64_t cc = v.count[c];  // lib.rs:273-275
            lo = cc + wide_rank(v, c, lo);
            hi = cc + wide_rank(v, c, hi);
        }
        out_start[q] = lo;
        out_end[q] = hi;
        out_status[q] = static_cast<uint8_t>(status);
    }
}

__global__ __launch_bounds__(kBlock) void wide_mark_heads_kernel(const uint64_t *__restrict__ start,
                                                                 const uint64_t *__restrict__ end, uint64_t m,
                                                                 const uint64_t *__restrict__ hit_offsets,
                                                                 uint32_t *__restrict__ heads)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; q < m; q += stride)
        if (end[q] != start[q]) heads[hit_offsets[q]] = static_cast<uint32_t>(q) + 1u;
}

// sampled_suffix_array.rs:110-138 + text_id_search_tree.rs:35-64, one lane per hit
__global__ __launch_bounds__(kBlock) void wide_locate_kernel(WideView v, const uint64_t *__restrict__ start,
                                                             const uint64_t *__restrict__ hit_offsets,
                                                             const uint32_t *__restrict__ query_of_hit, uint64_t total,
                                                             gdx_hit_t *__restrict__ hits)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t h = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; h < total; h += stride) {
        const uint32_t q = query_of_hit[h] - 1u;
        uint64_t i = start[q] + (h - hit_offsets[q]), steps = 0, pos;
        for (;;) {
            if (i % v.sa_rate == 0) {  // :118, :133-136
                pos = v.samples[i / v.sa_rate] + steps;
                break;
            }
            const RankLine l = wide_line(v, i);
            const uint32_t c = line_symbol_at(l, static_cast<uint32_t>(i & 127u));
            if (c == 0) {  // :121-126 BWT sentinel: the walk reached the start of a text
                pos = v.border_vals[lower_bound_u64(v.border_keys, v.n_texts, i)] + steps;
                break;
            }
            i = v.count[c] + v.sb[(i >> kSuperblockShift) * 8u + c] + line_block_offset(l, c) +
                line_popcount(l, c, static_cast<uint32_t>(i & 127u));
            steps++;
        }
        const uint64_t t = lower_bound_u64(v.sentinels, v.n_texts, pos);
        gdx_hit_t out;
        out.text_id = t;
        out.position = t == 0 ? pos : pos - v.sentinels[t - 1] - 1u;
        hits[h] = out;
    }
}

// Cursor::extend_query_front for m independent cursors (cursor.rs:34-51), one lane each
__global__ __launch_bounds__(kBlock) void wide_extend_front_kernel(WideView v, uint64_t *__restrict__ start, uint64_t *__restrict__ end,
                                                                   const uint8_t *__restrict__ io_symbols, uint64_t m,
                                                                   uint8_t *__restrict__ out_status)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < m; i += stride) {
        const uint64_t lo = start[i], hi = end[i];
        const uint32_t c = v.io_to_dense[io_symbols[i]];  // cursor.rs:34-38: translated before anything else
        uint32_t status = GDX_Q_OK;
        if (c == 0) {
            status = GDX_Q_INVALID_SYMBOL;
        } else if (lo != hi) {  // cursor.rs:41-48
            const uint64_t cc = v.count[c];
            start[i] = cc + wide_rank(v, c, lo);
            end[i] = cc + wide_rank(v, c, hi);
        }
        out_status[i] = static_cast<uint8_t>(status);
    }
}

// TextWithRankSupport::rank / symbol_at (text_with_rank_support/mod.rs:106-110, condensed.rs:343-362)
__global__ __launch_bounds__(kBlock) void wide_rank_kernel(WideView v, const uint8_t *__restrict__ symbols, const uint64_t *__restrict__ idx,
                                                           uint64_t m, uint64_t *__restrict__ out, uint32_t *__restrict__ error)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < m; i += stride) {
        const uint32_t c = symbols ? symbols[i] : 0u;
        const uint64_t p = idx[i];
        if (symbols) {
            if (c >= static_cast<uint32_t>(v.sigma) || p > v.n) {  // mod.rs:107-108
                *error = 1;
                out[i] = 0;
            } else {
                out[i] = wide_rank(v, c, p);
            }
        } else if (p >= v.n) {  // condensed.rs:344
            *error = 1;
            out[i] = 0;
        } else {
            out[i] = line_symbol_at(wide_line(v, p), static_cast<uint32_t>(p & 127u));
        }
    }
}

struct WideSize {
    const uint64_t *start, *end;
    uint64_t m;
    __host__ __device__ uint64_t operator()(uint64_t q) const { return q < m ? end[q] - start[q] : 0ull; }
};
using WideSizeIterator = rocprim::transform_iterator<rocprim::counting_iterator<uint64_t>, WideSize, uint64_t>;

// ---- construction ------------------------------------------------------------------------------------------------------

// construction/mod.rs:255-308: concatenate + densely encode, one sentinel (0) after every text; frequency table
__global__ __launch_bounds__(kBlock) void wide_encode_kernel(const uint8_t *__restrict__ io_text,
                                                             const uint64_t *__restrict__ sentinels, uint64_t n_texts, uint64_t n,
                                                             const uint8_t *__restrict__ io_to_dense, uint8_t *__restrict__ dense,
                                                             unsigned long long *__restrict__ hist, uint32_t *__restrict__ error)
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
        const uint64_t t = lower_bound_u64(sentinels, n_texts, p);
        uint8_t d = 0;
        if (sentinels[t] != p) {
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

// keys of the first k0 symbols ((symbol + 1) in `bits` bits each, most significant first; beyond the end 0) and pos[i] = i
__global__ __launch_bounds__(kBlock) void wide_first_keys_kernel(const uint8_t *__restrict__ text, uint64_t n, int k0, int bits,
                                                                 uint64_t *__restrict__ keys, uint64_t *__restrict__ pos)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        uint64_t key = 0;
        for (int t = 0; t < k0; t++) {
            const uint64_t p = i + t;
            key = (key << bits) | (p < n ? static_cast<uint64_t>(text[p]) + 1u : 0u);
        }
        keys[i] = key;
        pos[i] = i;
    }
}

// marks[j] = j + 1 where the sorted key changes (a group opens), else 0; *n_groups += heads
__global__ __launch_bounds__(kBlock) void wide_mark_keys_kernel(const uint64_t *__restrict__ keys, uint64_t n,
                                                                uint64_t *__restrict__ marks, unsigned long long *__restrict__ n_groups)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    unsigned long long heads = 0;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; j < n; j += stride) {
        const bool head = j == 0 || keys[j] != keys[j - 1];
        marks[j] = head ? j + 1 : 0;
        heads += head ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) heads += __shfl_xor(heads, off);
    if ((threadIdx.x & 63u) == 0 && heads) atomicAdd(n_groups, heads);
}

// rank[sa[j]] = group_of[j] - 1 (the first slot of j's group)
__global__ __launch_bounds__(kBlock) void wide_scatter_rank_kernel(const uint64_t *__restrict__ sa, const uint64_t *__restrict__ group_of,
                                                                   uint64_t n, uint64_t *__restrict__ rank)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; j < n; j += stride) rank[sa[j]] = group_of[j] - 1;
}

// keys[j] = rank[sa[j] + h] + 1 (0 beyond the end: the shorter suffix sorts first)
__global__ __launch_bounds__(kBlock) void wide_second_keys_kernel(const uint64_t *__restrict__ sa, const uint64_t *__restrict__ rank,
                                                                  uint64_t n, uint64_t h, uint64_t *__restrict__ keys)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; j < n; j += stride) {
        const uint64_t p = sa[j] + h;
        keys[j] = p < n ? rank[p] + 1 : 0;
    }
}

__global__ __launch_bounds__(kBlock) void wide_first_rank_keys_kernel(const uint64_t *__restrict__ sa, const uint64_t *__restrict__ rank,
                                                                      uint64_t n, uint64_t *__restrict__ keys)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; j < n; j += stride) keys[j] = rank[sa[j]];
}

// after the two sorts: first[j] = rank[sa[j]] (sorted); a group opens where (first, second) changes
__global__ __launch_bounds__(kBlock) void wide_mark_pairs_kernel(const uint64_t *__restrict__ first, const uint64_t *__restrict__ sa,
                                                                 const uint64_t *__restrict__ rank, uint64_t n, uint64_t h,
                                                                 uint64_t *__restrict__ marks, unsigned long long *__restrict__ n_groups)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    unsigned long long heads = 0;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; j < n; j += stride) {
        bool head = j == 0 || first[j] != first[j - 1];
        if (!head) {
            const uint64_t a = sa[j] + h, b = sa[j - 1] + h;
            const uint64_t ra = a < n ? rank[a] + 1 : 0, rb = b < n ? rank[b] + 1 : 0;
            head = ra != rb;
        }
        marks[j] = head ? j + 1 : 0;
        heads += head ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) heads += __shfl_xor(heads, off);
    if ((threadIdx.x & 63u) == 0 && heads) atomicAdd(n_groups, heads);
}

int bits_for(uint64_t values)  // bits needed to hold 0 .. values - 1
{
    int b = 1;
    while (b < 64 && (1ull << b) < values) b++;
    return b;
}

// d_sa[0 .. n) = suffix array of d_text (symbols < sigma; the end of the string compares smallest)
void wide_suffix_array(const uint8_t *d_text, uint64_t n, int sigma, uint64_t *d_sa, hipStream_t stream, uint64_t *rounds_out)
{
    const int sym_bits = bits_for(static_cast<uint64_t>(sigma) + 1);
    const int k0 = 64 / sym_bits > 32 ? 32 : 64 / sym_bits;
    DeviceBuffer<uint64_t> keys_a(n), keys_b(n), vals_b(n), rank(n);
    DeviceBuffer<unsigned long long> d_groups(1);
    uint64_t *vals_a = d_sa;  // the suffix array lives in one of the two value buffers; it ends up in d_sa (see below)
    size_t sort_tmp = 0, scan_tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, keys_a.get(), keys_b.get(), vals_a, vals_b.get(), static_cast<size_t>(n), 0u, 64u);
    (void)rocprim::inclusive_scan(nullptr, scan_tmp, keys_a.get(), keys_a.get(), static_cast<size_t>(n), rocprim::maximum<uint64_t>());
    DeviceBuffer<uint8_t> temp(sort_tmp > scan_tmp ? sort_tmp : scan_tmp);
    const unsigned grid = wgrid(n);
    auto groups_of = [&](uint64_t *marks) {  // marks -> group head slot + 1 (max-scan); -> number of groups
        size_t bytes = temp.bytes();
        GDX_HIP(rocprim::inclusive_scan(temp.get(), bytes, marks, marks, static_cast<size_t>(n), rocprim::maximum<uint64_t>(), stream));
        unsigned long long g = 0;
        GDX_HIP(hipMemcpyAsync(&g, d_groups.get(), sizeof(g), hipMemcpyDeviceToHost, stream));
        GDX_HIP(hipStreamSynchronize(stream));
        return static_cast<uint64_t>(g);
    };
    // order by the first k0 symbols
    hipLaunchKernelGGL(wide_first_keys_kernel, dim3(grid), dim3(kBlock), 0, stream, d_text, n, k0, sym_bits, keys_a.get(), vals_a);
    size_t bytes = temp.bytes();
    GDX_HIP(rocprim::radix_sort_pairs(temp.get(), bytes, keys_a.get(), keys_b.get(), vals_a, vals_b.get(), static_cast<size_t>(n), 0u,
                                      static_cast<unsigned>(k0 * sym_bits), stream));
    uint64_t *sa = vals_b.get(), *other = vals_a;  // current order / the spare value buffer
    GDX_HIP(hipMemsetAsync(d_groups.get(), 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(wide_mark_keys_kernel, dim3(grid), dim3(kBlock), 0, stream, keys_b.get(), n, keys_a.get(), d_groups.get());
    uint64_t groups = groups_of(keys_a.get());
    hipLaunchKernelGGL(wide_scatter_rank_kernel, dim3(grid), dim3(kBlock), 0, stream, sa, keys_a.get(), n, rank.get());
    const unsigned rank_bits = static_cast<unsigned>(bits_for(n + 2));
    uint64_t rounds = 0;
    for (uint64_t h = static_cast<uint64_t>(k0); groups < n; h *= 2, rounds++) {
        // sort by (rank[i], rank[i + h]): least significant key first, both sorts stable
        hipLaunchKernelGGL(wide_second_keys_kernel, dim3(grid), dim3(kBlock), 0, stream, sa, rank.get(), n, h, keys_a.get());
        bytes = temp.bytes();
        GDX_HIP(rocprim::radix_sort_pairs(temp.get(), bytes, keys_a.get(), keys_b.get(), sa, other, static_cast<size_t>(n), 0u, rank_bits, stream));
        std::swap(sa, other);
        hipLaunchKernelGGL(wide_first_rank_keys_kernel, dim3(grid), dim3(kBlock), 0, stream, sa, rank.get(), n, keys_a.get());
        bytes = temp.bytes();
        GDX_HIP(rocprim::radix_sort_pairs(temp.get(), bytes, keys_a.get(), keys_b.get(), sa, other, static_cast<size_t>(n), 0u, rank_bits, stream));
        std::swap(sa, other);
        GDX_HIP(hipMemsetAsync(d_groups.get(), 0, sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(wide_mark_pairs_kernel, dim3(grid), dim3(kBlock), 0, stream, keys_b.get(), sa, rank.get(), n, h, keys_a.get(),
                           d_groups.get());
        groups = groups_of(keys_a.get());
        hipLaunchKernelGGL(wide_scatter_rank_kernel, dim3(grid), dim3(kBlock), 0, stream, sa, keys_a.get(), n, rank.get());
        if (h > n) break;  // (cannot happen: distinct suffixes differ within n symbols)
    }
    if (sa != d_sa) GDX_HIP(hipMemcpyAsync(d_sa, sa, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    if (rounds_out) *rounds_out = rounds;
}

// bwt.rs:93-116 + sampled_suffix_array.rs:37-43 on a 64-bit suffix array
__global__ __launch_bounds__(kBlock) void wide_bwt_kernel(const uint8_t *__restrict__ text, const uint64_t *__restrict__ sa, uint64_t n,
                                                          uint64_t rate, uint8_t *__restrict__ bwt, uint64_t *__restrict__ samples,
                                                          uint64_t *__restrict__ border_keys, uint64_t *__restrict__ border_vals,
                                                          unsigned long long *__restrict__ n_borders)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; j < n; j += stride) {
        const uint64_t s = sa[j];
        const uint8_t b = text[(s > 0 ? s : n) - 1];
        bwt[j] = b;
        if (j % rate == 0) samples[j / rate] = s;
        if (b == 0) {
            const unsigned long long at = atomicAdd(n_borders, 1ull);
            border_keys[at] = j;
            border_vals[at] = s;
        }
    }
}

// exclusive prefix over the superblocks, one thread per symbol column (condensed.rs:104-115), into 64-bit offsets
__global__ void wide_sb_prefix_kernel(const uint32_t *__restrict__ totals, uint64_t n_sb, uint64_t *__restrict__ sb)
{
    const uint32_t c = threadIdx.x;
    if (c >= 8) return;
    uint64_t sum = 0;
    for (uint64_t s = 0; s < n_sb; s++) {
        sb[s * 8 + c] = sum;
        sum += totals[s * 8 + c];
    }
}

}  // namespace

struct WideIndex::Impl {
    IndexConfig cfg;
    uint64_t n = 0, n_texts = 0;
    std::vector<uint64_t> count_host, sentinels_host;
    DeviceBuffer<u32x4> lines;
    DeviceBuffer<uint64_t> sb, count, samples, border_keys, border_vals, sentinels;
    DeviceBuffer<uint8_t> io_to_dense;
    WideView view{};
};

WideIndex::WideIndex() : p_(new Impl) {}
WideIndex::~WideIndex() = default;
const IndexConfig &WideIndex::config() const { return p_->cfg; }
uint64_t WideIndex::total_text_len() const { return p_->n; }
uint64_t WideIndex::num_texts() const { return p_->n_texts; }
uint64_t WideIndex::device_bytes() const
{
    return p_->lines.bytes() + p_->sb.bytes() + p_->count.bytes() + p_->samples.bytes() + p_->border_keys.bytes() + p_->border_vals.bytes() +
           p_->sentinels.bytes() + p_->io_to_dense.bytes();
}

std::unique_ptr<WideIndex> WideIndex::construct_index(const uint8_t *texts_buf, bool texts_on_device, const uint64_t *text_offsets,
                                                      uint64_t n_texts, const IndexConfig &cfg)
{
    if (n_texts == 0) fail(GDX_ERR_INVALID_ARGUMENT, "There should be at least one text (construction/mod.rs:303)");
    if (!text_offsets) fail(GDX_ERR_INVALID_ARGUMENT, "text_offsets is null");
    if (cfg.sigma < 2 || cfg.sigma > 8) fail(GDX_ERR_UNSUPPORTED, "64-bit index storage: alphabets of up to 7 symbols + sentinel (rank lines)");
    if (cfg.lookup_depth != 0) fail(GDX_ERR_UNSUPPORTED, "64-bit index storage: lookup_table_depth must be 0");
    if (cfg.sa_rate == 0) fail(GDX_ERR_INVALID_ARGUMENT, "suffix_array_sampling_rate must be > 0 (config.rs:28)");
    for (int b = 0; b < 256; b++)
        if (cfg.io_to_dense[b] >= cfg.sigma) fail(GDX_ERR_INVALID_ARGUMENT, "io_to_dense[%d] is not a dense symbol", b);
    for (uint64_t t = 0; t < n_texts; t++)
        if (text_offsets[t + 1] < text_offsets[t]) fail(GDX_ERR_INVALID_ARGUMENT, "text_offsets must be non-decreasing");
    const uint64_t io_len = text_offsets[n_texts] - text_offsets[0];
    if (io_len > 0 && !texts_buf) fail(GDX_ERR_INVALID_ARGUMENT, "texts_buf is null");
    const uint64_t n = io_len + n_texts;
    std::unique_ptr<WideIndex> ix(new WideIndex());
    Impl &p = *ix->p_;
    p.cfg = cfg;
    p.n = n;
    p.n_texts = n_texts;
    GDX_HIP(hipSetDevice(cfg.device_id));
    hipStream_t stream = hipStreamPerThread;
    p.sentinels_host.resize(n_texts);
    for (uint64_t t = 0; t < n_texts; t++) p.sentinels_host[t] = (text_offsets[t + 1] - text_offsets[0]) + t;  // mod.rs:266-273
    p.sentinels.alloc(n_texts);
    GDX_HIP(hipMemcpy(p.sentinels.get(), p.sentinels_host.data(), n_texts * sizeof(uint64_t), hipMemcpyHostToDevice));
    p.io_to_dense.alloc(256);
    GDX_HIP(hipMemcpy(p.io_to_dense.get(), cfg.io_to_dense, 256, hipMemcpyHostToDevice));

    // encode + concatenate + frequency table
    DeviceBuffer<uint8_t> io_owned;
    const uint8_t *d_io = texts_buf ? texts_buf + (texts_on_device ? text_offsets[0] : 0) : nullptr;
    if (!texts_on_device) {
        io_owned.alloc(io_len ? io_len : 1);
        if (io_len) GDX_HIP(hipMemcpy(io_owned.get(), texts_buf + text_offsets[0], io_len, hipMemcpyHostToDevice));
        d_io = io_owned.get();
    }
    DeviceBuffer<uint8_t> d_text(n);
    DeviceBuffer<unsigned long long> d_hist(256);
    DeviceBuffer<uint32_t> d_flag(1);
    GDX_HIP(hipMemsetAsync(d_hist.get(), 0, 256 * sizeof(unsigned long long), stream));
    GDX_HIP(hipMemsetAsync(d_flag.get(), 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(wide_encode_kernel, dim3(wgrid(n)), dim3(kBlock), 0, stream, d_io, p.sentinels.get(), n_texts, n,
                       p.io_to_dense.get(), d_text.get(), d_hist.get(), d_flag.get());
    unsigned long long hist[256];
    uint32_t flag = 0;
    GDX_HIP(hipMemcpyAsync(hist, d_hist.get(), sizeof(hist), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipMemcpyAsync(&flag, d_flag.get(), sizeof(flag), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    if (flag) fail(GDX_ERR_INVALID_ARGUMENT, "a text holds a symbol that is not in the alphabet (alphabet.rs:195-198)");
    io_owned.release();
    p.count_host.assign(cfg.sigma + 1, 0);  // construction/mod.rs:318-336
    for (int c = 0; c < cfg.sigma; c++) p.count_host[c + 1] = p.count_host[c] + hist[c];
    p.count.alloc(cfg.sigma + 1);
    GDX_HIP(hipMemcpy(p.count.get(), p.count_host.data(), (cfg.sigma + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));

    // suffix array, BWT, samples, borders
    const uint64_t n_lines = div_ceil(n + 1, 128), padded = n_lines * 128;
    DeviceBuffer<uint8_t> d_bwt(padded);
    GDX_HIP(hipMemsetAsync(d_bwt.get(), 0, padded, stream));
    const uint64_t n_samples = div_ceil(n, cfg.sa_rate);
    p.samples.alloc(n_samples ? n_samples : 1);
    p.border_keys.alloc(n_texts);
    p.border_vals.alloc(n_texts);
    {
        DeviceBuffer<uint64_t> d_sa(n);
        wide_suffix_array(d_text.get(), n, cfg.sigma, d_sa.get(), stream, nullptr);
        DeviceBuffer<unsigned long long> d_nb(1);
        DeviceBuffer<uint64_t> bk(n_texts), bv(n_texts);
        GDX_HIP(hipMemsetAsync(d_nb.get(), 0, sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(wide_bwt_kernel, dim3(wgrid(n)), dim3(kBlock), 0, stream, d_text.get(), d_sa.get(), n, cfg.sa_rate,
                           d_bwt.get(), p.samples.get(), bk.get(), bv.get(), d_nb.get());
        unsigned long long nb = 0;
        GDX_HIP(hipMemcpyAsync(&nb, d_nb.get(), sizeof(nb), hipMemcpyDeviceToHost, stream));
        GDX_HIP(hipStreamSynchronize(stream));
        if (nb != n_texts) fail(GDX_ERR_DEVICE, "internal: %llu BWT sentinels for %llu texts", nb, static_cast<unsigned long long>(n_texts));
        std::vector<uint64_t> hk(n_texts), hv(n_texts);  // the border map sorted by key (sampled_suffix_array.rs:121-126 looks keys up)
        GDX_HIP(hipMemcpy(hk.data(), bk.get(), n_texts * sizeof(uint64_t), hipMemcpyDeviceToHost));
        GDX_HIP(hipMemcpy(hv.data(), bv.get(), n_texts * sizeof(uint64_t), hipMemcpyDeviceToHost));
        std::vector<uint64_t> order(n_texts);
        for (uint64_t t = 0; t < n_texts; t++) order[t] = t;
        std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return hk[a] < hk[b]; });
        std::vector<uint64_t> sk(n_texts), sv(n_texts);
        for (uint64_t t = 0; t < n_texts; t++) {
            sk[t] = hk[order[t]];
            sv[t] = hv[order[t]];
        }
        GDX_HIP(hipMemcpy(p.border_keys.get(), sk.data(), n_texts * sizeof(uint64_t), hipMemcpyHostToDevice));
        GDX_HIP(hipMemcpy(p.border_vals.get(), sv.data(), n_texts * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    d_text.release();
    // occurrence table: rank lines + 64-bit superblock offsets
    const uint64_t n_sb = div_ceil(n + 1, 65536);
    p.lines.alloc(n_lines * 4);
    p.sb.alloc(n_sb * 8);
    {
        DeviceBuffer<uint32_t> totals(n_sb * 8);
        launch_build_lines(d_bwt.get(), n, n_lines, p.lines.get(), totals.get(), n_sb, stream);
        hipLaunchKernelGGL(wide_sb_prefix_kernel, dim3(1), dim3(64), 0, stream, totals.get(), n_sb, p.sb.get());
        GDX_HIP(hipStreamSynchronize(stream));
        GDX_HIP(hipGetLastError());
    }
    p.view = WideView{p.lines.get(), p.sb.get(), p.count.get(), p.samples.get(), p.border_keys.get(), p.border_vals.get(),
                      p.sentinels.get(), p.io_to_dense.get(), n, n_texts, cfg.sa_rate, cfg.sigma};
    return ix;
}

namespace {

// queries in chunks that bound the device memory of a call
constexpr uint64_t kWideChunkQueries = 1ull << 24, kWideChunkBytes = 1ull << 30;

template <class F>
void for_each_chunk(const uint64_t *qoff, uint64_t nq, F f)
{
    for (uint64_t q0 = 0; q0 < nq;) {
        uint64_t hi = std::min(nq, q0 + kWideChunkQueries);
        if (qoff[hi] - qoff[q0] > kWideChunkBytes) {
            const uint64_t *p = std::upper_bound(qoff + q0 + 1, qoff + hi + 1, qoff[q0] + kWideChunkBytes);
            hi = static_cast<uint64_t>(p - qoff) - 1;
            if (hi <= q0) hi = q0 + 1;
        }
        f(q0, hi);
        q0 = hi;
    }
}

void check_wide_queries(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq)
{
    if (!qoff) fail(GDX_ERR_INVALID_ARGUMENT, "qoff is null");
    for (uint64_t i = 0; i < nq; i++)
        if (qoff[i + 1] < qoff[i]) fail(GDX_ERR_INVALID_ARGUMENT, "qoff must be non-decreasing");
    if (nq && qoff[nq] > qoff[0] && !qbuf) fail(GDX_ERR_INVALID_ARGUMENT, "qbuf is null");
}

}  // namespace

int WideIndex::cursors_for_many_queries(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_start, uint64_t *out_end,
                                        uint64_t *out_count, uint8_t *out_status) const
{
    check_wide_queries(qbuf, qoff, nq);
    if (nq == 0) return GDX_OK;
    GDX_HIP(hipSetDevice(p_->cfg.device_id));
    hipStream_t stream = hipStreamPerThread;
    bool any = false;
    for_each_chunk(qoff, nq, [&](uint64_t q0, uint64_t q1) {
        const uint64_t m = q1 - q0, base = qoff[q0], bytes = qoff[q1] - base;
        std::vector<uint64_t> off(m + 1);
        for (uint64_t i = 0; i <= m; i++) off[i] = qoff[q0 + i] - base;
        DeviceBuffer<uint8_t> d_q(bytes + 16), d_st(m);
        DeviceBuffer<uint64_t> d_off(m + 1), d_s(m), d_e(m);
        if (bytes) GDX_HIP(hipMemcpyAsync(d_q.get(), qbuf + base, bytes, hipMemcpyHostToDevice, stream));
        GDX_HIP(hipMemcpyAsync(d_off.get(), off.data(), (m + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(wide_search_kernel, dim3(wgrid(m)), dim3(kBlock), 0, stream, p_->view, d_q.get(), d_off.get(), m, d_s.get(),
                           d_e.get(), d_st.get());
        GDX_HIP(hipGetLastError());
        std::vector<uint64_t> s(m), e(m);
        std::vector<uint8_t> st(m);
        GDX_HIP(hipMemcpyAsync(s.data(), d_s.get(), m * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        GDX_HIP(hipMemcpyAsync(e.data(), d_e.get(), m * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        GDX_HIP(hipMemcpyAsync(st.data(), d_st.get(), m, hipMemcpyDeviceToHost, stream));
        GDX_HIP(hipStreamSynchronize(stream));
        for (uint64_t i = 0; i < m; i++) {
            if (out_start) out_start[q0 + i] = s[i];
            if (out_end) out_end[q0 + i] = e[i];
            if (out_count) out_count[q0 + i] = e[i] - s[i];
            if (out_status) out_status[q0 + i] = st[i];
            any |= st[i] != 0;
        }
    });
    return any ? GDX_ERR_QUERY_STATUS : GDX_OK;
}

// d_s / d_e (m intervals on the device) -> hit offsets (host, shifted by hit_base) and hits; returns the number of hits
static uint64_t wide_locate_intervals(const WideView &view, const uint64_t *d_s, const uint64_t *d_e, uint64_t m, uint64_t hit_base,
                                      uint64_t *out_hit_offsets /* [m + 1] slots from the chunk's first, or null */, gdx_hit_t *hits,
                                      uint64_t hits_capacity, bool &fits, hipStream_t stream)
{
    DeviceBuffer<uint64_t> d_hoff(m + 1);
    WideSizeIterator sizes(rocprim::counting_iterator<uint64_t>(0), WideSize{d_s, d_e, m});
    size_t scan_bytes = 0;
    GDX_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, sizes, d_hoff.get(), uint64_t(0), static_cast<size_t>(m + 1),
                                    rocprim::plus<uint64_t>(), stream));
    DeviceBuffer<uint8_t> scan_tmp(scan_bytes ? scan_bytes : 1);
    GDX_HIP(rocprim::exclusive_scan(scan_tmp.get(), scan_bytes, sizes, d_hoff.get(), uint64_t(0), static_cast<size_t>(m + 1),
                                    rocprim::plus<uint64_t>(), stream));
    std::vector<uint64_t> hoff(m + 1);
    GDX_HIP(hipMemcpyAsync(hoff.data(), d_hoff.get(), (m + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    const uint64_t total = hoff[m];
    if (out_hit_offsets)
        for (uint64_t i = 0; i < m; i++) out_hit_offsets[i + 1] = hit_base + hoff[i + 1];
    if (total != 0 && hits != nullptr && hit_base + total <= hits_capacity && fits) {
        DeviceBuffer<uint32_t> heads(total);
        DeviceBuffer<gdx_hit_t> d_hits(total);
        GDX_HIP(hipMemsetAsync(heads.get(), 0, total * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(wide_mark_heads_kernel, dim3(wgrid(m)), dim3(kBlock), 0, stream, d_s, d_e, m, d_hoff.get(), heads.get());
        size_t max_bytes = 0;
        GDX_HIP(rocprim::inclusive_scan(nullptr, max_bytes, heads.get(), heads.get(), static_cast<size_t>(total), rocprim::maximum<uint32_t>(), stream));
        DeviceBuffer<uint8_t> max_tmp(max_bytes ? max_bytes : 1);
        GDX_HIP(rocprim::inclusive_scan(max_tmp.get(), max_bytes, heads.get(), heads.get(), static_cast<size_t>(total), rocprim::maximum<uint32_t>(), stream));
        hipLaunchKernelGGL(wide_locate_kernel, dim3(wgrid(total)), dim3(kBlock), 0, stream, view, d_s, d_hoff.get(), heads.get(), total, d_hits.get());
        GDX_HIP(hipGetLastError());
        GDX_HIP(hipMemcpyAsync(hits + hit_base, d_hits.get(), total * sizeof(gdx_hit_t), hipMemcpyDeviceToHost, stream));
        GDX_HIP(hipStreamSynchronize(stream));
    } else if (total != 0) {
        fits = false;
    }
    return total;
}

int WideIndex::cursor_locate_many(const uint64_t *start, const uint64_t *end, uint64_t m, uint64_t *out_hit_offsets, gdx_hit_t *hits,
                                  uint64_t hits_capacity, uint64_t *out_total) const
{
    if (out_total) *out_total = 0;
    if (out_hit_offsets) out_hit_offsets[0] = 0;
    if (m == 0) return GDX_OK;
    if (!start || !end) fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
    for (uint64_t i = 0; i < m; i++)
        if (start[i] > end[i] || end[i] > p_->n) fail(GDX_ERR_INVALID_ARGUMENT, "cursor %llu is not an interval of this index", static_cast<unsigned long long>(i));
    GDX_HIP(hipSetDevice(p_->cfg.device_id));
    hipStream_t stream = hipStreamPerThread;
    bool fits = true;
    uint64_t hit_base = 0;
    for (uint64_t q0 = 0; q0 < m; q0 += kWideChunkQueries) {
        const uint64_t c = std::min(kWideChunkQueries, m - q0);
        DeviceBuffer<uint64_t> d_s(c), d_e(c);
        GDX_HIP(hipMemcpyAsync(d_s.get(), start + q0, c * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        GDX_HIP(hipMemcpyAsync(d_e.get(), end + q0, c * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        hit_base += wide_locate_intervals(p_->view, d_s.get(), d_e.get(), c, hit_base, out_hit_offsets ? out_hit_offsets + q0 : nullptr, hits,
                                          hits ? hits_capacity : 0, fits, stream);
    }
    if (out_total) *out_total = hit_base;
    return hit_base > 0 && !fits ? GDX_ERR_CAPACITY : GDX_OK;
}

int WideIndex::cursor_extend_front_many(uint64_t *start, uint64_t *end, const uint8_t *io_symbols, uint64_t m, uint8_t *out_status) const
{
    if (m == 0) return GDX_OK;
    if (!start || !end || !io_symbols) fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
    GDX_HIP(hipSetDevice(p_->cfg.device_id));
    hipStream_t stream = hipStreamPerThread;
    DeviceBuffer<uint64_t> d_s(m), d_e(m);
    DeviceBuffer<uint8_t> d_sym(m), d_st(m);
    GDX_HIP(hipMemcpyAsync(d_s.get(), start, m * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    GDX_HIP(hipMemcpyAsync(d_e.get(), end, m * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    GDX_HIP(hipMemcpyAsync(d_sym.get(), io_symbols, m, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(wide_extend_front_kernel, dim3(wgrid(m)), dim3(kBlock), 0, stream, p_->view, d_s.get(), d_e.get(), d_sym.get(), m,
                       d_st.get());
    GDX_HIP(hipGetLastError());
    std::vector<uint8_t> st(m);
    GDX_HIP(hipMemcpyAsync(start, d_s.get(), m * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipMemcpyAsync(end, d_e.get(), m * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipMemcpyAsync(st.data(), d_st.get(), m, hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    bool any = false;
    for (uint64_t i = 0; i < m; i++) {
        if (out_status) out_status[i] = st[i];
        any |= st[i] != 0;
    }
    return any ? GDX_ERR_QUERY_STATUS : GDX_OK;
}

int WideIndex::rank_or_symbol_many(const uint8_t *symbols, const uint64_t *idx, uint64_t m, uint64_t *out) const
{
    if (m == 0) return GDX_OK;
    if (!idx || !out) fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
    GDX_HIP(hipSetDevice(p_->cfg.device_id));
    hipStream_t stream = hipStreamPerThread;
    DeviceBuffer<uint8_t> d_sym(symbols ? m : 1);
    DeviceBuffer<uint64_t> d_idx(m), d_out(m);
    DeviceBuffer<uint32_t> d_err(1);
    if (symbols) GDX_HIP(hipMemcpyAsync(d_sym.get(), symbols, m, hipMemcpyHostToDevice, stream));
    GDX_HIP(hipMemcpyAsync(d_idx.get(), idx, m * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    GDX_HIP(hipMemsetAsync(d_err.get(), 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(wide_rank_kernel, dim3(wgrid(m)), dim3(kBlock), 0, stream, p_->view, symbols ? d_sym.get() : nullptr, d_idx.get(), m,
                       d_out.get(), d_err.get());
    GDX_HIP(hipGetLastError());
    uint32_t err = 0;
    GDX_HIP(hipMemcpyAsync(&err, d_err.get(), sizeof(err), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipMemcpyAsync(out, d_out.get(), m * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    if (err) fail(GDX_ERR_INVALID_ARGUMENT, symbols ? "rank: symbol >= alphabet size or idx > text_len (mod.rs:107-108)"
                                                     : "symbol_at: idx >= text_len (condensed.rs:344)");
    return GDX_OK;
}

int WideIndex::locate_many(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets, gdx_hit_t *hits,
                           uint64_t hits_capacity, uint64_t *out_total, uint8_t *out_status) const
{
    check_wide_queries(qbuf, qoff, nq);
    if (out_total) *out_total = 0;
    if (out_hit_offsets) out_hit_offsets[0] = 0;
    if (nq == 0) return GDX_OK;
    GDX_HIP(hipSetDevice(p_->cfg.device_id));
    hipStream_t stream = hipStreamPerThread;
    bool any = false, fits = true;
    uint64_t hit_base = 0;
    for_each_chunk(qoff, nq, [&](uint64_t q0, uint64_t q1) {
        const uint64_t m = q1 - q0, base = qoff[q0], bytes = qoff[q1] - base;
        std::vector<uint64_t> off(m + 1);
        for (uint64_t i = 0; i <= m; i++) off[i] = qoff[q0 + i] - base;
        DeviceBuffer<uint8_t> d_q(bytes + 16), d_st(m);
        DeviceBuffer<uint64_t> d_off(m + 1), d_s(m), d_e(m);
        if (bytes) GDX_HIP(hipMemcpyAsync(d_q.get(), qbuf + base, bytes, hipMemcpyHostToDevice, stream));
        GDX_HIP(hipMemcpyAsync(d_off.get(), off.data(), (m + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(wide_search_kernel, dim3(wgrid(m)), dim3(kBlock), 0, stream, p_->view, d_q.get(), d_off.get(), m, d_s.get(),
                           d_e.get(), d_st.get());
        std::vector<uint8_t> st(m);
        GDX_HIP(hipMemcpyAsync(st.data(), d_st.get(), m, hipMemcpyDeviceToHost, stream));
        GDX_HIP(hipStreamSynchronize(stream));
        for (uint64_t i = 0; i < m; i++) {
            if (out_status) out_status[q0 + i] = st[i];
            any |= st[i] != 0;
        }
        const uint64_t total = wide_locate_intervals(p_->view, d_s.get(), d_e.get(), m, hit_base,
                                                     out_hit_offsets ? out_hit_offsets + q0 : nullptr, hits, hits_capacity, fits, stream);
        hit_base += total;
    });
    if (out_total) *out_total = hit_base;
    if (hit_base > 0 && !fits) return GDX_ERR_CAPACITY;
    return any ? GDX_ERR_QUERY_STATUS : GDX_OK;
}

int WideIndex::locate_many_alloc(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets,
                                 gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status) const
{
    if (!out_hits || !out_hit_offsets) fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
    *out_hits = nullptr;
    uint64_t total = 0;
    int rc = locate_many(qbuf, qoff, nq, out_hit_offsets, nullptr, 0, &total, out_status);  // sizing pass
    if (rc != GDX_ERR_CAPACITY) {
        if (out_total) *out_total = total;
        return rc;
    }
    const size_t bytes = (total * sizeof(gdx_hit_t) + (2u << 20) - 1) / (2u << 20) * (2u << 20);
    void *mem = nullptr;
    if (posix_memalign(&mem, 2u << 20, bytes) != 0 || !mem) fail(GDX_ERR_DEVICE, "out of host memory for %llu hits", static_cast<unsigned long long>(total));
    (void)madvise(mem, bytes, MADV_HUGEPAGE);
    try {
        rc = locate_many(qbuf, qoff, nq, out_hit_offsets, static_cast<gdx_hit_t *>(mem), total, out_total, out_status);
    } catch (...) {
        std::free(mem);
        throw;
    }
    *out_hits = static_cast<gdx_hit_t *>(mem);
    return rc;
}

void WideIndex::export_bwt(uint8_t *bwt) const
{
    // decode from the rank lines (host side, for parity checks on small inputs)
    GDX_HIP(hipSetDevice(p_->cfg.device_id));
    const uint64_t n_lines = div_ceil(p_->n + 1, 128);
    std::vector<u32x4> lines(n_lines * 4);
    GDX_HIP(hipMemcpy(lines.data(), p_->lines.get(), lines.size() * sizeof(u32x4), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < p_->n; i++) {
        const u32x4 c = lines[(i >> 7) * 4 + ((i & 127u) >> 5)];
        const uint32_t t = static_cast<uint32_t>(i & 31u);
        bwt[i] = static_cast<uint8_t>(((c.x >> t) & 1u) | (((c.y >> t) & 1u) << 1) | (((c.z >> t) & 1u) << 2));
    }
}

}  // namespace gdx
