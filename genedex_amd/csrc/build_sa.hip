// build_sa.hip -- suffix array of the concatenated, densely encoded text, on the GPU.
//
// Stands in for the reference's third-party SACA (libsais 0.2.0 via construction/mod.rs:88-103):
// the result is THE suffix array of the byte string (sentinels are ordinary symbols 0, the end of
// the string compares smallest), which is unique, so it equals libsais' output.
//
// Method (plumbing, not the measured hot path): prefix doubling with discarding.
//   1. partition the suffixes by their first symbol (bucket c occupies SA[C[c] .. C[c+1]));
//   2. per bucket, radix-sort by the next k0 symbols packed into one 64-bit key (rocPRIM);
//      order h0 = 1 + k0 symbols is now fixed (k0 = 21 for DNA);
//   3. ISA[i] = first SA slot of i's group; suffixes in groups of size 1 are final;
//   4. rounds: the non-singleton suffixes are sorted by (group, ISA[i + h]) and regrouped, h doubles
//      (Larsson-Sadakane); on i.i.d. DNA only ~n^2/4^22 suffixes survive step 2.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <vector>

#include "build.hpp"
#include "common.hpp"

namespace gdx {

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kChunk = 4096;  // positions per block in the partition passes

unsigned grid_for_items(uint64_t items, uint64_t cap = 256u * 16u)
{
    const uint64_t blocks = (items + kBlock - 1) / kBlock;
    return static_cast<unsigned>(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

// ---- step 1: partition by first symbol --------------------------------------------------

// counts[c * n_chunks + chunk] = #positions in the chunk whose symbol is c
__global__ __launch_bounds__(kBlock) void chunk_histogram_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                                 int sigma, uint32_t n_chunks,
                                                                 uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_cnt[256];
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        for (int i = threadIdx.x; i < sigma; i += kBlock) s_cnt[i] = 0;
        __syncthreads();
        const uint64_t p0 = static_cast<uint64_t>(chunk) * kChunk;
        for (uint32_t t = threadIdx.x; t < kChunk; t += kBlock) {
            const uint64_t p = p0 + t;
            if (p < n) atomicAdd(&s_cnt[text[p]], 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < sigma; i += kBlock)
            counts[static_cast<uint64_t>(i) * n_chunks + chunk] = s_cnt[i];
        __syncthreads();
    }
}

// offsets = exclusive scan of counts (u64 because the running sum reaches n)
__global__ __launch_bounds__(kBlock) void chunk_scatter_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                               int sigma, uint32_t n_chunks,
                                                               const uint64_t *__restrict__ offsets,
                                                               uint32_t *__restrict__ idx_out)
{
    __shared__ uint32_t s_cnt[256];
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        for (int i = threadIdx.x; i < sigma; i += kBlock) s_cnt[i] = 0;
        __syncthreads();
        const uint64_t p0 = static_cast<uint64_t>(chunk) * kChunk;
        for (uint32_t t = threadIdx.x; t < kChunk; t += kBlock) {
            const uint64_t p = p0 + t;
            if (p < n) {
                const uint32_t c = text[p];
                const uint32_t r = atomicAdd(&s_cnt[c], 1u);  // order inside a bucket is irrelevant
                idx_out[offsets[static_cast<uint64_t>(c) * n_chunks + chunk] + r] = static_cast<uint32_t>(p);
            }
        }
        __syncthreads();
    }
}

// ---- step 2: keys of the next k0 symbols --------------------------------------------------

// key = symbols text[i+1 .. i+1+k0) as (symbol+1) in `bits` bits each, most significant first;
// positions >= n contribute 0, which sorts before every real symbol.
__global__ __launch_bounds__(kBlock) void make_keys_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                           const uint32_t *__restrict__ idx, uint64_t m, int k0,
                                                           int bits, uint64_t *__restrict__ keys)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; j < m; j += stride) {
        const uint64_t i = static_cast<uint64_t>(idx[j]) + 1;
        uint64_t key = 0;
        for (int t = 0; t < k0; t++) {
            const uint64_t p = i + t;
            const uint64_t s = p < n ? static_cast<uint64_t>(text[p]) + 1u : 0u;
            key = (key << bits) | s;
        }
        keys[j] = key;
    }
}

// ---- step 3: groups --------------------------------------------------------------------------

// marks[j] = slot+1 of j if j opens a new group (key differs from the predecessor), else 0
__global__ __launch_bounds__(kBlock) void mark_group_heads_kernel(const uint64_t *__restrict__ keys, uint64_t m,
                                                                  uint32_t slot0, uint32_t *__restrict__ marks)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; j < m; j += stride) {
        const bool head = j == 0 || keys[j] != keys[j - 1];
        marks[j] = head ? slot0 + static_cast<uint32_t>(j) + 1u : 0u;
    }
}

// after the max-scan group_of[j] = (first slot of j's group)+1.  Writes ISA and appends the slots of
// non-singleton groups to `pending`.
__global__ __launch_bounds__(kBlock) void commit_groups_kernel(const uint32_t *__restrict__ group_of, uint64_t m,
                                                               uint32_t slot0, const uint32_t *__restrict__ sa,
                                                               uint32_t *__restrict__ isa,
                                                               uint32_t *__restrict__ pending,
                                                               unsigned long long *__restrict__ n_pending)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; j < m; j += stride) {
        const uint32_t g = group_of[j] - 1u;
        const uint32_t slot = slot0 + static_cast<uint32_t>(j);
        isa[sa[slot]] = g;
        const bool is_head = (g == slot);
        const bool next_is_head = (j + 1 == m) || (group_of[j + 1] - 1u == slot + 1u);
        if (!(is_head && next_is_head)) {
            const unsigned long long at = atomicAdd(n_pending, 1ull);
            pending[at] = slot;
        }
    }
}

// ---- step 4: refinement rounds ------------------------------------------------------------------

__global__ __launch_bounds__(kBlock) void make_pairs_kernel(const uint32_t *__restrict__ pending, uint64_t m,
                                                            const uint32_t *__restrict__ sa,
                                                            const uint32_t *__restrict__ isa, uint64_t n, uint64_t h,
                                                            uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; t < m; t += stride) {
        const uint32_t s = sa[pending[t]];
        const uint64_t p = static_cast<uint64_t>(s) + h;
        const uint64_t second = p < n ? static_cast<uint64_t>(isa[p]) + 1u : 0u;
        keys[t] = (static_cast<uint64_t>(isa[s]) << 32) | second;
        vals[t] = s;
    }
}

// marks[t] = t+1 where the old group (high half of the key) changes, else 0
__global__ __launch_bounds__(kBlock) void mark_old_groups_kernel(const uint64_t *__restrict__ keys, uint64_t m,
                                                                 uint32_t *__restrict__ marks)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; t < m; t += stride) {
        const bool head = t == 0 || (keys[t] >> 32) != (keys[t - 1] >> 32);
        marks[t] = head ? static_cast<uint32_t>(t) + 1u : 0u;
    }
}

// first_of[t] = (index of the first list element of t's old group)+1.  slot = group + (t - first);
// marks2[t] = slot+1 where the full key changes (a new, finer group opens), else 0
__global__ __launch_bounds__(kBlock) void mark_new_groups_kernel(const uint64_t *__restrict__ keys, uint64_t m,
                                                                 const uint32_t *__restrict__ first_of,
                                                                 uint32_t *__restrict__ slots,
                                                                 uint32_t *__restrict__ marks2)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; t < m; t += stride) {
        const uint32_t group = static_cast<uint32_t>(keys[t] >> 32);
        const uint32_t slot = group + (static_cast<uint32_t>(t) - (first_of[t] - 1u));
        slots[t] = slot;
        const bool head = t == 0 || keys[t] != keys[t - 1];
        marks2[t] = head ? slot + 1u : 0u;
    }
}

__global__ __launch_bounds__(kBlock) void commit_round_kernel(const uint32_t *__restrict__ slots,
                                                              const uint32_t *__restrict__ new_group_of,
                                                              const uint32_t *__restrict__ vals, uint64_t m,
                                                              uint32_t *__restrict__ sa, uint32_t *__restrict__ isa,
                                                              uint32_t *__restrict__ pending_next,
                                                              unsigned long long *__restrict__ n_pending_next)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; t < m; t += stride) {
        const uint32_t slot = slots[t];
        const uint32_t g = new_group_of[t] - 1u;
        const uint32_t s = vals[t];
        sa[slot] = s;
        isa[s] = g;
        const bool is_head = (g == slot);
        // the successor in the list is the next slot of the same old group iff slots are consecutive
        const bool next_is_head = (t + 1 == m) || (new_group_of[t + 1] - 1u != g);
        if (!(is_head && next_is_head)) {
            const unsigned long long at = atomicAdd(n_pending_next, 1ull);
            pending_next[at] = slot;
        }
    }
}

size_t max_scan_bytes(uint64_t m)
{
    size_t bytes = 0;
    uint32_t *p = nullptr;
    (void)rocprim::inclusive_scan(nullptr, bytes, p, p, static_cast<size_t>(m), rocprim::maximum<uint32_t>());
    return bytes;
}

size_t sort_bytes(uint64_t m)
{
    size_t bytes = 0;
    uint64_t *k = nullptr;
    uint32_t *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, static_cast<size_t>(m), 0u, 64u);
    return bytes;
}

int bits_for(uint64_t values)  // bits needed to hold 0..values-1
{
    int b = 1;
    while ((1ull << b) < values) b++;
    return b;
}

}  // namespace

void build_suffix_array(const uint8_t *d_text, uint64_t n, int sigma, const std::vector<uint64_t> &freq,
                        uint32_t *d_sa, hipStream_t stream, BuildStats *stats)
{
    if (n == 0) return;
    const int sym_bits = bits_for(static_cast<uint64_t>(sigma) + 1);  // symbol+1, 0 = beyond the end
    const int k0 = 64 / sym_bits > 32 ? 32 : 64 / sym_bits;
    const uint64_t h0 = 1 + static_cast<uint64_t>(k0);

    // ---- 1. partition -----------------------------------------------------------------------
    const uint64_t n_chunks64 = div_ceil(n, kChunk);
    const uint32_t n_chunks = static_cast<uint32_t>(n_chunks64);
    const uint64_t n_cells = static_cast<uint64_t>(sigma) * n_chunks;
    DeviceBuffer<uint32_t> chunk_counts(n_cells);
    DeviceBuffer<uint64_t> chunk_offsets(n_cells);
    DeviceBuffer<uint32_t> idx_in(n);
    hipLaunchKernelGGL(chunk_histogram_kernel, dim3(n_chunks < 65536u ? n_chunks : 65536u), dim3(kBlock), 0, stream,
                       d_text, n, sigma, n_chunks, chunk_counts.get());
    {
        size_t bytes = 0;
        rocprim::transform_iterator<uint32_t *, rocprim::identity<uint64_t>, uint64_t> in(chunk_counts.get(),
                                                                                          rocprim::identity<uint64_t>());
        GDX_HIP(rocprim::exclusive_scan(nullptr, bytes, in, chunk_offsets.get(), uint64_t(0),
                                        static_cast<size_t>(n_cells), rocprim::plus<uint64_t>(), stream));
        DeviceBuffer<uint8_t> temp(bytes);
        GDX_HIP(rocprim::exclusive_scan(temp.get(), bytes, in, chunk_offsets.get(), uint64_t(0),
                                        static_cast<size_t>(n_cells), rocprim::plus<uint64_t>(), stream));
        GDX_HIP(hipStreamSynchronize(stream));
    }
    hipLaunchKernelGGL(chunk_scatter_kernel, dim3(n_chunks < 65536u ? n_chunks : 65536u), dim3(kBlock), 0, stream,
                       d_text, n, sigma, n_chunks, chunk_offsets.get(), idx_in.get());
    GDX_HIP(hipStreamSynchronize(stream));
    chunk_counts.release();
    chunk_offsets.release();

    // ---- 2./3. per-bucket key sort, groups, ISA ---------------------------------------------------
    uint64_t max_bucket = 0;
    for (int c = 0; c < sigma; c++) max_bucket = freq[c] > max_bucket ? freq[c] : max_bucket;
    if (max_bucket >= (1ull << 31))
        fail(GDX_ERR_UNSUPPORTED, "suffix sorter: a first-symbol bucket holds %llu suffixes (limit 2^31-1)",
             static_cast<unsigned long long>(max_bucket));

    DeviceBuffer<uint32_t> isa(n);
    DeviceBuffer<uint32_t> pending(n);  // worst case: every suffix stays unresolved (repetitive text)
    DeviceBuffer<unsigned long long> n_pending(1);
    GDX_HIP(hipMemsetAsync(n_pending.get(), 0, sizeof(unsigned long long), stream));
    {
        DeviceBuffer<uint64_t> keys_in(max_bucket), keys_out(max_bucket);
        DeviceBuffer<uint32_t> marks(max_bucket);
        const size_t sort_tmp = sort_bytes(max_bucket), scan_tmp = max_scan_bytes(max_bucket);
        DeviceBuffer<uint8_t> temp(sort_tmp > scan_tmp ? sort_tmp : scan_tmp);
        uint64_t slot0 = 0;
        for (int c = 0; c < sigma; c++) {
            const uint64_t m = freq[c];
            if (m == 0) continue;
            const unsigned grid = grid_for_items(m);
            hipLaunchKernelGGL(make_keys_kernel, dim3(grid), dim3(kBlock), 0, stream, d_text, n,
                               idx_in.get() + slot0, m, k0, sym_bits, keys_in.get());
            size_t bytes = temp.bytes();
            GDX_HIP(rocprim::radix_sort_pairs(temp.get(), bytes, keys_in.get(), keys_out.get(),
                                              idx_in.get() + slot0, d_sa + slot0, static_cast<size_t>(m), 0u,
                                              static_cast<unsigned>(k0 * sym_bits), stream));
            hipLaunchKernelGGL(mark_group_heads_kernel, dim3(grid), dim3(kBlock), 0, stream, keys_out.get(), m,
                               static_cast<uint32_t>(slot0), marks.get());
            bytes = temp.bytes();
            GDX_HIP(rocprim::inclusive_scan(temp.get(), bytes, marks.get(), marks.get(), static_cast<size_t>(m),
                                            rocprim::maximum<uint32_t>(), stream));
            hipLaunchKernelGGL(commit_groups_kernel, dim3(grid), dim3(kBlock), 0, stream, marks.get(), m,
                               static_cast<uint32_t>(slot0), d_sa, isa.get(), pending.get(), n_pending.get());
            slot0 += m;
        }
        GDX_HIP(hipStreamSynchronize(stream));
    }
    idx_in.release();

    // ---- 4. refinement rounds ------------------------------------------------------------------------
    unsigned long long m_pending = 0;
    GDX_HIP(hipMemcpy(&m_pending, n_pending.get(), sizeof(m_pending), hipMemcpyDeviceToHost));
    if (stats) {
        stats->sa_initial_order = h0;
        stats->sa_pending_after_sort = m_pending;
        stats->sa_rounds = 0;
    }
    if (m_pending >= (1ull << 31))
        fail(GDX_ERR_UNSUPPORTED, "suffix sorter: %llu unresolved suffixes after the key sort (limit 2^31-1)",
             m_pending);
    DeviceBuffer<uint32_t> pending_next;
    uint64_t h = h0;
    while (m_pending > 0) {
        const uint64_t m = m_pending;
        const unsigned grid = grid_for_items(m);
        DeviceBuffer<uint64_t> keys_in(m), keys_out(m);
        DeviceBuffer<uint32_t> vals_in(m), vals_out(m), first_of(m), slots(m), marks2(m);
        if (pending_next.count < m) pending_next.alloc(m);
        const size_t sort_tmp = sort_bytes(m), scan_tmp = max_scan_bytes(m);
        DeviceBuffer<uint8_t> temp(sort_tmp > scan_tmp ? sort_tmp : scan_tmp);

        hipLaunchKernelGGL(make_pairs_kernel, dim3(grid), dim3(kBlock), 0, stream, pending.get(), m, d_sa,
                           isa.get(), n, h, keys_in.get(), vals_in.get());
        size_t bytes = temp.bytes();
        GDX_HIP(rocprim::radix_sort_pairs(temp.get(), bytes, keys_in.get(), keys_out.get(), vals_in.get(),
                                          vals_out.get(), static_cast<size_t>(m), 0u, 64u, stream));
        hipLaunchKernelGGL(mark_old_groups_kernel, dim3(grid), dim3(kBlock), 0, stream, keys_out.get(), m,
                           first_of.get());
        bytes = temp.bytes();
        GDX_HIP(rocprim::inclusive_scan(temp.get(), bytes, first_of.get(), first_of.get(), static_cast<size_t>(m),
                                        rocprim::maximum<uint32_t>(), stream));
        hipLaunchKernelGGL(mark_new_groups_kernel, dim3(grid), dim3(kBlock), 0, stream, keys_out.get(), m,
                           first_of.get(), slots.get(), marks2.get());
        bytes = temp.bytes();
        GDX_HIP(rocprim::inclusive_scan(temp.get(), bytes, marks2.get(), marks2.get(), static_cast<size_t>(m),
                                        rocprim::maximum<uint32_t>(), stream));
        GDX_HIP(hipMemsetAsync(n_pending.get(), 0, sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(commit_round_kernel, dim3(grid), dim3(kBlock), 0, stream, slots.get(), marks2.get(),
                           vals_out.get(), m, d_sa, isa.get(), pending_next.get(), n_pending.get());
        GDX_HIP(hipStreamSynchronize(stream));
        GDX_HIP(hipMemcpy(&m_pending, n_pending.get(), sizeof(m_pending), hipMemcpyDeviceToHost));
        std::swap(pending.ptr, pending_next.ptr);
        std::swap(pending.count, pending_next.count);
        h *= 2;
        if (stats) stats->sa_rounds++;
    }
}

}  // namespace gdx
