// synth.hip -- synthetic DNA texts / query sets generated in HBM, and the two bandwidth
// micro-benchmarks the roofline fractions are quoted against.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"
#include "layout.hpp"
#include <cstdlib>

#include "synth.hpp"

namespace gdx {

namespace {

constexpr int kBlock = 256;
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;

__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// output number i (0-based) of the splitmix64 stream seeded with `seed`
__host__ __device__ inline uint64_t splitmix_at(uint64_t seed, uint64_t i) { return mix64(seed + (i + 1) * kGolden); }

__host__ __device__ inline uint64_t hash3(uint64_t seed, uint64_t a, uint64_t b)
{
    return mix64(splitmix_at(seed, a) ^ (b * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull));
}

unsigned grid_for_items(uint64_t items, uint64_t cap = 256u * 16u)
{
    const uint64_t blocks = (items + kBlock - 1) / kBlock;
    return static_cast<unsigned>(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

__global__ __launch_bounds__(kBlock) void synth_text_kernel(uint8_t *__restrict__ out, uint64_t n, uint64_t seed,
                                                            uint64_t n_threshold)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t r = splitmix_at(seed, i);
        const char acgt[4] = {'A', 'C', 'G', 'T'};
        out[i] = (r >> 32) < n_threshold ? 'N' : acgt[(r >> 8) & 3u];
    }
}

struct QueryLength {
    uint64_t seed, nq;
    uint32_t len_min, span;
    __host__ __device__ uint64_t operator()(uint64_t q) const
    {
        return q < nq ? len_min + (span ? hash3(seed, q, 0) % (span + 1ull) : 0ull) : 0ull;
    }
};

__global__ __launch_bounds__(kBlock) void synth_queries_kernel(const uint8_t *__restrict__ io_text,
                                                               const uint64_t *__restrict__ text_offsets,
                                                               uint64_t n_texts, uint64_t nq,
                                                               uint32_t sampled_per_million, uint64_t seed,
                                                               const uint64_t *__restrict__ qoff,
                                                               uint8_t *__restrict__ qbuf)
{
    const uint64_t io_len = text_offsets[n_texts] - text_offsets[0];
    const uint64_t base = text_offsets[0];
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; q < nq; q += stride) {
        const uint64_t b = qoff[q], len = qoff[q + 1] - b;
        bool done = false;
        if (len > 0 && len <= io_len && hash3(seed, q, 1) % 1000000ull < sampled_per_million) {
            for (uint32_t a = 0; a < 8 && !done; a++) {
                const uint64_t pos = base + hash3(seed, q, 2 + a) % (io_len - len + 1);
                // the window must lie inside one text: first offset > pos
                uint64_t lo = 0, hi = n_texts + 1;
                while (lo < hi) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if (text_offsets[mid] <= pos) lo = mid + 1;
                    else hi = mid;
                }
                if (lo > n_texts || pos + len > text_offsets[lo]) continue;
                bool clean = true;
                for (uint64_t j = 0; j < len; j++) {
                    const uint8_t s = io_text[pos + j];
                    clean &= (s != 'N' && s != 'n');
                }
                if (!clean) continue;
                for (uint64_t j = 0; j < len; j++) qbuf[b + j] = io_text[pos + j];
                done = true;
            }
        }
        if (!done) {
            const char acgt[4] = {'A', 'C', 'G', 'T'};
            uint64_t r = 0;
            for (uint64_t j = 0; j < len; j++) {
                if ((j & 31u) == 0) r = hash3(seed, q, 16 + (j >> 5));
                qbuf[b + j] = acgt[(r >> (2 * (j & 31u))) & 3u];
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void stream_copy_kernel(u32x4 *__restrict__ dst, const u32x4 *__restrict__ src,
                                                             uint64_t n16)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

__device__ __forceinline__ uint32_t pcg_next(uint32_t &state)
{
    state = state * 747796405u + 2891336453u;
    const uint32_t w = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (w >> 22u) ^ w;
}

__device__ __forceinline__ uint64_t pick_line(uint32_t r, uint64_t n_lines)
{
    return (static_cast<uint64_t>(r) * n_lines) >> 32;  // n_lines < 2^32
}

// mode 0: every lane reads whole random lines (4 or 8 x 16 B), two independent lines in flight;
// mode 2: the same, but the next line depends on the data just loaded (one dependent chain per lane)
template <int kVecPerLine, bool kDependent>
__global__ __launch_bounds__(kBlock) void gather_lane_kernel(const u32x4 *__restrict__ src, uint64_t n_lines,
                                                             uint64_t per_thread, uint64_t seed,
                                                             uint32_t *__restrict__ sink)
{
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    uint32_t state = static_cast<uint32_t>(mix64(seed + tid));
    uint32_t acc = 0;
    for (uint64_t k = 0; k < per_thread; k += 2) {
        const uint64_t l0 = pick_line(pcg_next(state), n_lines), l1 = pick_line(pcg_next(state), n_lines);
        u32x4 v0[kVecPerLine], v1[kVecPerLine];
#pragma unroll
        for (int j = 0; j < kVecPerLine; j++) {
            v0[j] = src[l0 * kVecPerLine + j];
            v1[j] = src[l1 * kVecPerLine + j];
        }
#pragma unroll
        for (int j = 0; j < kVecPerLine; j++) acc ^= v0[j].x ^ v0[j].w ^ v1[j].y ^ v1[j].z;
        if (kDependent) state ^= acc;
    }
    if (acc == 0x12345u) *sink = acc;  // keeps the loads alive
}

// mode 3: every lane reads one random entry of kBytes (8, 16 or 32: a top-table entry, a jump entry) with one or two
// loads, two independent entries in flight: the access shape of a one-lane-per-query search
template <int kBytes>
__global__ __launch_bounds__(kBlock) void gather_small_kernel(const uint2 *__restrict__ src, uint64_t n_entries,
                                                              uint64_t per_thread, uint64_t seed,
                                                              uint32_t *__restrict__ sink)
{
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    uint32_t state = static_cast<uint32_t>(mix64(seed + tid));
    uint32_t acc = 0;
    for (uint64_t k = 0; k < per_thread; k += 2) {
        const uint64_t e0 = pick_line(pcg_next(state), n_entries), e1 = pick_line(pcg_next(state), n_entries);
        if (kBytes == 8) {
            const uint2 a = src[e0], b = src[e1];
            acc ^= a.x ^ a.y ^ b.x ^ b.y;
        } else {
            const u32x4 *s16 = reinterpret_cast<const u32x4 *>(src);
            constexpr int kVec = kBytes >= 16 ? kBytes / 16 : 1;
            u32x4 v0[kVec], v1[kVec];
#pragma unroll
            for (int j = 0; j < kVec; j++) {
                v0[j] = s16[e0 * kVec + j];
                v1[j] = s16[e1 * kVec + j];
            }
#pragma unroll
            for (int j = 0; j < kVec; j++) acc ^= v0[j].x ^ v0[j].w ^ v1[j].y ^ v1[j].z;
        }
    }
    if (acc == 0x12345u) *sink = acc;
}

// mode 1: kVecPerLine adjacent lanes read one line, 16 bytes each, two independent lines in flight
template <int kVecPerLine>
__global__ __launch_bounds__(kBlock) void gather_group_kernel(const u32x4 *__restrict__ src, uint64_t n_lines,
                                                              uint64_t per_group, uint64_t seed,
                                                              uint32_t *__restrict__ sink)
{
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const uint64_t group = tid / kVecPerLine, sub = tid % kVecPerLine;
    uint32_t state = static_cast<uint32_t>(mix64(seed + group));
    uint32_t acc = 0;
    for (uint64_t k = 0; k < per_group; k += 2) {
        const uint64_t l0 = pick_line(pcg_next(state), n_lines), l1 = pick_line(pcg_next(state), n_lines);
        const u32x4 v0 = src[l0 * kVecPerLine + sub];
        const u32x4 v1 = src[l1 * kVecPerLine + sub];
        acc ^= v0.x ^ v0.w ^ v1.y ^ v1.z;
    }
    if (acc == 0x12345u) *sink = acc;
}

// streaming read: sums 16-byte words, kUnroll loads in flight per lane, optionally non-temporal (no cache allocation)
template <int kUnroll, bool kNt>
__global__ __launch_bounds__(kBlock) void stream_read_kernel(const u32x4 *__restrict__ src, uint64_t n16,
                                                             uint32_t *__restrict__ sink)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    uint32_t acc = 0;
    uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    for (; i + (kUnroll - 1) * stride < n16; i += kUnroll * stride) {
        u32x4 v[kUnroll];
#pragma unroll
        for (int k = 0; k < kUnroll; k++) v[k] = kNt ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < kUnroll; k++) acc ^= v[k].x ^ v[k].w;
    }
    for (; i < n16; i += stride) acc ^= src[i].x;
    if (acc == 0x12345u) *sink = acc;
}

}  // namespace

void launch_synth_text(uint8_t *d_out, uint64_t n, uint64_t seed, uint32_t n_per_million, hipStream_t stream)
{
    if (n == 0) return;
    const uint64_t threshold = (static_cast<uint64_t>(n_per_million) << 32) / 1000000ull;
    hipLaunchKernelGGL(synth_text_kernel, dim3(grid_for_items(n)), dim3(kBlock), 0, stream, d_out, n, seed, threshold);
}

void synth_queries(const uint8_t *d_io_text, const uint64_t *d_text_offsets, uint64_t n_texts, uint64_t nq,
                   uint32_t len_min, uint32_t len_max, uint32_t sampled_per_million, uint64_t seed, uint64_t *d_qoff,
                   uint8_t *d_qbuf, uint64_t qbuf_capacity, uint64_t *out_total_bytes, hipStream_t stream)
{
    if (len_max < len_min) fail(GDX_ERR_INVALID_ARGUMENT, "len_max < len_min");
    using LenIt = rocprim::transform_iterator<rocprim::counting_iterator<uint64_t>, QueryLength, uint64_t>;
    LenIt in(rocprim::counting_iterator<uint64_t>(0), QueryLength{seed, nq, len_min, len_max - len_min});
    size_t bytes = 0;
    GDX_HIP(rocprim::exclusive_scan(nullptr, bytes, in, d_qoff, uint64_t(0), static_cast<size_t>(nq + 1),
                                    rocprim::plus<uint64_t>(), stream));
    DeviceBuffer<uint8_t> temp(bytes ? bytes : 1);
    GDX_HIP(rocprim::exclusive_scan(temp.get(), bytes, in, d_qoff, uint64_t(0), static_cast<size_t>(nq + 1),
                                    rocprim::plus<uint64_t>(), stream));
    uint64_t total = 0;
    GDX_HIP(hipMemcpyAsync(&total, d_qoff + nq, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    if (out_total_bytes) *out_total_bytes = total;
    if (total > qbuf_capacity)
        fail(GDX_ERR_CAPACITY, "query buffer too small: need %llu bytes", static_cast<unsigned long long>(total));
    if (nq == 0) return;
    hipLaunchKernelGGL(synth_queries_kernel, dim3(grid_for_items(nq)), dim3(kBlock), 0, stream, d_io_text,
                       d_text_offsets, n_texts, nq, sampled_per_million, seed, d_qoff, d_qbuf);
    GDX_HIP(hipStreamSynchronize(stream));
    GDX_HIP(hipGetLastError());
}

void launch_stream_copy(void *d_dst, const void *d_src, uint64_t bytes, hipStream_t stream)
{
    const uint64_t n16 = bytes / 16;
    if (n16 == 0) return;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(256u * 8u), dim3(kBlock), 0, stream, static_cast<u32x4 *>(d_dst),
                       static_cast<const u32x4 *>(d_src), n16);
}

void launch_stream_read(const void *d_src, uint64_t bytes, uint32_t *d_sink, hipStream_t stream)
{
    const uint64_t n16 = bytes / 16;
    if (n16 == 0) return;
    // GDX_STREAM_VARIANT (experiments): 0 = 4 loads in flight, 1 = 8 non-temporal, 2 = 8 plain on twice the blocks
    static const int variant = [] { const char *e = getenv("GDX_STREAM_VARIANT"); return e ? atoi(e) : 1; }();
    const u32x4 *src = static_cast<const u32x4 *>(d_src);
    if (variant == 0) hipLaunchKernelGGL((stream_read_kernel<4, false>), dim3(256u * 8u), dim3(kBlock), 0, stream, src, n16, d_sink);
    else if (variant == 2) hipLaunchKernelGGL((stream_read_kernel<8, false>), dim3(256u * 16u), dim3(kBlock), 0, stream, src, n16, d_sink);
    else hipLaunchKernelGGL((stream_read_kernel<8, true>), dim3(256u * 8u), dim3(kBlock), 0, stream, src, n16, d_sink);
}

void launch_random_gather(const void *d_src, uint64_t n_lines, uint32_t line_bytes, uint64_t n_accesses,
                          uint64_t seed, uint32_t mode, uint32_t *d_sink, hipStream_t stream)
{
    if (n_lines >= (1ull << 32)) fail(GDX_ERR_INVALID_ARGUMENT, "n_lines must be < 2^32");
    if (n_lines == 0 || n_accesses == 0) return;
    const unsigned grid = 256u * 8u;
    const uint64_t threads = static_cast<uint64_t>(grid) * kBlock;
    if (mode == 3) {
        uint64_t per_thread = div_ceil(n_accesses, threads);
        per_thread += per_thread & 1u;
        const uint2 *s8 = static_cast<const uint2 *>(d_src);
        if (line_bytes == 8)
            hipLaunchKernelGGL(gather_small_kernel<8>, dim3(grid), dim3(kBlock), 0, stream, s8, n_lines, per_thread, seed, d_sink);
        else if (line_bytes == 16)
            hipLaunchKernelGGL(gather_small_kernel<16>, dim3(grid), dim3(kBlock), 0, stream, s8, n_lines, per_thread, seed, d_sink);
        else if (line_bytes == 32)
            hipLaunchKernelGGL(gather_small_kernel<32>, dim3(grid), dim3(kBlock), 0, stream, s8, n_lines, per_thread, seed, d_sink);
        else
            fail(GDX_ERR_INVALID_ARGUMENT, "mode 3: line_bytes must be 8, 16 or 32");
        return;
    }
    if (line_bytes != 64 && line_bytes != 128) fail(GDX_ERR_INVALID_ARGUMENT, "line_bytes must be 64 or 128");
    const u32x4 *src = static_cast<const u32x4 *>(d_src);
    if (mode == 0 || mode == 2) {
        uint64_t per_thread = div_ceil(n_accesses, threads);
        per_thread += per_thread & 1u;
#define GDX_LANE(V, D) \
    hipLaunchKernelGGL((gather_lane_kernel<V, D>), dim3(grid), dim3(kBlock), 0, stream, src, n_lines, per_thread, seed, d_sink)
        if (line_bytes == 64) {
            if (mode == 0) GDX_LANE(4, false);
            else GDX_LANE(4, true);
        } else {
            if (mode == 0) GDX_LANE(8, false);
            else GDX_LANE(8, true);
        }
#undef GDX_LANE
    } else {
        const uint64_t groups = threads / (line_bytes / 16);
        uint64_t per_group = div_ceil(n_accesses, groups);
        per_group += per_group & 1u;
        if (line_bytes == 64)
            hipLaunchKernelGGL(gather_group_kernel<4>, dim3(grid), dim3(kBlock), 0, stream, src, n_lines, per_group, seed, d_sink);
        else
            hipLaunchKernelGGL(gather_group_kernel<8>, dim3(grid), dim3(kBlock), 0, stream, src, n_lines, per_group, seed, d_sink);
    }
}

}  // namespace gdx
