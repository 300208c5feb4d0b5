// host_api.hip -- the host-pointer query calls (gdx_count_many, gdx_cursors_for_many_queries, gdx_locate_many):
// FmIndex::count_many / cursors_for_many_queries / locate_many of the reference (lib.rs:155-185) for callers whose
// queries and results live in host memory.
//
// A call is a pipeline over chunks of the batch, four chunks in flight, three host threads:
//
//     feeder thread + workers: user memory -> pinned staging  |  H2D (copy-in stream)  |  kernels (compute stream)
//     calling thread: waits for a chunk's search (locate: its number of hits), enqueues locate and the copies out
//                                                             |  D2H (copy-out stream) |  drainer thread + workers:
//                                                                                         pinned -> the user's arrays
//
// so that the PCIe transfers of chunk k + 1 / k - 1 run beside the kernels of chunk k and a call costs about
// max(H2D time, D2H time, kernel time) instead of their sum.  Counts and interval borders travel narrow (u32) and are
// widened by the host threads while they copy; so do the hits of a locate since round 4 (8-byte gdx_hit32_t over PCIe,
// 16-byte gdx_hit_t in the caller's array; round 3 shipped them wide, when 58 bytes per read came in and the D2H link had
// room -- with reads as 2-bit codes, 12.5 bytes in, the results are the longer leg); the chunk's hit offsets stay u64.  The
// host cores of a container often do not.  Results are order preserving: chunk boundaries are invisible to the caller.
#include <malloc.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>

#include <sched.h>
#include <sys/mman.h>

#include "fm_index.hpp"
#include "kernels.hpp"
#include "pack_host.hpp"
#include "wire_host.hpp"

namespace gdx {

namespace {
double now_seconds()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

constexpr int kSlots = 4;
std::atomic<uint64_t> g_chunk_bytes{32ull << 20};   // query bytes per chunk
std::atomic<uint64_t> g_chunk_queries{1ull << 20};  // and at most this many queries

// ---- a small persistent worker pool for the staging copies ------------------------------------------------
class WorkerPool {
public:
    explicit WorkerPool(unsigned n) : n_(n < 1 ? 1 : n)
    {
        for (unsigned i = 1; i < n_; i++) threads_.emplace_back([this, i] { loop(i); });
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    unsigned size() const { return n_; }
    // fn(worker, n_workers) on every worker (the caller is worker 0); returns when all are done
    void run(const std::function<void(unsigned, unsigned)> &fn)
    {
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            pending_ = n_ - 1;
            gen_++;
        }
        cv_.notify_all();
        fn(0, n_);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }
    // [0, n) split into contiguous parts of whole `align` units
    template <class F>
    void parallel_range(uint64_t n, uint64_t align, F f)
    {
        if (n < (1u << 16) || n_ == 1) {
            f(0, n);
            return;
        }
        run([&](unsigned w, unsigned nw) {
            const uint64_t per = (n / nw + align) / align * align;
            const uint64_t lo = std::min<uint64_t>(n, per * w), hi = std::min<uint64_t>(n, lo + per);
            if (hi > lo) f(lo, hi);
        });
    }

private:
    void loop(unsigned idx)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned, unsigned)> *fn = nullptr;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
            }
            if (fn) (*fn)(idx, n_);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    unsigned n_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned, unsigned)> *fn_ = nullptr;
    unsigned pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// CPUs this process may use: the affinity mask cut by the cgroup quota (a container on a 256-thread host may own 16)
unsigned usable_cpus()
{
    static const unsigned n = [] {
        unsigned cpus = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = static_cast<unsigned>(CPU_COUNT(&set));
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
            char quota[32] = {0};
            unsigned long period = 0;
            if (std::fscanf(f, "%31s %lu", quota, &period) == 2 && period != 0 && std::strcmp(quota, "max") != 0)
                cpus = std::min<unsigned>(cpus, std::max(1ul, std::strtoul(quota, nullptr, 10) / period));
            std::fclose(f);
        } else if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1 (period 100 ms default)
            long q = -1;
            if (std::fscanf(g, "%ld", &q) == 1 && q > 0) cpus = std::min<unsigned>(cpus, std::max(1l, q / 100000));
            std::fclose(g);
        }
        return std::max(1u, cpus);
    }();
    return n;
}

// workers of ONE pool; a call runs two pools (staging in, draining out) beside its three pipeline threads
unsigned host_threads()
{
    static const unsigned n = [] {
        if (const char *e = getenv("GDX_HOST_THREADS")) return static_cast<unsigned>(std::max(1, atoi(e)));
        const unsigned cpus = usable_cpus();
        return std::min(16u, std::max(2u, cpus > 6u ? (cpus - 2u) / 2u : 2u));
    }();
    return n;
}

// ---- grow-only pinned host memory and device memory per (thread, device, slot) -----------------------------
struct Buf {
    void *ptr = nullptr;
    size_t bytes = 0;
};
struct BufCache {
    std::vector<std::pair<uint64_t, Buf>> pinned, device;  // key = device << 8 | id
    ~BufCache()
    {
        for (auto &kv : pinned)
            if (kv.second.ptr) (void)hipHostFree(kv.second.ptr);
        for (auto &kv : device)
            if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    }
};
BufCache &cache()
{
    thread_local BufCache c;
    return c;
}
void *cached(std::vector<std::pair<uint64_t, Buf>> &v, uint64_t key, size_t bytes, bool pinned)
{
    Buf *b = nullptr;
    for (auto &kv : v)
        if (kv.first == key) b = &kv.second;
    if (!b) {
        v.emplace_back(key, Buf{});
        b = &v.back().second;
    }
    if (b->bytes < bytes) {
        if (b->ptr) {
            GDX_HIP(hipDeviceSynchronize());
            if (pinned) GDX_HIP(hipHostFree(b->ptr));
            else GDX_HIP(hipFree(b->ptr));
            b->ptr = nullptr;
            b->bytes = 0;
        }
        const size_t want = bytes + bytes / 4 + 4096;
        if (pinned) GDX_HIP(hipHostMalloc(&b->ptr, want, hipHostMallocDefault));
        else GDX_HIP(hipMalloc(&b->ptr, want));
        b->bytes = want;
    }
    return b->ptr;
}
template <class T>
T *pinned_buf(int device, int id, size_t count)
{
    return static_cast<T *>(cached(cache().pinned, (static_cast<uint64_t>(device) << 8) | id, count * sizeof(T) + 16, true));
}
template <class T>
T *device_buf(int device, int id, size_t count)
{
    return static_cast<T *>(cached(cache().device, (static_cast<uint64_t>(device) << 8) | id, count * sizeof(T) + 16, false));
}

struct Streams {
    hipStream_t in = nullptr, k = nullptr, out = nullptr;
    hipEvent_t ev_in[kSlots] = {}, ev_k[kSlots] = {}, ev_out[kSlots] = {}, ev_total[kSlots] = {};
    Streams()
    {
        GDX_HIP(hipStreamCreateWithFlags(&in, hipStreamNonBlocking));
        GDX_HIP(hipStreamCreateWithFlags(&k, hipStreamNonBlocking));
        GDX_HIP(hipStreamCreateWithFlags(&out, hipStreamNonBlocking));
        for (int s = 0; s < kSlots; s++) {
            GDX_HIP(hipEventCreateWithFlags(&ev_in[s], hipEventDisableTiming));
            GDX_HIP(hipEventCreateWithFlags(&ev_k[s], hipEventDisableTiming));
            GDX_HIP(hipEventCreateWithFlags(&ev_out[s], hipEventDisableTiming));
            GDX_HIP(hipEventCreateWithFlags(&ev_total[s], hipEventDisableTiming));
        }
    }
    ~Streams()
    {
        for (int s = 0; s < kSlots; s++) {
            if (ev_in[s]) (void)hipEventDestroy(ev_in[s]);
            if (ev_k[s]) (void)hipEventDestroy(ev_k[s]);
            if (ev_out[s]) (void)hipEventDestroy(ev_out[s]);
            if (ev_total[s]) (void)hipEventDestroy(ev_total[s]);
        }
        if (in) (void)hipStreamDestroy(in);
        if (k) (void)hipStreamDestroy(k);
        if (out) (void)hipStreamDestroy(out);
    }
};

__global__ __launch_bounds__(256) void unpack_status_kernel(const uint4 *__restrict__ rec, uint64_t nq,
                                                            uint8_t *__restrict__ status)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; q < nq; q += stride)
        status[q] = static_cast<uint8_t>(rec[q].w >> 24);
}

// *flag |= 1 when a status byte is not 0 (then, and only then, a chunk's status bytes cross the link)
__global__ __launch_bounds__(256) void any_status_kernel(const uint8_t *__restrict__ status, uint64_t nq, unsigned long long *__restrict__ flag)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u, quads = nq / 16u;
    uint32_t any = 0;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; i < quads; i += stride) {
        const uint4 v = reinterpret_cast<const uint4 *>(status)[i];
        any |= v.x | v.y | v.z | v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (nq & 15u)) any |= status[quads * 16u + threadIdx.x];
    if (any != 0u) atomicOr(flag, 1ull);
}

void check_queries(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, WorkerPool &pool, uint64_t uniform_len)
{
    if (nq >= 0xffffffffull) fail(GDX_ERR_UNSUPPORTED, "more than 2^32-2 queries in one call");
    if (uniform_len != 0) {  // a uniform batch: query i = symbols [i * uniform_len, (i + 1) * uniform_len), no offsets
        if (uniform_len >= (1ull << 21)) fail(GDX_ERR_INVALID_ARGUMENT, "uniform_len must be below 2^21");
        if (nq != 0 && !qbuf) fail(GDX_ERR_INVALID_ARGUMENT, "qbuf is null");
        return;
    }
    if (!qoff) fail(GDX_ERR_INVALID_ARGUMENT, "qoff is null");
    std::atomic<bool> bad{false};
    pool.parallel_range(nq, 1, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++)
            if (qoff[i + 1] < qoff[i]) bad.store(true);
    });
    if (bad.load()) fail(GDX_ERR_INVALID_ARGUMENT, "qoff must be non-decreasing");
    if (qoff[nq] > qoff[0] && !qbuf) fail(GDX_ERR_INVALID_ARGUMENT, "qbuf is null");
}

enum class Kind { kIntervals, kCounts, kLocate, kLocate32 };

// every slot's u32 offsets (entries 1 .. n of a chunk) shifted by the hits of the chunks before it
__global__ __launch_bounds__(256) void add_hit_base_kernel(uint32_t *__restrict__ off, uint64_t n, uint32_t base)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; i < n; i += stride) off[i] += base;
}

// true: `p` is host memory the device can copy from directly (hipHostMalloc / hipHostRegister): no staging copy
bool is_pinned_host(const void *p)
{
    if (p == nullptr) return false;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();  // (ordinary pageable memory: not an error of the call)
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

struct Chunk {
    uint64_t q0 = 0, nq = 0, bytes = 0, total = 0, hit_base = 0;
};

// A chunk of the wide locate calls whose hits do not fit the fused step's 32-bit offsets: the call is run again with the
// device-written results of rounds 1-4 (u64 offsets throughout), which have no such limit (host_pipeline, locate_many*)
struct RetryWithWideOffsets {};
thread_local bool t_force_plain_locate = false;
// (tests: GDX_TEST_CHUNK_HIT_LIMIT lowers the number of hits a chunk of the fused step may hold; read per call)
uint64_t chunk_hit_limit()
{
    const char *e = getenv("GDX_TEST_CHUNK_HIT_LIMIT");
    const uint64_t v = e ? std::strtoull(e, nullptr, 10) : 0;
    return v != 0 ? v : 0xffffffffull;
}

}  // namespace

// The pipeline behind all three calls.  Intervals: out_a = start, out_b = end.  Counts: out_a = counts.  Locate:
// out_a = hit offsets (nq + 1), hits / hits_capacity / out_total as in gdx_locate_many; grow_hits (optional) is asked
// for room when the caller's buffer is managed by the library (gdx_locate_many_alloc).
int FmIndex::host_pipeline(int kind_i, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_a,
                           uint64_t *out_b, uint8_t *out_status, gdx_hit_t *hits, uint64_t hits_capacity,
                           uint64_t *out_total, const std::function<gdx_hit_t *(uint64_t, uint64_t *)> *grow_hits,
                           bool packed, uint64_t uniform_len, Narrow32Sink *narrow) const
{
    // packed: qbuf holds 2-bit codes (symbol j in bits 2 (j & 3) of byte j >> 2) and qoff counts symbols
    // uniform_len != 0: query i = symbols [i * uniform_len, (i + 1) * uniform_len) of the buffer, qoff is not looked at: no
    // offsets are staged, copied or read by the kernels (gdx_query_layout_t)
    const Kind kind = static_cast<Kind>(kind_i);
    // Locate calls: by default a chunk's results do not cross the link as offsets and hits but as the "found bitmap" wire the
    // multi-GPU gather uses (launch_wire_pack: a bit per read, position + text id byte per found read, the exceptions with their
    // hits -- 3.7-4.6 bytes per read instead of 12.2 narrow / 16.2 wide), and the drainer's workers expand them into the
    // caller's arrays (wire_host.hpp): the link's two directions share its rate (65 GB/s together on the test box against 57
    // alone), so bytes saved going out are time saved.  GDX_HOST_RESULTS=dma: the device-written forms (hosts with few cores
    // to spare).  Collections of more than 256 texts take those too (text ids travel as bytes).
    static const int narrow_mode = [] {
        const char *e = getenv("GDX_HOST_RESULTS");
        if (e && std::strcmp(e, "dma") == 0) return 0;
        if (e && std::strcmp(e, "wire") == 0) return 1;
        return host_threads() >= 4 ? 1 : 0;
    }();
    // A sizing call of gdx_locate_many (no hit buffer): search, totals and offsets only -- nothing is located, packed or copied out
    // that the call would throw away (round-5 advisor); it takes the plain path, whose locate launch is then skipped
    const bool sizing_only = kind == Kind::kLocate && grow_hits == nullptr && (hits == nullptr || hits_capacity == 0);
    const bool wire = (kind == Kind::kLocate32 || kind == Kind::kLocate) && narrow_mode == 1 && n_texts_ <= 256 && !sizing_only &&
                      !(kind == Kind::kLocate && t_force_plain_locate);
    // stepped: every chunk is ONE fused step (launch_locate_step: search, totals, u32 offsets, 8-byte hits, no host round trip)
    const bool stepped = kind == Kind::kLocate32 || (kind == Kind::kLocate && wire);
    const bool plain_locate = kind == Kind::kLocate && !stepped;  // search | total to the host | locate, u64 offsets (rounds 1-4)
    const bool pinned_input = is_pinned_host(qbuf);
    // (more than host_threads() workers did not shorten the expansion -- 13 ms of a 100 M-read call with 7 workers, 11 with 13 --
    // and starved the runtime's own threads on a 16-CPU container)
    WorkerPool pool(host_threads());
    check_queries(qbuf, qoff, nq, pool, uniform_len);
    // ASCII batches of the count / locate calls are packed ON THE HOST, chunk by chunk in the feeder's worker pool (pack_host.hpp:
    // 32 symbols per step when the alphabet's table has a nucleotide alphabet's shape), and cross the link as 2-bit codes -- 12.5
    // bytes per len-50 read (+ 8 of offsets) instead of 58: the link, not the kernels, bounds these calls.  A chunk that holds a
    // symbol 2 bits cannot name (N, an invalid byte) goes as it is, ASCII, and gives the reference's result for it.  Exact
    // intervals stay ASCII (their kernels read bytes).  GDX_HOST_PACK=0: never.
    const bool env_host_pack = [] { const char *e = getenv("GDX_HOST_PACK"); return !(e && atoi(e) == 0); }();  // (read per call: tests)
    const PackPlan pack_plan = make_pack_plan(cfg_.io_to_dense);
    const bool host_pack = !packed && env_host_pack && kind != Kind::kIntervals && view_.layout == 0 && view_.n_searchable >= 4 &&
                           pack_plan.fast && pack_have_avx2();
    const bool uniform = uniform_len != 0;
    auto off_of = [&](uint64_t i) { return uniform ? i * uniform_len : qoff[i]; };
    if (out_total) *out_total = 0;
    if (kind == Kind::kLocate && out_a) out_a[0] = 0;
    if (kind == Kind::kLocate32 && narrow != nullptr && narrow->offsets != nullptr) narrow->offsets[0] = 0;
    if (nq == 0) return GDX_OK;
    make_current();
    const int dev = cfg_.device_id;
    const QueryOptions qo = query_options();
    Streams st;

    // chunk boundaries: at most kChunkBytes of query bytes and kChunkQueries queries each
    // (a chunk's limit counts the bytes that cross the link: four symbols per byte of a packed batch; the narrow locate runs a
    // whole fused step per chunk, a dozen launches, and takes up to four times the queries)
    const uint64_t kChunkBytes = g_chunk_bytes.load() * ((packed || host_pack) ? 4 : 1), kChunkQueries = g_chunk_queries.load() * (stepped ? 4 : 1);
    std::vector<Chunk> chunks;
    // (uniform: every chunk but the last holds a multiple of 8 queries, so that a chunk starts on a 16-bit unit of a packed
    // buffer and on a byte of an ASCII one, and is a uniform batch of its own)
    const uint64_t uniform_chunk = uniform ? std::max<uint64_t>(8, std::min(kChunkQueries, kChunkBytes / uniform_len) / 8 * 8) : 0;
    for (uint64_t q0 = 0; q0 < nq;) {
        uint64_t hi = std::min(nq, q0 + (uniform ? uniform_chunk : kChunkQueries));
        if (!uniform && qoff[hi] - qoff[q0] > kChunkBytes) {
            const uint64_t *p = std::upper_bound(qoff + q0 + 1, qoff + hi + 1, qoff[q0] + kChunkBytes);
            hi = static_cast<uint64_t>(p - qoff) - 1;
            if (hi <= q0) hi = q0 + 1;  // a single query longer than a chunk
        }
        Chunk c;
        c.q0 = q0;
        c.nq = hi - q0;
        c.bytes = off_of(hi) - off_of(q0);
        chunks.push_back(c);
        q0 = hi;
    }
    const size_t n_chunks = chunks.size();
    uint64_t max_nq = 0, max_bytes = 0;
    for (const Chunk &c : chunks) {
        max_nq = std::max(max_nq, c.nq);
        max_bytes = std::max(max_bytes, c.bytes);
    }
    // (packed: max_bytes counts symbols, four per byte, and a chunk starts up to seven symbols before its first query)
    const uint64_t qbuf_cap = div_ceil((packed ? div_ceil(max_bytes + 8, 4) : max_bytes) + 1, 8) * 8 + 24;

    // per-slot buffers (ids: slot * 16 + n)
    uint8_t *h_in[kSlots], *d_qbuf[kSlots], *h_status[kSlots], *d_status[kSlots];
    uint64_t *h_qoff[kSlots], *d_qoff[kSlots], *d_off[kSlots];
    uint32_t *h_a[kSlots], *h_b[kSlots], *d_a[kSlots], *d_b[kSlots];
    uint4 *d_rec[kSlots];
    uint64_t *h_total[kSlots];
    void *d_scan[kSlots];
    size_t scan_bytes = 0;
    if (plain_locate) scan_bytes = hit_offsets_rec_temp_bytes(max_nq);
    for (int s = 0; s < kSlots; s++) {
        h_in[s] = pinned_buf<uint8_t>(dev, s * 16 + 0, qbuf_cap);
        h_qoff[s] = pinned_buf<uint64_t>(dev, s * 16 + 1, max_nq + 1);
        h_a[s] = pinned_buf<uint32_t>(dev, s * 16 + 2, max_nq);
        h_b[s] = kind == Kind::kIntervals ? pinned_buf<uint32_t>(dev, s * 16 + 3, max_nq) : nullptr;
        h_status[s] = pinned_buf<uint8_t>(dev, s * 16 + 4, max_nq);
        h_total[s] = pinned_buf<uint64_t>(dev, s * 16 + 5, 1);
        d_qbuf[s] = device_buf<uint8_t>(dev, s * 16 + 0, qbuf_cap);
        d_qoff[s] = device_buf<uint64_t>(dev, s * 16 + 1, max_nq + 1);
        d_a[s] = device_buf<uint32_t>(dev, s * 16 + 2, max_nq);
        d_b[s] = kind == Kind::kIntervals ? device_buf<uint32_t>(dev, s * 16 + 3, max_nq) : nullptr;
        d_status[s] = device_buf<uint8_t>(dev, s * 16 + 4, max_nq);
        d_rec[s] = plain_locate ? device_buf<uint4>(dev, s * 16 + 5, max_nq) : nullptr;
        d_off[s] = plain_locate ? device_buf<uint64_t>(dev, s * 16 + 6, max_nq + 1) : nullptr;
        d_scan[s] = plain_locate ? device_buf<uint8_t>(dev, s * 16 + 7, scan_bytes ? scan_bytes : 1) : nullptr;
    }
    // stepped: device buffers sized for the chunk's queries and a margin.  Without the wire (narrow call only) the u32 offsets
    // and 8-byte hits go by D2H copy straight into the caller-visible pinned arrays of the sink
    uint32_t *d_cmp[kSlots] = {}, *d_off32[kSlots] = {};
    unsigned long long *d_tot[kSlots] = {};
    uint64_t n32_cap = 0;  // hit slots a chunk's device buffers hold
    void *d_hits[kSlots] = {}, *h_hits[kSlots] = {};
    void *d_ws[kSlots] = {};
    uint8_t *d_wire[kSlots] = {}, *h_wire[kSlots] = {}, *d_exc[kSlots] = {}, *h_exc[kSlots] = {}, *d_wws[kSlots] = {};
    uint64_t exc_hits_cap[kSlots] = {}, w_found[kSlots] = {}, w_exc[kSlots] = {}, w_exc_hits[kSlots] = {};
    bool w_status[kSlots] = {};
    // a slot's wire: {meta 16 B | bitmap, 256 B per tile | tile_found | tile_off | found_pos} and -- collections of more than one
    // text -- the found reads' text ids behind it, the same offsets on both sides, so that ONE copy brings everything up to the
    // last found read's position and one more the ids
    const bool w_ids = n_texts_ > 1;
    const uint64_t w_tiles = div_ceil(max_nq, kHostWireTile);
    const uint64_t w_bitmap = 256, w_tile_found = w_bitmap + w_tiles * 256, w_tile_off = w_tile_found + div_ceil((w_tiles + 1) * 4, 256) * 256,
                   w_found_pos = w_tile_off + div_ceil((w_tiles + 1) * 4, 256) * 256, w_found_ids = w_found_pos + div_ceil(max_nq * 4 + 4, 256) * 256,
                   w_bytes = w_found_ids + (w_ids ? max_nq + 1 : 0);
    const bool flagged = kind == Kind::kCounts || kind == Kind::kIntervals;  // status bytes stay on the device unless one is set
    if (flagged)
        for (int s = 0; s < kSlots; s++) {
            d_tot[s] = device_buf<unsigned long long>(dev, s * 16 + 13, 4);
            h_total[s] = pinned_buf<uint64_t>(dev, s * 16 + 5, 8);
        }
    if (stepped) {
        if (kind == Kind::kLocate32 && narrow == nullptr) fail(GDX_ERR_INVALID_ARGUMENT, "internal: the narrow locate needs its sink");
        n32_cap = max_nq + max_nq / 4 + 4096;
        const size_t tws = scan_totals_workspace_bytes(max_nq);
        for (int s = 0; s < kSlots; s++) {
            d_rec[s] = device_buf<uint4>(dev, s * 16 + 5, max_nq);
            d_cmp[s] = device_buf<uint32_t>(dev, s * 16 + 11, max_nq);
            d_off32[s] = device_buf<uint32_t>(dev, s * 16 + 12, max_nq + 1);
            d_tot[s] = device_buf<unsigned long long>(dev, s * 16 + 13, 4);
            d_scan[s] = device_buf<uint8_t>(dev, s * 16 + 7, tws ? tws : 1);
            h_total[s] = pinned_buf<uint64_t>(dev, s * 16 + 5, 8);
            d_hits[s] = device_buf<uint8_t>(dev, s * 16 + 8, n32_cap * sizeof(gdx_hit32_t));
            d_ws[s] = device_buf<uint8_t>(dev, s * 16 + 9, locate_workspace_bytes(n32_cap));
            if (wire) {
                d_wire[s] = device_buf<uint8_t>(dev, s * 16 + 14, w_bytes);
                h_wire[s] = pinned_buf<uint8_t>(dev, s * 16 + 7, w_bytes);
                exc_hits_cap[s] = n32_cap;
                d_exc[s] = device_buf<uint8_t>(dev, s * 16 + 15, max_nq * 8 + exc_hits_cap[s] * sizeof(gdx_hit32_t));
                d_wws[s] = device_buf<uint8_t>(dev, s * 16 + 10, wire_pack_workspace_bytes(max_nq));
            }
        }
    }
    // (the exceptions of a slot: {read numbers | counts | hits}, max_nq entries of the first two)
    auto pack_wire = [&](int s, uint64_t chunk_nq) {
        uint8_t *w = d_wire[s], *e = d_exc[s];
        WireHostForm hf;
        hf.d_exc_hits32 = reinterpret_cast<gdx_hit32_t *>(e + max_nq * 8);
        hf.d_tile_off = reinterpret_cast<uint32_t *>(w + w_tile_off);
        hf.d_found_ids = w_ids ? w + w_found_ids : nullptr;
        hf.ix = &view_;
        hf.hits_stored = exc_hits_cap[s];
        launch_wire_pack(d_cmp[s], d_off32[s], true, static_cast<const gdx_hit32_t *>(d_hits[s]), chunk_nq, w + w_bitmap,
                         reinterpret_cast<uint32_t *>(w + w_tile_found), reinterpret_cast<uint32_t *>(w + w_found_pos), max_nq,
                         reinterpret_cast<uint32_t *>(e), reinterpret_cast<uint32_t *>(e + max_nq * 4), max_nq, nullptr, nullptr,
                         exc_hits_cap[s], reinterpret_cast<uint32_t *>(w), d_wws[s], st.k, &hf);
        GDX_HIP(hipGetLastError());
        GDX_HIP(hipMemcpyAsync(h_total[s] + 4, w, 16, hipMemcpyDeviceToHost, st.k));
    };
    uint64_t *h_off[kSlots] = {};  // locate: the chunk's hit offsets (h_off[i] = hits of its queries before query i)
    if (plain_locate)
        for (int s = 0; s < kSlots; s++) h_off[s] = pinned_buf<uint64_t>(dev, s * 16 + 10, max_nq + 1);
    // hits leave the device NARROW (gdx_hit32_t, 8 bytes) and are widened into the ABI's 16-byte gdx_hit_t by the drainer's
    // workers: with reads handed over as 2-bit codes the D2H link is the longer leg of a locate call (23 bytes per read out
    // against 12.5 in), and the drainer reads half as much from the staging buffer.  GDX_HOST_WIDE_HITS=1: 16-byte hits from
    // the device as in round 3 (the drainer then only copies)
    static const bool wide_hits = [] { const char *e = getenv("GDX_HOST_WIDE_HITS"); return e && atoi(e) != 0; }();
    const size_t hit_bytes = wide_hits ? sizeof(gdx_hit_t) : sizeof(gdx_hit32_t);

    std::atomic<bool> any_status{false};
    uint64_t hit_base = 0;    // hits of the chunks drained so far
    bool capacity_ok = true;  // locate: the caller's buffer holds everything so far

    // user memory -> pinned staging -> device, search (+ scan, total) enqueued; runs on the feeder thread
    auto stage_in_with = [&](size_t k, WorkerPool &pool) {
        const int s = static_cast<int>(k % kSlots);
        Chunk &c = chunks[k];
        bool as_packed = packed, packed_here = false;  // packed_here: the chunk's codes were made by this thread's workers, in h_in
        uint64_t chunk_uniform = uniform ? uniform_len : 0;  // != 0: every query of the chunk has this many symbols
        if (host_pack && c.bytes != 0) {
            // the chunk's symbols are packed from ITS first symbol on (symbol j of the chunk in bits 2 (j & 3) of byte j >> 2)
            const uint64_t first = off_of(c.q0), n_sym = off_of(c.q0 + c.nq) - first, nb = div_ceil(n_sym, 4);
            std::atomic<bool> other{false};  // a symbol that is not one of the dense codes 1..4
            std::atomic<bool> ragged{false};
            const uint64_t len0 = uniform ? uniform_len : qoff[c.q0 + 1] - qoff[c.q0];
            uint8_t *out = h_in[s];
            pool.run([&](unsigned w, unsigned nw) {
                const uint64_t per = (nb / nw + 64) / 64 * 64;
                const uint64_t lo = std::min(nb, per * w), hi = std::min(nb, lo + per);
                pack_range(pack_plan, cfg_.io_to_dense, qbuf + first, 0, n_sym, lo, hi, out,
                           [&](uint64_t) { other.store(true, std::memory_order_relaxed); });
                // reads of one length (a sequencer's) need no offsets at all: neither rebased, nor copied, nor read by the kernels
                if (!uniform && len0 != 0) {
                    const uint64_t qper = div_ceil(c.nq, nw), q_lo = std::min(c.nq, qper * w), q_hi = std::min(c.nq, q_lo + qper);
                    bool same = true;
                    for (uint64_t i = q_lo; i < q_hi; i++) same &= qoff[c.q0 + i + 1] - qoff[c.q0 + i] == len0;
                    if (!same) ragged.store(true, std::memory_order_relaxed);
                }
            });
            as_packed = packed_here = !other.load();
            if (packed_here && !uniform && len0 != 0 && len0 < (1ull << 21) && !ragged.load()) chunk_uniform = len0;
        }
        // packed (by the caller): the chunk starts at the 16-bit unit that holds its first symbol, offsets are rebased to that unit
        const uint64_t base = (as_packed && !packed_here) ? (off_of(c.q0) & ~7ull) : off_of(c.q0);
        const uint64_t src_byte = as_packed ? base / 4 : base;
        const uint64_t n_bytes = as_packed ? div_ceil(off_of(c.q0 + c.nq) - base, 4) : c.bytes;
        if (!pinned_input && !packed_here)
            pool.parallel_range(n_bytes, 64, [&](uint64_t lo, uint64_t hi) { stream_copy(h_in[s] + lo, qbuf + src_byte + lo, hi - lo); });
        if (chunk_uniform == 0)
            pool.parallel_range(c.nq + 1, 8, [&](uint64_t lo, uint64_t hi) {
                for (uint64_t i = lo; i < hi; i++) h_qoff[s][i] = qoff[c.q0 + i] - base;
            });
        const uint64_t padded = div_ceil(n_bytes + 2, 8) * 8;
        if (pinned_input && !packed_here) {  // the caller's buffer is pinned: the device reads it as it is, no staging copy
            if (n_bytes) GDX_HIP(hipMemcpyAsync(d_qbuf[s], qbuf + src_byte, n_bytes, hipMemcpyHostToDevice, st.in));
            GDX_HIP(hipMemsetAsync(d_qbuf[s] + n_bytes, 0, padded - n_bytes, st.in));  // windows may read past the last query
        } else {
            std::memset(h_in[s] + n_bytes, 0, padded - n_bytes);
            GDX_HIP(hipMemcpyAsync(d_qbuf[s], h_in[s], padded, hipMemcpyHostToDevice, st.in));
        }
        if (chunk_uniform == 0) GDX_HIP(hipMemcpyAsync(d_qoff[s], h_qoff[s], (c.nq + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st.in));
        GDX_HIP(hipEventRecord(st.ev_in[s], st.in));
        GDX_HIP(hipStreamWaitEvent(st.k, st.ev_in[s], 0));
        SearchCall call;
        call.d_qbuf = d_qbuf[s];
        if (chunk_uniform != 0) {  // (base == the chunk's first symbol: the chunk is a uniform batch of its own)
            call.uniform_len = static_cast<uint32_t>(chunk_uniform);
        } else {
            call.d_qbeg = d_qoff[s];
            call.d_qend = d_qoff[s] + 1;
        }
        call.nq = c.nq;
        call.packed = as_packed;
        if (stepped) {
            LocateStep step;
            step.call = call;
            step.call.d_rec = d_rec[s];
            step.call.d_compact = d_cmp[s];
            step.max_hits = qo.max_hits_per_query;
            step.take = true;  // locate(q).take(k)
            step.d_scan_workspace = d_scan[s];
            step.d_totals = d_tot[s];
            step.d_hit_offsets = d_off32[s];
            step.narrow = true;
            step.d_hits = d_hits[s];  // (sized for n32_cap hit slots; a chunk with more takes the second half again, stage_mid)
            step.hits_capacity = n32_cap;
            step.d_workspace = d_ws[s];
            GDX_HIP(hipMemsetAsync(d_tot[s] + 2, 0, sizeof(unsigned long long), st.k));
            launch_locate_step(view_, step, st.k, qo);
            GDX_HIP(hipGetLastError());
            launch_unpack_records(d_rec[s], c.nq, nullptr, d_status[s], st.k, d_cmp[s], d_tot[s] + 2);
            GDX_HIP(hipMemcpyAsync(h_total[s], d_tot[s], 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, st.k));
            if (wire) pack_wire(s, c.nq);
            GDX_HIP(hipEventRecord(st.ev_total[s], st.k));
            return;
        }
        if (kind == Kind::kIntervals) {
            call.d_start = d_a[s];
            call.d_end = d_b[s];
            call.d_status = d_status[s];
            call.mode = 0;
        } else if (kind == Kind::kCounts) {
            call.d_count = d_a[s];
            call.d_status = d_status[s];
            call.mode = 1;
        } else {
            call.d_rec = d_rec[s];
            call.mode = 1;
        }
        launch_search_call(view_, call, st.k, qo);
        GDX_HIP(hipGetLastError());
        if (plain_locate) {
            const unsigned blocks = static_cast<unsigned>(std::min<uint64_t>((c.nq + 255) / 256, 4096));
            hipLaunchKernelGGL(unpack_status_kernel, dim3(blocks), dim3(256), 0, st.k, d_rec[s], c.nq, d_status[s]);
            // (max_hits_per_query: a query gets slots for its first k rows only -- locate(q).take(k))
            launch_hit_offsets_rec(d_rec[s], c.nq, d_off[s], d_scan[s], scan_bytes, st.k, qo.max_hits_per_query, true);
            GDX_HIP(hipMemcpyAsync(h_total[s], d_off[s] + c.nq, sizeof(uint64_t), hipMemcpyDeviceToHost, st.k));
            GDX_HIP(hipEventRecord(st.ev_total[s], st.k));
        } else {
            GDX_HIP(hipEventRecord(st.ev_k[s], st.k));
            GDX_HIP(hipMemsetAsync(d_tot[s], 0, sizeof(unsigned long long), st.k));
            hipLaunchKernelGGL(any_status_kernel, dim3(static_cast<unsigned>(std::min<uint64_t>(c.nq / 4096 + 1, 1024))), dim3(256), 0, st.k,
                               d_status[s], c.nq, d_tot[s]);
            GDX_HIP(hipMemcpyAsync(h_total[s], d_tot[s], sizeof(uint64_t), hipMemcpyDeviceToHost, st.k));
            GDX_HIP(hipEventRecord(st.ev_total[s], st.k));
        }
    };

    uint64_t mid_hit_base = 0;  // narrow locate: hits of the chunks whose copies out are enqueued
    auto stage_mid = [&](size_t k) {  // locate: the chunk's total is known -> locate, then all D2H; else just D2H
        const int s = static_cast<int>(k % kSlots);
        Chunk &c = chunks[k];
        if (stepped) {
            GDX_HIP(hipEventSynchronize(st.ev_total[s]));
            c.total = h_total[s][0];
            const uint64_t rest = h_total[s][1];
            if (kind == Kind::kLocate32 && mid_hit_base + c.total >= (1ull << 32))
                fail(GDX_ERR_CAPACITY, "more than 2^32 - 1 hits: 32-bit hit offsets do not hold them (gdx_locate_many_alloc_layout does)");
            if (c.total >= chunk_hit_limit() && kind == Kind::kLocate) throw RetryWithWideOffsets{};  // (the wide call has no such limit)
            if (c.total >= 0xffffffffull)  // (the step counts a chunk's hit slots in 32 bits)
                fail(GDX_ERR_CAPACITY, "%llu queries of the batch have %llu hits, more than a chunk's 32-bit offsets hold: cap them "
                     "(gdx_query_options_t.max_hits_per_query), or take the device-written results (environment GDX_HOST_RESULTS=dma)",
                     static_cast<unsigned long long>(c.nq), static_cast<unsigned long long>(c.total));
            if (c.total > n32_cap) {  // rare: more hits than the chunk's buffers were sized for -- the second half again, with room
                d_hits[s] = device_buf<uint8_t>(dev, s * 16 + 8, c.total * sizeof(gdx_hit32_t));
                d_ws[s] = device_buf<uint8_t>(dev, s * 16 + 9, locate_workspace_bytes(c.total));
                launch_offsets_hits(view_, d_rec[s], d_cmp[s], c.nq, qo.max_hits_per_query, true, d_scan[s], d_off32[s], true, c.total, rest,
                                    d_hits[s], d_ws[s], st.k, qo);
                GDX_HIP(hipGetLastError());
                if (wire) {  // and the wire again, from the complete hits
                    exc_hits_cap[s] = c.total;
                    d_exc[s] = device_buf<uint8_t>(dev, s * 16 + 15, max_nq * 8 + exc_hits_cap[s] * sizeof(gdx_hit32_t));
                    pack_wire(s, c.nq);
                    GDX_HIP(hipStreamSynchronize(st.k));
                }
            }
            if (wire) {
                const uint32_t *meta = reinterpret_cast<const uint32_t *>(h_total[s] + 4);
                w_exc[s] = meta[0], w_exc_hits[s] = meta[1], w_found[s] = meta[2];
                w_status[s] = h_total[s][2] != 0;
                if (w_exc[s] > max_nq || w_exc_hits[s] > exc_hits_cap[s] || w_found[s] + w_exc_hits[s] != c.total)
                    fail(GDX_ERR_DEVICE, "internal: the wire of a chunk does not add up (%llu found + %llu exception hits, %llu hits)",
                         static_cast<unsigned long long>(w_found[s]), static_cast<unsigned long long>(w_exc_hits[s]),
                         static_cast<unsigned long long>(c.total));
                GDX_HIP(hipEventRecord(st.ev_k[s], st.k));
                GDX_HIP(hipStreamWaitEvent(st.out, st.ev_k[s], 0));
                GDX_HIP(hipMemcpyAsync(h_wire[s], d_wire[s], w_found_pos + w_found[s] * 4, hipMemcpyDeviceToHost, st.out));
                if (w_ids && w_found[s] != 0)
                    GDX_HIP(hipMemcpyAsync(h_wire[s] + w_found_ids, d_wire[s] + w_found_ids, w_found[s], hipMemcpyDeviceToHost, st.out));
                if (w_exc[s] != 0) {
                    h_exc[s] = pinned_buf<uint8_t>(dev, s * 16 + 8, w_exc[s] * 8 + w_exc_hits[s] * sizeof(gdx_hit32_t));
                    GDX_HIP(hipMemcpyAsync(h_exc[s], d_exc[s], w_exc[s] * 4, hipMemcpyDeviceToHost, st.out));
                    GDX_HIP(hipMemcpyAsync(h_exc[s] + w_exc[s] * 4, d_exc[s] + max_nq * 4, w_exc[s] * 4, hipMemcpyDeviceToHost, st.out));
                    if (w_exc_hits[s] != 0)
                        GDX_HIP(hipMemcpyAsync(h_exc[s] + w_exc[s] * 8, d_exc[s] + max_nq * 8, w_exc_hits[s] * sizeof(gdx_hit32_t),
                                               hipMemcpyDeviceToHost, st.out));
                }
                if (w_status[s]) GDX_HIP(hipMemcpyAsync(h_status[s], d_status[s], c.nq, hipMemcpyDeviceToHost, st.out));
                GDX_HIP(hipEventRecord(st.ev_out[s], st.out));
                c.hit_base = mid_hit_base;
                mid_hit_base += c.total;
                return;
            }
            if (mid_hit_base + c.total > narrow->cap) {  // the pinned hit array moves: nothing may be on its way into the old one
                GDX_HIP(hipStreamSynchronize(st.out));
                narrow->hits = narrow->grow(mid_hit_base + c.total, mid_hit_base, &narrow->cap);
            }
            if (mid_hit_base != 0)
                hipLaunchKernelGGL(add_hit_base_kernel, dim3(static_cast<unsigned>(std::min<uint64_t>((c.nq + 255) / 256, 2048))), dim3(256),
                                   0, st.k, d_off32[s] + 1, c.nq, static_cast<uint32_t>(mid_hit_base));
            GDX_HIP(hipEventRecord(st.ev_k[s], st.k));
            GDX_HIP(hipStreamWaitEvent(st.out, st.ev_k[s], 0));
            GDX_HIP(hipMemcpyAsync(narrow->offsets + c.q0 + 1, d_off32[s] + 1, c.nq * sizeof(uint32_t), hipMemcpyDeviceToHost, st.out));
            GDX_HIP(hipMemcpyAsync(h_status[s], d_status[s], c.nq, hipMemcpyDeviceToHost, st.out));
            if (c.total)
                GDX_HIP(hipMemcpyAsync(narrow->hits + mid_hit_base, d_hits[s], c.total * sizeof(gdx_hit32_t), hipMemcpyDeviceToHost, st.out));
            GDX_HIP(hipEventRecord(st.ev_out[s], st.out));
            c.hit_base = mid_hit_base;
            mid_hit_base += c.total;
            return;
        }
        if (plain_locate) {
            GDX_HIP(hipEventSynchronize(st.ev_total[s]));
            c.total = *h_total[s];
            if (c.total && !sizing_only) {
                d_hits[s] = device_buf<uint8_t>(dev, s * 16 + 8, c.total * hit_bytes);
                h_hits[s] = pinned_buf<uint8_t>(dev, s * 16 + 6, c.total * hit_bytes);
                d_ws[s] = device_buf<uint8_t>(dev, s * 16 + 9, locate_workspace_bytes(c.total));
                launch_locate(view_, nullptr, nullptr, c.nq, d_off[s], c.total, d_hits[s], wide_hits, d_ws[s], st.k, nullptr,
                              nullptr, qo, d_rec[s]);
                GDX_HIP(hipGetLastError());
            }
            GDX_HIP(hipEventRecord(st.ev_k[s], st.k));
        }
        if (flagged) {
            GDX_HIP(hipEventSynchronize(st.ev_total[s]));
            w_status[s] = h_total[s][0] != 0;
        }
        GDX_HIP(hipStreamWaitEvent(st.out, st.ev_k[s], 0));
        if (plain_locate)
            GDX_HIP(hipMemcpyAsync(h_off[s], d_off[s], (c.nq + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st.out));
        else
            GDX_HIP(hipMemcpyAsync(h_a[s], d_a[s], c.nq * sizeof(uint32_t), hipMemcpyDeviceToHost, st.out));
        if (kind == Kind::kIntervals)
            GDX_HIP(hipMemcpyAsync(h_b[s], d_b[s], c.nq * sizeof(uint32_t), hipMemcpyDeviceToHost, st.out));
        if (!flagged || w_status[s]) GDX_HIP(hipMemcpyAsync(h_status[s], d_status[s], c.nq, hipMemcpyDeviceToHost, st.out));
        if (plain_locate && c.total && !sizing_only)
            GDX_HIP(hipMemcpyAsync(h_hits[s], d_hits[s], c.total * hit_bytes, hipMemcpyDeviceToHost, st.out));
        GDX_HIP(hipEventRecord(st.ev_out[s], st.out));
    };

    // GDX_HOST_TIMING=1 (debug): where the three threads spend the call, in seconds
    static const bool timing = getenv("GDX_HOST_TIMING") != nullptr;
    double t_in_wait = 0, t_in_work = 0, t_mid_wait = 0, t_mid_work = 0, t_out_wait = 0, t_out_work = 0, t_out_sync = 0;
    auto stage_out = [&](size_t k) {  // pinned staging -> the caller's arrays, widened
        const int s = static_cast<int>(k % kSlots);
        Chunk &c = chunks[k];
        const double t_sync0 = timing ? now_seconds() : 0.0;
        GDX_HIP(hipEventSynchronize(st.ev_out[s]));
        if (timing) t_out_sync += now_seconds() - t_sync0;
        const uint32_t *a = h_a[s], *b = h_b[s];
        const uint8_t *stt = h_status[s];
        if (wire) {  // the chunk's wire -> offsets and hits in the caller's arrays, tiles shared out among the workers
            const uint64_t need = c.hit_base + c.total;
            const HostWire w{h_wire[s] + w_bitmap,
                             reinterpret_cast<const uint32_t *>(h_wire[s] + w_tile_found),
                             reinterpret_cast<const uint32_t *>(h_wire[s] + w_tile_off),
                             reinterpret_cast<const uint32_t *>(h_wire[s] + w_found_pos),
                             w_ids ? h_wire[s] + w_found_ids : nullptr,
                             reinterpret_cast<const uint32_t *>(h_exc[s]),
                             reinterpret_cast<const uint32_t *>(h_exc[s] + w_exc[s] * 4),
                             reinterpret_cast<const gdx_hit32_t *>(h_exc[s] + w_exc[s] * 8),
                             w_exc[s]};
            const uint64_t tiles = div_ceil(c.nq, kHostWireTile);
            auto share = [&](unsigned wk, unsigned nw, uint64_t &lo, uint64_t &hi) {
                const uint64_t per = div_ceil(tiles, nw);
                lo = std::min<uint64_t>(tiles, per * wk), hi = std::min<uint64_t>(tiles, lo + per);
            };
            if (kind == Kind::kLocate32) {
                if (need > narrow->cap) narrow->hits = narrow->grow(need, c.hit_base, &narrow->cap);  // (only this thread's workers write there)
                uint32_t *offs = narrow->offsets + c.q0;
                gdx_hit32_t *hh = narrow->hits;
                const uint32_t hb = static_cast<uint32_t>(c.hit_base);
                pool.run([&](unsigned wk, unsigned nw) {
                    uint64_t lo, hi;
                    share(wk, nw, lo, hi);
                    wire_expand_tiles<uint32_t, gdx_hit32_t>(w, c.nq, lo, hi, hb, offs, hh);
                });
            } else {  // u64 offsets and 16-byte hits; the hits only while the caller's buffer holds them (gdx_locate_many's sizing pass)
                if (grow_hits && need > hits_capacity) hits = (*grow_hits)(need, &hits_capacity);  // at least `need`
                if (!(hits && need <= hits_capacity && capacity_ok)) capacity_ok = false;
                uint64_t *offs = out_a ? out_a + c.q0 : nullptr;
                gdx_hit_t *hh = capacity_ok ? hits : nullptr;
                if (offs != nullptr || hh != nullptr)
                    pool.run([&](unsigned wk, unsigned nw) {
                        uint64_t lo, hi;
                        share(wk, nw, lo, hi);
                        wire_expand_tiles<uint64_t, gdx_hit_t>(w, c.nq, lo, hi, c.hit_base, offs, hh);
                    });
                hit_base = need;
            }
        } else if (kind == Kind::kLocate32) {
            // (offsets and hits are where they belong already)
        } else if (plain_locate) {
            c.hit_base = hit_base;
            if (out_a) {  // the chunk's offsets (scanned on the device) shifted by the hits of the chunks before it
                const uint64_t *off = h_off[s];
                const uint64_t shift_by = hit_base;
                pool.parallel_range(c.nq, 8, [&](uint64_t lo, uint64_t hi) {
                    for (uint64_t i = lo; i < hi; i++) out_a[c.q0 + i + 1] = off[i + 1] + shift_by;
                });
            }
            const uint64_t need = hit_base + c.total;
            if (grow_hits && need > hits_capacity) hits = (*grow_hits)(need, &hits_capacity);  // at least `need`
            if (hits && need <= hits_capacity && capacity_ok && !sizing_only) {
                gdx_hit_t *dst = hits + hit_base;
                if (wide_hits) {
                    const gdx_hit_t *src = static_cast<const gdx_hit_t *>(h_hits[s]);
                    pool.parallel_range(c.total, 8, [&](uint64_t lo, uint64_t hi) { std::memcpy(dst + lo, src + lo, (hi - lo) * sizeof(gdx_hit_t)); });
                } else {
                    const gdx_hit32_t *src = static_cast<const gdx_hit32_t *>(h_hits[s]);
                    pool.parallel_range(c.total, 8, [&](uint64_t lo, uint64_t hi) {
                        for (uint64_t i = lo; i < hi; i++) {
                            dst[i].text_id = src[i].text_id;
                            dst[i].position = src[i].position;
                        }
                    });
                }
            } else {
                capacity_ok = false;
            }
            hit_base = need;
        } else {
            pool.parallel_range(c.nq, 8, [&](uint64_t lo, uint64_t hi) {
                if (out_a) widen_u32(a + lo, hi - lo, out_a + c.q0 + lo);
                if (out_b) widen_u32(b + lo, hi - lo, out_b + c.q0 + lo);
            });
        }
        if ((wire || flagged) && !w_status[s]) {  // (no read of the chunk has a status: the bytes stayed on the device)
            if (out_status) pool.parallel_range(c.nq, 64, [&](uint64_t lo, uint64_t hi) { std::memset(out_status + c.q0 + lo, 0, hi - lo); });
            return;
        }
        pool.parallel_range(c.nq, 64, [&](uint64_t lo, uint64_t hi) {
            bool any = false;
            for (uint64_t i = lo; i < hi; i++) any |= stt[i] != 0;
            if (out_status) std::memcpy(out_status + c.q0 + lo, stt + lo, hi - lo);
            if (any) any_status.store(true);
        });
    };

    // Software pipeline with three host threads: the feeder stages chunks in (its own worker pool) as soon as a slot is
    // free; this thread waits for the search of chunk k (locate: for its number of hits), enqueues locate and the
    // copies out; the drainer widens chunk k - 1 into the caller's arrays (its own pool).  Copy-in, kernels, copy-out
    // and both host-side copies of different chunks overlap; with the middle and the drain stage on one thread the
    // locate call took 1.6-1.8 x the PCIe time of its input, the thread being busy widening hits while a finished
    // search waited for its locate launch.
    std::mutex mu;
    std::condition_variable cv;
    size_t staged = 0, launched = 0, drained = 0;  // chunks staged in / with their copy-out enqueued / fully drained
    bool abort = false;
    std::exception_ptr worker_error;
    // (a feeder that packs is the call's bound: it gets half as many workers again as the drainer -- 10 of a 16-CPU container's:
    // count 1.0 -> 2.1, wide locate 0.9 -> 1.8 G reads/s on one box; both pools at 10 or 14 were slower than both at 7.
    // GDX_HOST_IN_THREADS: experiments)
    const unsigned in_threads = [&] {
        if (const char *e = getenv("GDX_HOST_IN_THREADS")) return static_cast<unsigned>(std::max(1, atoi(e)));
        if (!host_pack || getenv("GDX_HOST_THREADS")) return host_threads();
        const unsigned cpus = usable_cpus();
        return std::max(host_threads(), std::min(host_threads() * 3u / 2u, cpus > 4u ? cpus - 4u : 1u));
    }();
    WorkerPool in_pool(in_threads);
    auto fail_all = [&](std::exception_ptr e) {
        std::lock_guard<std::mutex> g(mu);
        if (!worker_error) worker_error = e;
        abort = true;
        cv.notify_all();
    };
    std::thread feeder([&] {
        try {
            GDX_HIP(hipSetDevice(dev));
            for (size_t k = 0; k < n_chunks; k++) {
                const double t0 = timing ? now_seconds() : 0.0;
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return abort || k < drained + kSlots; });
                    if (abort) return;
                }
                const double t1 = timing ? now_seconds() : 0.0;
                stage_in_with(k, in_pool);
                if (timing) {
                    t_in_wait += t1 - t0;
                    t_in_work += now_seconds() - t1;
                }
                {
                    std::lock_guard<std::mutex> g(mu);
                    staged = k + 1;
                }
                cv.notify_all();
            }
        } catch (...) {
            fail_all(std::current_exception());
        }
    });
    std::thread drainer([&] {
        try {
            GDX_HIP(hipSetDevice(dev));
            for (size_t k = 0; k < n_chunks; k++) {
                const double t0 = timing ? now_seconds() : 0.0;
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return abort || launched > k; });
                    if (abort) return;
                }
                const double t1 = timing ? now_seconds() : 0.0;
                stage_out(k);
                if (timing) {
                    t_out_wait += t1 - t0;
                    t_out_work += now_seconds() - t1;
                }
                {
                    std::lock_guard<std::mutex> g(mu);
                    drained = k + 1;
                }
                cv.notify_all();
            }
        } catch (...) {
            fail_all(std::current_exception());
        }
    });
    try {
        for (size_t k = 0; k < n_chunks; k++) {
            const double t0 = timing ? now_seconds() : 0.0;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return abort || staged > k; });
                if (abort) break;
            }
            const double t1 = timing ? now_seconds() : 0.0;
            stage_mid(k);
            if (timing) {
                t_mid_wait += t1 - t0;
                t_mid_work += now_seconds() - t1;
            }
            {
                std::lock_guard<std::mutex> g(mu);
                launched = k + 1;
            }
            cv.notify_all();
        }
    } catch (...) {
        fail_all(std::current_exception());
    }
    feeder.join();
    drainer.join();
    if (timing)
        fprintf(stderr, "gdx host pipeline: %zu chunks; feeder wait %.3f work %.3f; launcher wait %.3f work %.3f; drainer wait %.3f work %.3f (of it %.3f waiting for the copies out)\n",
                n_chunks, t_in_wait, t_in_work, t_mid_wait, t_mid_work, t_out_wait, t_out_work, t_out_sync);
    if (worker_error) {
        (void)hipDeviceSynchronize();
        std::rethrow_exception(worker_error);
    }
    GDX_HIP(hipStreamSynchronize(st.out));
    if (kind == Kind::kLocate32) {
        narrow->offsets[0] = 0;
        if (out_total) *out_total = mid_hit_base;
        return any_status.load() ? GDX_ERR_QUERY_STATUS : GDX_OK;
    }
    if (out_total) *out_total = hit_base;
    if (kind == Kind::kLocate && hit_base > 0 && !capacity_ok) return GDX_ERR_CAPACITY;
    return any_status.load() ? GDX_ERR_QUERY_STATUS : GDX_OK;
}

// ASCII -> 2-bit on the host, by all pool threads; exceptions = the queries that hold a symbol which is not one of
// the dense codes 1..4 (sorted, unique); returns how many there are (the first `capacity` of them are stored)
uint64_t FmIndex::pack_queries_host(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint8_t *out_packed,
                                    uint64_t *out_exc, uint64_t capacity) const
{
    return pack_queries_with_table(cfg_.io_to_dense, qbuf, qoff, nq, out_packed, out_exc, capacity);
}

// (no index, no device: the alphabet's table is all the packing needs -- gdx_pack_queries_table, for a reader that packs
// what it parses before any index is at hand)
uint64_t pack_queries_with_table(const uint8_t *tab, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint8_t *out_packed,
                                 uint64_t *out_exc, uint64_t capacity)
{
    if (!tab) fail(GDX_ERR_INVALID_ARGUMENT, "io_to_dense is null");
    // (a call of its own, not a stage of the pipeline: every CPU the process may use, unless GDX_HOST_THREADS says otherwise)
    WorkerPool pool(getenv("GDX_HOST_THREADS") ? host_threads() : std::min(32u, usable_cpus()));
    check_queries(qbuf, qoff, nq, pool, 0);
    if (!out_packed && nq && qoff[nq]) fail(GDX_ERR_INVALID_ARGUMENT, "out_packed is null");
    const uint64_t n_sym = nq ? qoff[nq] : 0;
    const uint64_t n_bytes = div_ceil(n_sym, 4);
    const uint64_t first = nq ? qoff[0] : 0;  // symbols before the first query are not looked at (code 0)
    std::vector<std::vector<uint64_t>> bad(pool.size());
    const PackPlan plan = make_pack_plan(tab);  // (pack_host.hpp: 32 symbols per step where the table has the shape for it)
    pool.run([&](unsigned w, unsigned nw) {
        const uint64_t per = (n_bytes / nw + 64) / 64 * 64;
        const uint64_t lo = std::min(n_bytes, per * w), hi = std::min(n_bytes, lo + per);
        std::vector<uint64_t> &mine = bad[w];
        pack_range(plan, tab, qbuf, first, n_sym, lo, hi, out_packed, [&](uint64_t j) {
            const uint64_t q = static_cast<uint64_t>(std::upper_bound(qoff, qoff + nq + 1, j) - qoff) - 1;
            if (mine.empty() || mine.back() != q) mine.push_back(q);
        });
    });
    std::vector<uint64_t> all;
    for (auto &v : bad) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    for (uint64_t i = 0; i < all.size() && i < capacity; i++) out_exc[i] = all[i];
    return all.size();
}

unsigned fastx_default_threads() { return std::min(32u, usable_cpus()); }

void set_host_chunking(uint64_t queries, uint64_t bytes)
{
    g_chunk_queries.store(queries ? queries : (1ull << 20));
    g_chunk_bytes.store(bytes ? bytes : (32ull << 20));
}

int FmIndex::cursors_for_many_queries(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_start,
                                      uint64_t *out_end, uint64_t *out_count, uint8_t *out_status, bool packed,
                                      uint64_t uniform_len) const
{
    if (packed && (view_.layout != 0 || view_.n_searchable < 4))
        fail(GDX_ERR_UNSUPPORTED, "packed queries need the rank-line layout (sigma <= 8) with dense symbols 1..4 searchable");
    if (out_start || out_end) {
        const int rc = host_pipeline(static_cast<int>(Kind::kIntervals), qbuf, qoff, nq, out_start, out_end, out_status,
                                     nullptr, 0, nullptr, nullptr, packed, uniform_len);
        if (out_count && out_start && out_end)
            for (uint64_t i = 0; i < nq; i++) out_count[i] = out_end[i] - out_start[i];
        return rc;
    }
    return host_pipeline(static_cast<int>(Kind::kCounts), qbuf, qoff, nq, out_count, nullptr, out_status, nullptr, 0,
                         nullptr, nullptr, packed, uniform_len);
}

int FmIndex::locate_many(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets,
                         gdx_hit_t *hits, uint64_t hits_capacity, uint64_t *out_total, uint8_t *out_status) const
{
    try {
        return host_pipeline(static_cast<int>(Kind::kLocate), qbuf, qoff, nq, out_hit_offsets, nullptr, out_status, hits,
                             hits ? hits_capacity : 0, out_total, nullptr);
    } catch (const RetryWithWideOffsets &) {  // a chunk with more hits than the fused step's 32-bit offsets hold
        (void)hipDeviceSynchronize();
        struct Restore {
            ~Restore() { t_force_plain_locate = false; }
        } restore;
        t_force_plain_locate = true;
        return host_pipeline(static_cast<int>(Kind::kLocate), qbuf, qoff, nq, out_hit_offsets, nullptr, out_status, hits,
                             hits ? hits_capacity : 0, out_total, nullptr);
    }
}

// One hit array released with gdx_free_hits is kept for the next gdx_locate_many_alloc: a caller that locates batch
// after batch otherwise pays the first touch of 1.6 GB per 100 M hits in every call (page faults and zeroing inside the
// drainer's copy: 0.08-0.14 of a 0.14 s call, tools/host_timing.py).  At most one array is held, the larger of the two
// when another comes back; it is ordinary malloc memory (a caller may free() it as well).
namespace {
std::mutex g_hits_mutex;
void *g_hits_cached = nullptr;
size_t g_hits_cached_bytes = 0;
constexpr size_t kHitsCacheMin = 16u << 20;  // smaller arrays are not worth holding
}  // namespace

void recycle_hits(gdx_hit_t *hits)
{
    if (!hits) return;
    const size_t bytes = malloc_usable_size(hits);
    void *drop = hits;
    if (bytes >= kHitsCacheMin) {
        std::lock_guard<std::mutex> g(g_hits_mutex);
        if (bytes > g_hits_cached_bytes) {
            drop = g_hits_cached;
            g_hits_cached = hits;
            g_hits_cached_bytes = bytes;
        }
    }
    std::free(drop);
}

static void *take_cached_hits(size_t bytes, size_t *got)
{
    std::lock_guard<std::mutex> g(g_hits_mutex);
    // (only for a request of at least half its size: a ten-query call must not walk away with gigabytes)
    if (g_hits_cached == nullptr || g_hits_cached_bytes < bytes || g_hits_cached_bytes / 2 > bytes) return nullptr;
    void *p = g_hits_cached;
    *got = g_hits_cached_bytes;
    g_hits_cached = nullptr;
    g_hits_cached_bytes = 0;
    return p;
}

int FmIndex::locate_many_alloc(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets,
                               gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status, bool packed,
                               uint64_t uniform_len) const
{
    if (packed && (view_.layout != 0 || view_.n_searchable < 4))
        fail(GDX_ERR_UNSUPPORTED, "packed queries need the rank-line layout (sigma <= 8) with dense symbols 1..4 searchable");
    if (!out_hits) fail(GDX_ERR_INVALID_ARGUMENT, "out_hits is null");
    *out_hits = nullptr;
    gdx_hit_t *buf = nullptr;
    uint64_t cap = 0;
    // The hit array is large (16 bytes per hit) and written exactly once: 2 MB alignment + MADV_HUGEPAGE keeps the
    // first-touch page faults (one per 4 KB otherwise, ~100 ms per GB single-threaded) out of the pipeline.
    const std::function<gdx_hit_t *(uint64_t, uint64_t *)> grow = [&](uint64_t need, uint64_t *new_cap) {
        const uint64_t want = std::max<uint64_t>(need, cap + cap / 2 + 4096);
        size_t bytes = (want * sizeof(gdx_hit_t) + (2u << 20) - 1) / (2u << 20) * (2u << 20);
        void *p = take_cached_hits(bytes, &bytes);  // (an array a caller gave back: its pages are there already)
        if (p == nullptr) {
            if (posix_memalign(&p, 2u << 20, bytes) != 0 || !p) {
                std::free(buf);
                buf = nullptr;
                fail(GDX_ERR_DEVICE, "out of host memory for %llu hits", static_cast<unsigned long long>(want));
            }
            (void)madvise(p, bytes, MADV_HUGEPAGE);
        }
        if (buf) {  // rare: the first guess (one hit per query and a bit) was too small
            std::memcpy(p, buf, cap * sizeof(gdx_hit_t));
            std::free(buf);
        }
        buf = static_cast<gdx_hit_t *>(p);
        cap = bytes / sizeof(gdx_hit_t);
        if (new_cap) *new_cap = cap;
        return buf;
    };
    int rc;
    try {
        // a first guess from the batch size spares most reallocations (one hit per query is the common shape)
        grow(nq + nq / 8, nullptr);
        try {
            rc = host_pipeline(static_cast<int>(Kind::kLocate), qbuf, qoff, nq, out_hit_offsets, nullptr, out_status, buf, cap,
                               out_total, &grow, packed, uniform_len);
        } catch (const RetryWithWideOffsets &) {  // (as in locate_many)
            (void)hipDeviceSynchronize();
            struct Restore {
                ~Restore() { t_force_plain_locate = false; }
            } restore;
            t_force_plain_locate = true;
            rc = host_pipeline(static_cast<int>(Kind::kLocate), qbuf, qoff, nq, out_hit_offsets, nullptr, out_status, buf, cap,
                               out_total, &grow, packed, uniform_len);
        }
    } catch (...) {
        std::free(buf);
        throw;
    }
    *out_hits = buf;
    return rc;
}

// ---- narrow results in library-owned pinned memory (gdx_locate_many_alloc_layout32) ---------------------------------------
// The device writes u32 offsets and 8-byte hits into these arrays by D2H copy; the caller reads them where they are.  Pinning
// host memory costs about a millisecond per 10 MB, so the arrays a caller gives back (gdx_free_hits32) are kept for its next
// call -- one pair, the largest seen -- until gdx_release_cached_hits.
namespace {
struct PinnedBlock {
    void *ptr = nullptr;
    size_t bytes = 0;
};
std::mutex g_pinned_mutex;
PinnedBlock g_pinned_cache[2];  // [0] offsets, [1] hits

void *pinned_take(int which, size_t bytes, size_t *got)
{
    {
        std::lock_guard<std::mutex> g(g_pinned_mutex);
        PinnedBlock &b = g_pinned_cache[which];
        if (b.ptr != nullptr && b.bytes >= bytes) {
            void *p = b.ptr;
            *got = b.bytes;
            b = PinnedBlock{};
            return p;
        }
    }
    void *p = nullptr;
    GDX_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    *got = bytes;
    return p;
}

void pinned_give(int which, void *ptr, size_t bytes)
{
    if (ptr == nullptr) return;
    void *drop = ptr;
    {
        std::lock_guard<std::mutex> g(g_pinned_mutex);
        PinnedBlock &b = g_pinned_cache[which];
        if (bytes > b.bytes) {
            drop = b.ptr;
            b.ptr = ptr;
            b.bytes = bytes;
        }
    }
    if (drop != nullptr) (void)hipHostFree(drop);
}
}  // namespace

int FmIndex::locate_many_alloc32(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, gdx_hits32_t *out, uint8_t *out_status,
                                 bool packed, uint64_t uniform_len) const
{
    if (packed && (view_.layout != 0 || view_.n_searchable < 4))
        fail(GDX_ERR_UNSUPPORTED, "packed queries need the rank-line layout (sigma <= 8) with dense symbols 1..4 searchable");
    if (!out) fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
    std::memset(out, 0, sizeof(*out));
    make_current();
    size_t off_bytes = 0, hit_bytes = 0;
    Narrow32Sink sink;
    sink.offsets = static_cast<uint32_t *>(pinned_take(0, (nq + 1) * sizeof(uint32_t) + 64, &off_bytes));
    try {
        // a first guess from the batch size spares most reallocations (one hit per query is the common shape)
        sink.hits = static_cast<gdx_hit32_t *>(pinned_take(1, (nq + nq / 8 + 4096) * sizeof(gdx_hit32_t), &hit_bytes));
    } catch (...) {
        pinned_give(0, sink.offsets, off_bytes);
        throw;
    }
    sink.cap = hit_bytes / sizeof(gdx_hit32_t);
    sink.grow = [&](uint64_t need, uint64_t keep, uint64_t *new_cap) {
        size_t bytes = 0;
        gdx_hit32_t *p = static_cast<gdx_hit32_t *>(pinned_take(1, std::max<uint64_t>(need, sink.cap + sink.cap / 2) * sizeof(gdx_hit32_t), &bytes));
        std::memcpy(p, sink.hits, keep * sizeof(gdx_hit32_t));
        pinned_give(1, sink.hits, hit_bytes);
        hit_bytes = bytes;
        *new_cap = bytes / sizeof(gdx_hit32_t);
        return p;
    };
    uint64_t total = 0;
    int rc;
    try {
        rc = host_pipeline(static_cast<int>(Kind::kLocate32), qbuf, qoff, nq, nullptr, nullptr, out_status, nullptr, 0, &total, nullptr,
                           packed, uniform_len, &sink);
    } catch (...) {
        (void)hipDeviceSynchronize();
        pinned_give(0, sink.offsets, off_bytes);
        pinned_give(1, sink.hits, hit_bytes);
        throw;
    }
    out->hit_offsets = sink.offsets;
    out->hits = sink.hits;
    out->total_hits = total;
    out->nq = nq;
    out->reserved[0] = off_bytes;
    out->reserved[1] = hit_bytes;
    return rc;
}

void recycle_hits32(gdx_hits32_t *r)
{
    if (!r) return;
    pinned_give(0, r->hit_offsets, r->reserved[0]);
    pinned_give(1, r->hits, r->reserved[1]);
    std::memset(r, 0, sizeof(*r));
}

void release_cached_hits()
{
    void *drop[3] = {nullptr, nullptr, nullptr};
    {
        std::lock_guard<std::mutex> g(g_pinned_mutex);
        for (int i = 0; i < 2; i++) {
            drop[i] = g_pinned_cache[i].ptr;
            g_pinned_cache[i] = PinnedBlock{};
        }
    }
    {
        std::lock_guard<std::mutex> g(g_hits_mutex);
        drop[2] = g_hits_cached;
        g_hits_cached = nullptr;
        g_hits_cached_bytes = 0;
    }
    if (drop[0]) (void)hipHostFree(drop[0]);
    if (drop[1]) (void)hipHostFree(drop[1]);
    std::free(drop[2]);
}

}  // namespace gdx
