// multi.hip -- several index replicas, one per GPU of a node, behind one handle (gdx_multi_*): the multi-GPU path of
// SURVEY.md section 8e for a host that is NOT a set of torch.distributed ranks (the Rust host of the north star): one
// process, one thread per device.  The batch is cut into contiguous shards (shard r = queries [nq r / g, nq (r + 1) /
// g)), every replica runs the chunked H2D || kernels || D2H pipeline of host_api.hip on its shard over its own PCIe
// link and writes its results straight into the caller's arrays at the shard's offset; there is no exchange between
// the devices at all (results live in host memory).  Order preserving: the output equals the one-GPU output.
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include <sys/mman.h>

#include "fm_index.hpp"

namespace gdx {

struct ReplicaWorker::Impl {
    std::mutex m;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> jobs;
    size_t running = 0;
    bool stop = false;
    std::thread thread;
};

ReplicaWorker::ReplicaWorker() : impl_(new Impl)
{
    Impl *p = impl_.get();
    p->thread = std::thread([p] {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> g(p->m);
                p->cv.wait(g, [p] { return p->stop || !p->jobs.empty(); });
                if (p->jobs.empty()) return;  // stop requested and nothing left
                job = std::move(p->jobs.front());
                p->jobs.pop_front();
                p->running++;
            }
            job();
            {
                std::lock_guard<std::mutex> g(p->m);
                p->running--;
            }
            p->idle.notify_all();
        }
    });
}

ReplicaWorker::~ReplicaWorker()
{
    {
        std::lock_guard<std::mutex> g(impl_->m);
        impl_->stop = true;
    }
    impl_->cv.notify_all();
    if (impl_->thread.joinable()) impl_->thread.join();
}

void ReplicaWorker::submit(std::function<void()> job)
{
    {
        std::lock_guard<std::mutex> g(impl_->m);
        impl_->jobs.push_back(std::move(job));
    }
    impl_->cv.notify_one();
}

void ReplicaWorker::wait()
{
    std::unique_lock<std::mutex> g(impl_->m);
    impl_->idle.wait(g, [this] { return impl_->jobs.empty() && impl_->running == 0; });
}

void Multi::start_workers()
{
    workers.clear();
    for (size_t r = 0; r < replicas.size(); r++) workers.push_back(std::make_unique<ReplicaWorker>());
}

namespace {

struct ShardResult {
    int rc = GDX_OK;
    std::string error;
    gdx_hit_t *hits = nullptr;
    uint64_t total = 0;
};

// per_shard(r, lo, hi) on the worker thread of every replica (calls on one handle are serialised by the caller's
// use of the workers: a second concurrent call queues behind the first)
template <class F>
int run_shards(Multi &m, uint64_t nq, F per_shard)
{
    const size_t g = m.replicas.size();
    // one call at a time per handle: the workers' queues then hold this call's jobs only, so waiting for them to go
    // idle is waiting for this call
    std::lock_guard<std::recursive_mutex> serial(m.call_mutex);
    std::vector<ShardResult> res(g);
    size_t submitted = 0;
    try {
        for (size_t r = 0; r < g; r++) {
            m.worker(r).submit([&, r] {
                const uint64_t lo = nq * r / g, hi = nq * (r + 1) / g;
                try {
                    res[r].rc = per_shard(r, lo, hi, res[r]);
                } catch (const Error &e) {
                    res[r].rc = e.status;
                    res[r].error = e.what();
                } catch (const std::exception &e) {
                    res[r].rc = GDX_ERR_DEVICE;
                    res[r].error = e.what();
                }
            });
            submitted++;
        }
    } catch (...) {  // (a submit that ran out of memory) the jobs already queued still reference res: let them finish
        for (size_t r = 0; r < submitted; r++) m.worker(r).wait();
        throw;
    }
    for (size_t r = 0; r < g; r++) m.worker(r).wait();
    int rc = GDX_OK;
    for (size_t r = 0; r < g; r++) {
        if (res[r].rc != GDX_OK && res[r].rc != GDX_ERR_QUERY_STATUS) fail(res[r].rc, "replica %zu: %s", r, res[r].error.c_str());
        if (res[r].rc == GDX_ERR_QUERY_STATUS) rc = GDX_ERR_QUERY_STATUS;
    }
    return rc;
}

}  // namespace

int multi_cursors(Multi &m, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_start,
                  uint64_t *out_end, uint64_t *out_count, uint8_t *out_status)
{
    if (!qoff) fail(GDX_ERR_INVALID_ARGUMENT, "qoff is null");
    return run_shards(m, nq, [&](size_t r, uint64_t lo, uint64_t hi, ShardResult &) {
        return m.replicas[r]->cursors_for_many_queries(qbuf, qoff + lo, hi - lo, out_start ? out_start + lo : nullptr,
                                                       out_end ? out_end + lo : nullptr,
                                                       out_count ? out_count + lo : nullptr,
                                                       out_status ? out_status + lo : nullptr);
    });
}

int multi_locate_alloc(Multi &m, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets,
                       gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status)
{
    if (!qoff || !out_hits || !out_hit_offsets) fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
    *out_hits = nullptr;
    if (out_total) *out_total = 0;
    const size_t g = m.replicas.size();
    std::vector<gdx_hit_t *> shard_hits(g, nullptr);
    std::vector<uint64_t> shard_total(g, 0);
    int rc;
    try {
        // every shard writes its LOCAL offsets into its own window of out_hit_offsets (shifted by one so that the
        // windows do not overlap: window r = out_hit_offsets[lo + 1 .. hi + 1), local offset of its first query
        // is 0 and not stored); they are made global below
        rc = run_shards(m, nq, [&](size_t r, uint64_t lo, uint64_t hi, ShardResult &) {
            std::vector<uint64_t> local(hi - lo + 1);
            gdx_hit_t *h = nullptr;
            uint64_t total = 0;
            const int s = m.replicas[r]->locate_many_alloc(qbuf, qoff + lo, hi - lo, local.data(), &h, &total,
                                                           out_status ? out_status + lo : nullptr);
            shard_hits[r] = h;
            shard_total[r] = total;
            std::memcpy(out_hit_offsets + lo + 1, local.data() + 1, (hi - lo) * sizeof(uint64_t));
            return s;
        });
        uint64_t total = 0;
        std::vector<uint64_t> base(g + 1, 0);
        for (size_t r = 0; r < g; r++) base[r + 1] = base[r] + shard_total[r];
        total = base[g];
        out_hit_offsets[0] = 0;
        gdx_hit_t *all = nullptr;
        if (total) {
            const size_t bytes = (total * sizeof(gdx_hit_t) + (2u << 20) - 1) / (2u << 20) * (2u << 20);
            void *p = nullptr;
            if (posix_memalign(&p, 2u << 20, bytes) != 0 || !p) fail(GDX_ERR_DEVICE, "out of host memory for %llu hits", (unsigned long long)total);
            (void)madvise(p, bytes, MADV_HUGEPAGE);
            all = static_cast<gdx_hit_t *>(p);
        }
        // global offsets and the concatenated hit array, one thread per shard
        std::vector<std::thread> threads;
        for (size_t r = 0; r < g; r++) {
            threads.emplace_back([&, r] {
                const uint64_t lo = nq * r / g, hi = nq * (r + 1) / g;
                if (base[r] != 0)
                    for (uint64_t i = lo + 1; i <= hi; i++) out_hit_offsets[i] += base[r];
                if (shard_total[r]) std::memcpy(all + base[r], shard_hits[r], shard_total[r] * sizeof(gdx_hit_t));
            });
        }
        for (auto &t : threads) t.join();
        *out_hits = all;
        if (out_total) *out_total = total;
    } catch (...) {
        for (gdx_hit_t *h : shard_hits) std::free(h);
        throw;
    }
    for (gdx_hit_t *h : shard_hits) std::free(h);
    return rc;
}

}  // namespace gdx

// =====================================================================================================================
// Device-resident shards, results gathered to one GPU with RCCL point-to-point transfers (gdx_multi_locate_many_gather_dev)

#include <dlfcn.h>

namespace gdx {

uint64_t FmIndex::locate_shard_dev(const uint8_t *d_qbuf, const uint64_t *d_qoff, uint64_t nq, DeviceBuffer<uint32_t> &counts,
                                   DeviceBuffer<uint8_t> &status, DeviceBuffer<gdx_hit32_t> &hits, hipStream_t stream) const
{
    if (nq == 0) return 0;
    if (nq >= 0xffffffffull) fail(GDX_ERR_UNSUPPORTED, "more than 2^32-2 queries in one shard");
    const QueryOptions qo = query_options();
    DeviceBuffer<uint4> rec(nq);
    DeviceBuffer<uint64_t> off(nq + 1);
    const size_t scan_bytes = hit_offsets_rec_temp_bytes(nq);
    DeviceBuffer<uint8_t> scan(scan_bytes ? scan_bytes : 1);
    if (counts.count < nq) counts.alloc(nq);
    if (status.count < nq) status.alloc(nq);
    SearchCall call;
    call.d_qbuf = d_qbuf;
    call.d_qbeg = d_qoff;
    call.d_qend = d_qoff + 1;
    call.nq = nq;
    call.d_rec = rec.get();
    call.mode = 1;
    launch_search_call(view_, call, stream, qo);
    GDX_HIP(hipGetLastError());
    launch_unpack_records(rec.get(), nq, counts.get(), status.get(), stream);
    launch_hit_offsets_rec(rec.get(), nq, off.get(), scan.get(), scan_bytes, stream);
    uint64_t total = 0;
    GDX_HIP(hipMemcpyAsync(&total, off.get() + nq, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    GDX_HIP(hipStreamSynchronize(stream));
    if (total) {
        if (hits.count < total) hits.alloc(total);
        DeviceBuffer<uint8_t> ws(locate_workspace_bytes(total));
        launch_locate(view_, nullptr, nullptr, nq, off.get(), total, hits.get(), false, ws.get(), stream, nullptr, nullptr, qo,
                      rec.get());
        GDX_HIP(hipGetLastError());
        GDX_HIP(hipStreamSynchronize(stream));
    }
    return total;
}

namespace {

// the few entry points of RCCL this needs, resolved at first use: libgdx.so has no link-time dependency on it (a box
// without RCCL, or a single-GPU one, never loads it)
struct Rccl {
    using comm_t = void *;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    Rccl()
    {
        void *h = nullptr;
        // (GDX_RCCL_LIBRARY: another library with RCCL's entry points first -- tests/rccl_shim records what this file sends where
        // and moves the bytes itself, so that the exchange below runs on a box with one GPU)
        if (const char *e = getenv("GDX_RCCL_LIBRARY")) h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if (h == nullptr && (h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!h) return;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(h, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(dlsym(h, "ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(dlsym(h, "ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
    }
};
Rccl &rccl()
{
    static Rccl r;
    return r;
}
constexpr int kNcclUint8 = 1, kNcclUint32 = 3;  // ncclDataType_t (rccl.h)

void nccl_check(int rc, const char *what)
{
    if (rc != 0) fail(GDX_ERR_DEVICE, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
}

}  // namespace

Multi::~Multi()
{
    if (!comms.empty() && rccl().ok)
        for (void *c : comms)
            if (c) (void)rccl().CommDestroy(c);
    if (g_device >= 0 && hipSetDevice(g_device) == hipSuccess) {
        g_counts.release();
        g_offsets.release();
        g_hits.release();
        g_status.release();
        g_scan.release();
    }
}

void multi_locate_gather_dev(Multi &m, const DeviceShard *shards, int n_shards, int root, Gathered *out)
{
    const size_t g = m.replicas.size();
    if (!shards || !out || n_shards != static_cast<int>(g)) fail(GDX_ERR_INVALID_ARGUMENT, "one shard per replica is expected");
    if (root < 0 || root >= n_shards) fail(GDX_ERR_INVALID_ARGUMENT, "root replica out of range");
    std::vector<int> dev(g);
    bool all_same = true, all_distinct = true;
    for (size_t r = 0; r < g; r++) {
        dev[r] = m.replicas[r]->config().device_id;
        all_same &= dev[r] == dev[0];
        for (size_t k = 0; k < r; k++) all_distinct &= dev[k] != dev[r];
        if (shards[r].nq != 0 && (!shards[r].d_qoff || !shards[r].d_qbuf)) fail(GDX_ERR_INVALID_ARGUMENT, "shard %zu: null pointer", r);
        // (the search kernels read the query bytes as aligned 8-byte words, like gdx_locate_many_search_compact_dev)
        if (shards[r].nq != 0 && (reinterpret_cast<uintptr_t>(shards[r].d_qbuf) & 7u) != 0)
            fail(GDX_ERR_INVALID_ARGUMENT, "shard %zu: d_qbuf must be 8-byte aligned", r);
    }
    std::lock_guard<std::recursive_mutex> serial(m.call_mutex);  // the handle's result buffers are this call's until it returns
    // (GDX_MULTI_FORCE_RCCL=1, tests: replicas that share a device exchange through the RCCL entry points all the same)
    const bool force_rccl = [] { const char *e = getenv("GDX_MULTI_FORCE_RCCL"); return e && atoi(e) != 0; }();
    const bool use_rccl = g > 1 && (!all_same || force_rccl);
    if (use_rccl && !all_distinct && !force_rccl) fail(GDX_ERR_UNSUPPORTED, "replicas must sit on distinct devices, or all on one");
    if (use_rccl && !rccl().ok) fail(GDX_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded");

    // every replica: search -> scan -> locate of its shard on its own device and stream (the replica's worker thread)
    // (the buffers and the stream of a replica are released on its device whichever way this function is left: a failing
    // shard throws out of run_shards)
    struct Local {
        DeviceBuffer<uint32_t> counts;
        DeviceBuffer<uint8_t> status;
        DeviceBuffer<gdx_hit32_t> hits;
        hipStream_t stream = nullptr;
        uint64_t total = 0;
        int device = -1;
        Local() = default;
        Local(const Local &) = delete;
        Local &operator=(const Local &) = delete;
        ~Local()
        {
            if (device < 0 || hipSetDevice(device) != hipSuccess) return;
            if (stream) (void)hipStreamSynchronize(stream);
            counts.release();
            status.release();
            hits.release();
            if (stream) (void)hipStreamDestroy(stream);
        }
    };
    std::vector<Local> loc(g);
    for (size_t r = 0; r < g; r++) loc[r].device = dev[r];
    const int rc = run_shards(m, 0, [&](size_t r, uint64_t, uint64_t, ShardResult &) {
        GDX_HIP(hipSetDevice(dev[r]));
        GDX_HIP(hipStreamCreateWithFlags(&loc[r].stream, hipStreamNonBlocking));
        loc[r].total = m.replicas[r]->locate_shard_dev(shards[r].d_qbuf, shards[r].d_qoff, shards[r].nq, loc[r].counts,
                                                       loc[r].status, loc[r].hits, loc[r].stream);
        return static_cast<int>(GDX_OK);
    });
    // (GDX_ERR_QUERY_STATUS: some query has a status byte set -- they are gathered into out->d_status like the counts)
    if (rc != GDX_OK && rc != GDX_ERR_QUERY_STATUS) fail(static_cast<gdx_status>(rc), "a shard failed");
    // where every shard lands on the root
    std::vector<uint64_t> qbase(g + 1, 0), hbase(g + 1, 0);
    for (size_t r = 0; r < g; r++) {
        qbase[r + 1] = qbase[r] + shards[r].nq;
        hbase[r + 1] = hbase[r] + loc[r].total;
    }
    const uint64_t nq = qbase[g], total = hbase[g];
    GDX_HIP(hipSetDevice(dev[root]));
    if (m.g_device != dev[root]) {
        m.g_counts.release();
        m.g_offsets.release();
        m.g_hits.release();
        m.g_status.release();
        m.g_scan.release();
        m.g_device = dev[root];
    }
    if (m.g_counts.count < nq + 1) m.g_counts.alloc(nq + 1);
    if (m.g_status.count < nq + 1) m.g_status.alloc(nq + 1);
    if (m.g_offsets.count < nq + 1) m.g_offsets.alloc(nq + 1);
    if (m.g_hits.count < total + 1) m.g_hits.alloc(total + 1);
    hipStream_t root_stream = loc[root].stream;
    if (use_rccl) {
        if (m.comms.empty()) {
            m.comms.assign(g, nullptr);
            nccl_check(rccl().CommInitAll(m.comms.data(), static_cast<int>(g), dev.data()), "ncclCommInitAll");
        }
        // one group: the root posts a receive per shard and array, every other replica the matching sends -- the
        // transfers of all shards run at once over their own xGMI links
        nccl_check(rccl().GroupStart(), "ncclGroupStart");
        for (size_t r = 0; r < g; r++) {
            if (static_cast<int>(r) == root) continue;
            const int rr = static_cast<int>(r);
            if (shards[r].nq) {
                nccl_check(rccl().Send(loc[r].counts.get(), shards[r].nq, kNcclUint32, root, m.comms[r], loc[r].stream), "ncclSend");
                nccl_check(rccl().Recv(m.g_counts.get() + qbase[r], shards[r].nq, kNcclUint32, rr, m.comms[root], root_stream), "ncclRecv");
                nccl_check(rccl().Send(loc[r].status.get(), shards[r].nq, kNcclUint8, root, m.comms[r], loc[r].stream), "ncclSend");
                nccl_check(rccl().Recv(m.g_status.get() + qbase[r], shards[r].nq, kNcclUint8, rr, m.comms[root], root_stream), "ncclRecv");
            }
            if (loc[r].total) {
                nccl_check(rccl().Send(loc[r].hits.get(), loc[r].total * 2, kNcclUint32, root, m.comms[r], loc[r].stream), "ncclSend");
                nccl_check(rccl().Recv(m.g_hits.get() + hbase[r], loc[r].total * 2, kNcclUint32, rr, m.comms[root], root_stream), "ncclRecv");
            }
        }
        nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
    }
    for (size_t r = 0; r < g; r++) {  // the root's own shard, and every shard when all replicas share the device
        if (use_rccl && static_cast<int>(r) != root) continue;
        if (shards[r].nq) {
            GDX_HIP(hipMemcpyAsync(m.g_counts.get() + qbase[r], loc[r].counts.get(), shards[r].nq * sizeof(uint32_t), hipMemcpyDeviceToDevice, root_stream));
            GDX_HIP(hipMemcpyAsync(m.g_status.get() + qbase[r], loc[r].status.get(), shards[r].nq, hipMemcpyDeviceToDevice, root_stream));
        }
        if (loc[r].total)
            GDX_HIP(hipMemcpyAsync(m.g_hits.get() + hbase[r], loc[r].hits.get(), loc[r].total * sizeof(gdx_hit32_t), hipMemcpyDeviceToDevice, root_stream));
    }
    const size_t scan_bytes = count_offsets_temp_bytes(nq);
    if (m.g_scan.count < scan_bytes + 1) m.g_scan.alloc(scan_bytes + 1);
    launch_count_offsets(m.g_counts.get(), nq, m.g_offsets.get(), m.g_scan.get(), scan_bytes, root_stream);
    GDX_HIP(hipGetLastError());
    for (size_t r = 0; r < g; r++) {  // every sender's buffers must outlive its transfers
        GDX_HIP(hipSetDevice(dev[r]));
        GDX_HIP(hipStreamSynchronize(loc[r].stream));
    }
    loc.clear();  // (releases every replica's buffers and stream on its device)
    GDX_HIP(hipSetDevice(dev[root]));
    out->d_counts = m.g_counts.get();
    out->d_hit_offsets = m.g_offsets.get();
    out->d_hits = m.g_hits.get();
    out->d_status = m.g_status.get();
    out->nq = nq;
    out->total_hits = total;
    out->device_id = dev[root];
    out->used_rccl = use_rccl ? 1 : 0;
}

}  // namespace gdx
