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
    std::lock_guard<std::mutex> serial(m.call_mutex);
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
