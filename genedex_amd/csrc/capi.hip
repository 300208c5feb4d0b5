// capi.hip -- the extern "C" boundary declared in include/gdx.h (and the bench / synthetic-data
// helpers of include/gdx_bench.h).  Exceptions never cross it: they become a gdx_status plus a
// thread-local message.
#include <cstring>
#include <map>
#include <new>
#include <string>

#include "../../include/gdx.h"
#include "../../include/gdx_bench.h"
#include "fastx.hpp"
#include "fm_index.hpp"
#include "kernels.hpp"
#include "synth.hpp"

namespace gdx {

namespace {
struct Arena {
    void *ptr = nullptr;
    size_t bytes = 0;
    int device = -1;
};
struct ArenaKey {
    hipStream_t stream;
    int slot;
    int device;
    bool operator<(const ArenaKey &o) const
    {
        if (stream != o.stream) return stream < o.stream;
        if (slot != o.slot) return slot < o.slot;
        return device < o.device;
    }
};
struct ArenaMap {
    std::map<ArenaKey, Arena> arenas;
    ~ArenaMap()
    {
        for (auto &kv : arenas)
            if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    }
};
}  // namespace

void *stream_scratch(hipStream_t stream, int slot, size_t bytes)
{
    thread_local ArenaMap map;
    int dev = 0;
    GDX_HIP(hipGetDevice(&dev));
    Arena &a = map.arenas[ArenaKey{stream, slot, dev}];
    if (a.bytes < bytes) {
        if (a.ptr) {
            GDX_HIP(hipStreamSynchronize(stream));
            GDX_HIP(hipFree(a.ptr));
            a.ptr = nullptr;
            a.bytes = 0;
        }
        const size_t want = bytes + bytes / 8 + 4096;
        GDX_HIP(hipMalloc(&a.ptr, want));
        a.bytes = want;
    }
    return a.ptr;
}

}  // namespace gdx

struct gdx_index {
    std::unique_ptr<gdx::FmIndex> impl;
};

struct gdx_fastx {
    std::unique_ptr<gdx::FastxReader> impl;
};

namespace {

thread_local std::string g_last_error;

template <class F>
int guarded(F &&f)
{
    try {
        const int rc = f();
        if (rc == GDX_ERR_QUERY_STATUS)
            g_last_error = "at least one query has a non-zero status (symbol outside the alphabet, or a "
                           "non-searchable symbol inside the lookup-table suffix); see out_status";
        else if (rc == GDX_ERR_CAPACITY)
            g_last_error = "output buffer too small; required size reported in out_total";
        return rc;
    } catch (const gdx::Error &e) {
        g_last_error = e.what();
        return e.status;
    } catch (const std::bad_alloc &) {
        g_last_error = "out of host memory";
        return GDX_ERR_DEVICE;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return GDX_ERR_DEVICE;
    }
}

gdx::IndexConfig make_config(const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate,
                             int lookup_depth, int index_width, int device_id)
{
    if (!io_to_dense) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "io_to_dense is null");
    gdx::IndexConfig cfg;
    std::memcpy(cfg.io_to_dense, io_to_dense, 256);
    cfg.sigma = sigma;
    cfg.n_searchable = n_searchable;
    cfg.sa_rate = sa_rate;
    cfg.lookup_depth = lookup_depth;
    cfg.index_width = index_width;
    cfg.device_id = device_id;
    return cfg;
}

const gdx::FmIndex &deref(const gdx_index_t *ix)
{
    if (!ix || !ix->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "index handle is null");
    return *ix->impl;
}

hipStream_t as_stream(void *s) { return static_cast<hipStream_t>(s); }

}  // namespace

extern "C" {

const char *gdx_last_error(void) { return g_last_error.c_str(); }

int gdx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gdx_index_build(const uint8_t *texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                    const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate, int lookup_depth,
                    int index_width, int device_id, gdx_index_t **out)
{
    return guarded([&] {
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        auto cfg = make_config(io_to_dense, sigma, n_searchable, sa_rate, lookup_depth, index_width, device_id);
        auto impl = gdx::FmIndex::construct_index(texts_buf, false, text_offsets, n_texts, cfg);
        *out = new gdx_index{std::move(impl)};
        return (int)GDX_OK;
    });
}

int gdx_index_build_dev(const void *d_texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                        const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate, int lookup_depth,
                        int index_width, int device_id, gdx_index_t **out)
{
    return guarded([&] {
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        auto cfg = make_config(io_to_dense, sigma, n_searchable, sa_rate, lookup_depth, index_width, device_id);
        auto impl = gdx::FmIndex::construct_index(static_cast<const uint8_t *>(d_texts_buf), true, text_offsets,
                                                  n_texts, cfg);
        *out = new gdx_index{std::move(impl)};
        return (int)GDX_OK;
    });
}

int gdx_index_from_parts_ex(int table_kind, int block_bits, const uint64_t *count, const uint64_t *interleaved_blocks,
                            uint64_t n, const uint32_t *sa_samples, uint64_t sa_rate, const uint64_t *border_keys,
                            const uint64_t *border_vals, const uint64_t *sentinel_indices, uint64_t n_texts,
                            const uint8_t *io_to_dense, int sigma, int n_searchable, int lookup_depth,
                            int index_width, int device_id, gdx_index_t **out)
{
    return guarded([&] {
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        auto cfg = make_config(io_to_dense, sigma, n_searchable, sa_rate, lookup_depth, index_width, device_id);
        auto impl = gdx::FmIndex::from_parts(table_kind, block_bits, count, interleaved_blocks, n, sa_samples,
                                             border_keys, border_vals, sentinel_indices, n_texts, cfg);
        *out = new gdx_index{std::move(impl)};
        return (int)GDX_OK;
    });
}

int gdx_index_from_parts(const uint64_t *count, const uint64_t *interleaved_blocks, uint64_t n,
                         const uint32_t *sa_samples, uint64_t sa_rate, const uint64_t *border_keys,
                         const uint64_t *border_vals, const uint64_t *sentinel_indices, uint64_t n_texts,
                         const uint8_t *io_to_dense, int sigma, int n_searchable, int lookup_depth, int index_width,
                         int device_id, gdx_index_t **out)
{
    return gdx_index_from_parts_ex(0, 64, count, interleaved_blocks, n, sa_samples, sa_rate, border_keys, border_vals,
                                   sentinel_indices, n_texts, io_to_dense, sigma, n_searchable, lookup_depth,
                                   index_width, device_id, out);
}

int gdx_index_save(const gdx_index_t *ix, const char *path)
{
    return guarded([&] {
        deref(ix).save(path);
        return (int)GDX_OK;
    });
}

int gdx_index_load(const char *path, int device_id, gdx_index_t **out)
{
    return guarded([&] {
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        auto impl = gdx::FmIndex::load(path, device_id);
        *out = new gdx_index{std::move(impl)};
        return (int)GDX_OK;
    });
}

void gdx_index_free(gdx_index_t *ix)
{
    if (!ix) return;
    (void)guarded([&] {
        if (ix->impl) ix->impl->make_current();
        delete ix;
        return (int)GDX_OK;
    });
}

int gdx_index_info(const gdx_index_t *ix, gdx_index_info_t *out)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        out->total_text_len = f.total_text_len();
        out->num_texts = f.num_texts();
        out->sigma = f.config().sigma;
        out->n_searchable = f.config().n_searchable;
        out->lookup_depth = f.config().lookup_depth;
        out->index_width = f.config().index_width;
        out->sa_rate = f.config().sa_rate;
        out->device_bytes = f.device_bytes();
        out->device_id = f.config().device_id;
        out->table_layout = f.view().layout;
        return (int)GDX_OK;
    });
}

int gdx_index_build_stats(const gdx_index_t *ix, gdx_build_stats_t *out)
{
    return guarded([&] {
        const gdx::BuildStats &s = deref(ix).build_stats();
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        out->sa_initial_order = s.sa_initial_order;
        out->sa_pending_after_sort = s.sa_pending_after_sort;
        out->sa_rounds = s.sa_rounds;
        out->seconds_encode = s.seconds_encode;
        out->seconds_sa = s.seconds_sa;
        out->seconds_bwt = s.seconds_bwt;
        out->seconds_table = s.seconds_table;
        out->seconds_lookup = s.seconds_lookup;
        out->seconds_pairs = s.seconds_pairs;
        return (int)GDX_OK;
    });
}

#define GDX_EXPORT(NAME, CALL)                       \
    return guarded([&] {                             \
        const gdx::FmIndex &f = deref(ix);           \
        CALL;                                        \
        return (int)GDX_OK;                          \
    })

int gdx_index_export_count(const gdx_index_t *ix, uint64_t *count) { GDX_EXPORT(count, f.export_count(count)); }
int gdx_index_export_bwt(const gdx_index_t *ix, uint8_t *bwt) { GDX_EXPORT(bwt, f.export_bwt(bwt)); }
int gdx_index_export_sa_samples(const gdx_index_t *ix, uint32_t *s) { GDX_EXPORT(sa, f.export_sa_samples(s)); }
int gdx_index_export_borders(const gdx_index_t *ix, uint64_t *k, uint64_t *v) { GDX_EXPORT(b, f.export_borders(k, v)); }
int gdx_index_export_sentinel_indices(const gdx_index_t *ix, uint64_t *o) { GDX_EXPORT(s, f.export_sentinel_indices(o)); }
int gdx_index_export_lookup_table(const gdx_index_t *ix, int depth, uint32_t *pairs)
{
    GDX_EXPORT(l, f.export_lookup_table(depth, pairs));
}
int gdx_index_export_condensed_table(const gdx_index_t *ix, uint64_t *blocks, uint16_t *block_offsets,
                                     uint32_t *superblock_offsets)
{
    GDX_EXPORT(t, f.export_condensed_table(blocks, block_offsets, superblock_offsets));
}

int gdx_rank_many(const gdx_index_t *ix, const uint8_t *symbols, const uint64_t *idx, uint64_t m, uint64_t *out)
{
    return guarded([&] { return deref(ix).rank_many(symbols, idx, m, out); });
}

int gdx_symbol_at_many(const gdx_index_t *ix, const uint64_t *idx, uint64_t m, uint8_t *out)
{
    return guarded([&] { return deref(ix).symbol_at_many(idx, m, out); });
}

int gdx_count_many(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                   uint64_t *out_counts, uint8_t *out_status)
{
    return guarded([&] {
        if (!out_counts && nq) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_counts is null");
        return deref(ix).cursors_for_many_queries(qbuf, qoff, nq, nullptr, nullptr, out_counts, out_status);
    });
}

int gdx_cursors_for_many_queries(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                 uint64_t *out_start, uint64_t *out_end, uint8_t *out_status)
{
    return guarded([&] {
        if ((!out_start || !out_end) && nq) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_start / out_end is null");
        return deref(ix).cursors_for_many_queries(qbuf, qoff, nq, out_start, out_end, nullptr, out_status);
    });
}

int gdx_locate_many(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                    uint64_t *out_hit_offsets, gdx_hit_t *hits, uint64_t hits_capacity, uint64_t *out_total,
                    uint8_t *out_status)
{
    return guarded([&] {
        return deref(ix).locate_many(qbuf, qoff, nq, out_hit_offsets, hits, hits_capacity, out_total, out_status);
    });
}

int gdx_cursor_empty(const gdx_index_t *ix, uint64_t *start, uint64_t *end)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (!start || !end) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
        *start = 0;  // lib.rs:202-210
        *end = f.total_text_len();
        return (int)GDX_OK;
    });
}

int gdx_cursor_extend_front_many(const gdx_index_t *ix, uint64_t *start, uint64_t *end, const uint8_t *io_symbols,
                                 uint64_t m, uint8_t *out_status)
{
    return guarded([&] { return deref(ix).cursor_extend_front_many(start, end, io_symbols, m, out_status); });
}

int gdx_cursor_locate_many(const gdx_index_t *ix, const uint64_t *start, const uint64_t *end, uint64_t m,
                           uint64_t *out_hit_offsets, gdx_hit_t *hits, uint64_t hits_capacity, uint64_t *out_total)
{
    return guarded(
        [&] { return deref(ix).cursor_locate_many(start, end, m, out_hit_offsets, hits, hits_capacity, out_total); });
}

// ---- device-resident entry points ---------------------------------------------------------------------

int gdx_cursors_for_many_queries_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                     void *d_out_start, void *d_out_end, void *d_out_status, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if ((reinterpret_cast<uintptr_t>(d_qbuf) & 7u) != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_qbuf must be 8-byte aligned");
        gdx::launch_search(f.view(), static_cast<const uint8_t *>(d_qbuf), static_cast<const uint64_t *>(d_qoff), nq,
                           static_cast<uint32_t *>(d_out_start), static_cast<uint32_t *>(d_out_end), nullptr,
                           static_cast<uint8_t *>(d_out_status), as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_cursors_for_many_queries_hint_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                          void *d_out_start, void *d_out_end, void *d_out_status, void *d_hint,
                                          void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if ((reinterpret_cast<uintptr_t>(d_qbuf) & 7u) != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_qbuf must be 8-byte aligned");
        if (!d_hint || (reinterpret_cast<uintptr_t>(d_hint) & 7u) != 0)
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_hint must be a non-null 8-byte aligned device pointer");
        gdx::launch_search(f.view(), static_cast<const uint8_t *>(d_qbuf), static_cast<const uint64_t *>(d_qoff), nq,
                           static_cast<uint32_t *>(d_out_start), static_cast<uint32_t *>(d_out_end), nullptr,
                           static_cast<uint8_t *>(d_out_status), as_stream(stream), nullptr,
                           static_cast<uint2 *>(d_hint));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_count_many_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq, void *d_out_counts,
                       void *d_out_status, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if ((reinterpret_cast<uintptr_t>(d_qbuf) & 7u) != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_qbuf must be 8-byte aligned");
        gdx::launch_search(f.view(), static_cast<const uint8_t *>(d_qbuf), static_cast<const uint64_t *>(d_qoff), nq,
                           nullptr, nullptr, static_cast<uint32_t *>(d_out_counts),
                           static_cast<uint8_t *>(d_out_status), as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_cursor_extend_front_many_dev(const gdx_index_t *ix, void *d_start, void *d_end, const void *d_io_symbols,
                                     uint64_t m, void *d_out_status, void *stream)
{
    return guarded([&] {
        gdx::launch_extend_front(deref(ix).view(), static_cast<uint32_t *>(d_start), static_cast<uint32_t *>(d_end),
                                 static_cast<const uint8_t *>(d_io_symbols), m, static_cast<uint8_t *>(d_out_status),
                                 as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_hit_offsets_dev(const gdx_index_t *ix, const void *d_start, const void *d_end, uint64_t m,
                        void *d_hit_offsets, void *stream)
{
    return guarded([&] {
        (void)deref(ix);
        const size_t tb = gdx::hit_offsets_temp_bytes(m);
        void *temp = gdx::stream_scratch(as_stream(stream), 8, tb ? tb : 1);
        gdx::launch_hit_offsets(static_cast<const uint32_t *>(d_start), static_cast<const uint32_t *>(d_end), m,
                                static_cast<uint64_t *>(d_hit_offsets), temp, tb, as_stream(stream));
        return (int)GDX_OK;
    });
}

uint64_t gdx_locate_workspace_bytes(uint64_t total_hits) { return gdx::locate_workspace_bytes(total_hits); }

int gdx_locate_intervals_dev(const gdx_index_t *ix, const void *d_start, const void *d_end, uint64_t m,
                             const void *d_hit_offsets, uint64_t total_hits, void *d_hits, void *d_workspace,
                             void *stream)
{
    return guarded([&] {
        gdx::launch_locate(deref(ix).view(), static_cast<const uint32_t *>(d_start),
                           static_cast<const uint32_t *>(d_end), m, static_cast<const uint64_t *>(d_hit_offsets),
                           total_hits, d_hits, false, d_workspace, as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_locate_intervals_hint_dev(const gdx_index_t *ix, const void *d_start, const void *d_end, uint64_t m,
                                  const void *d_hit_offsets, uint64_t total_hits, void *d_hits, void *d_workspace,
                                  const void *d_hint, void *stream)
{
    return guarded([&] {
        gdx::launch_locate(deref(ix).view(), static_cast<const uint32_t *>(d_start),
                           static_cast<const uint32_t *>(d_end), m, static_cast<const uint64_t *>(d_hit_offsets),
                           total_hits, d_hits, false, d_workspace, as_stream(stream), nullptr,
                           static_cast<const uint2 *>(d_hint));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_rank_many_dev(const gdx_index_t *ix, const void *d_symbols, const void *d_idx, uint64_t m, void *d_out,
                      void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        uint32_t *d_err = static_cast<uint32_t *>(gdx::stream_scratch(as_stream(stream), 9, sizeof(uint32_t)));
        GDX_HIP(hipMemsetAsync(d_err, 0, sizeof(uint32_t), as_stream(stream)));
        gdx::launch_rank_many(f.view(), static_cast<const uint8_t *>(d_symbols), static_cast<const uint32_t *>(d_idx),
                              m, static_cast<uint32_t *>(d_out), d_err, as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

// ---- include/gdx_bench.h: synthetic workloads and roofline micro-benchmarks ------------------------------

int gdx_synth_text_dev(void *d_out, uint64_t n, uint64_t seed, uint32_t n_per_million, void *stream)
{
    return guarded([&] {
        gdx::launch_synth_text(static_cast<uint8_t *>(d_out), n, seed, n_per_million, as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_synth_queries_dev(const void *d_io_text, const void *d_text_offsets, uint64_t n_texts, uint64_t nq,
                          uint32_t len_min, uint32_t len_max, uint32_t sampled_per_million, uint64_t seed,
                          void *d_qoff, void *d_qbuf, uint64_t qbuf_capacity, uint64_t *out_total_bytes, void *stream)
{
    return guarded([&] {
        gdx::synth_queries(static_cast<const uint8_t *>(d_io_text), static_cast<const uint64_t *>(d_text_offsets),
                           n_texts, nq, len_min, len_max, sampled_per_million, seed, static_cast<uint64_t *>(d_qoff),
                           static_cast<uint8_t *>(d_qbuf), qbuf_capacity, out_total_bytes, as_stream(stream));
        return (int)GDX_OK;
    });
}

int gdx_debug_set_search_variant(int variant)
{
    if (variant < -1 || variant > 2) return GDX_ERR_INVALID_ARGUMENT;
    gdx::set_search_variant(variant);
    return GDX_OK;
}

int gdx_bench_stream_copy(void *d_dst, const void *d_src, uint64_t bytes, void *stream)
{
    return guarded([&] {
        gdx::launch_stream_copy(d_dst, d_src, bytes, as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_bench_stream_read(const void *d_src, uint64_t bytes, void *d_sink, void *stream)
{
    return guarded([&] {
        gdx::launch_stream_read(d_src, bytes, static_cast<uint32_t *>(d_sink), as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_bench_random_gather(const void *d_src, uint64_t n_lines, uint32_t line_bytes, uint64_t n_accesses,
                            uint64_t seed, uint32_t mode, void *d_sink, void *stream)
{
    return guarded([&] {
        gdx::launch_random_gather(d_src, n_lines, line_bytes, n_accesses, seed, mode, static_cast<uint32_t *>(d_sink),
                                  as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_search_step_stats_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                              void *d_steps, void *stream)
{
    return guarded([&] {
        gdx::launch_search(deref(ix).view(), static_cast<const uint8_t *>(d_qbuf),
                           static_cast<const uint64_t *>(d_qoff), nq, nullptr, nullptr, nullptr, nullptr,
                           as_stream(stream), static_cast<unsigned long long *>(d_steps));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_index_aux_info(const gdx_index_t *ix, uint32_t out[4])
{
    return guarded([&] {
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        const gdx::IndexView &v = deref(ix).view();
        out[0] = v.pair_lines != nullptr;
        out[1] = v.jump ? v.jump_bytes : 0u;
        out[2] = v.top ? v.top_depth : 0u;
        out[3] = 0;
        return (int)GDX_OK;
    });
}

int gdx_locate_step_stats_dev(const gdx_index_t *ix, const void *d_start, const void *d_end, uint64_t m,
                              const void *d_hit_offsets, uint64_t total_hits, void *d_hits, void *d_workspace,
                              void *d_steps, void *stream)
{
    return guarded([&] {
        gdx::launch_locate(deref(ix).view(), static_cast<const uint32_t *>(d_start),
                           static_cast<const uint32_t *>(d_end), m, static_cast<const uint64_t *>(d_hit_offsets),
                           total_hits, d_hits, false, d_workspace, as_stream(stream),
                           static_cast<unsigned long long *>(d_steps));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

// ---- FASTA / FASTQ ingestion (host only) ------------------------------------------------------------------

int gdx_fastx_open(const char *path, gdx_fastx_t **out)
{
    return guarded([&] {
        if (!path || !out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "path / out is null");
        auto r = std::make_unique<gdx_fastx>();
        r->impl = std::make_unique<gdx::FastxReader>(path);
        *out = r.release();
        return (int)GDX_OK;
    });
}

int gdx_fastx_next_batch(gdx_fastx_t *reader, uint8_t *qbuf, uint64_t qbuf_capacity, uint64_t *qoff,
                         uint64_t max_records, uint64_t *n_out)
{
    return guarded([&] {
        if (!reader || !reader->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "reader handle is null");
        if (!qoff || !n_out || (!qbuf && qbuf_capacity)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "output pointer is null");
        *n_out = reader->impl->next_batch(qbuf, qbuf_capacity, qoff, max_records);
        return (int)GDX_OK;
    });
}

void gdx_fastx_close(gdx_fastx_t *reader) { delete reader; }

}  // extern "C"
