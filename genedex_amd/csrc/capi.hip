// capi.hip -- the extern "C" boundary declared in include/gdx.h (and the bench / synthetic-data
// helpers of include/gdx_bench.h).  Exceptions never cross it: they become a gdx_status plus a
// thread-local message.
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gdx.h"
#include "../../include/gdx_experimental.h"
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
    std::unique_ptr<gdx::WideIndex> wide;  // index storage beyond 32 bits (wide.hip): exactly one of the two is set
};

struct gdx_multi {
    std::unique_ptr<gdx::Multi> impl;
};

struct gdx_parts {
    std::unique_ptr<gdx::Parts> impl;
};

struct gdx_fastx {
    std::unique_ptr<gdx::FastxMappedReader> mapped;  // a regular file: parsed by several threads
    std::unique_ptr<gdx::FastxReader> impl;          // anything else, or GDX_FASTX_THREADS=0
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

gdx::BuildOptions make_build_options(const gdx_build_options_t *o);

gdx::IndexConfig make_config(const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate,
                             int lookup_depth, int index_width, int device_id,
                             const gdx_build_options_t *opts = nullptr)
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
    cfg.build = make_build_options(opts);
    return cfg;
}

const gdx::FmIndex &deref(const gdx_index_t *ix)
{
    if (ix && ix->wide)
        gdx::fail(GDX_ERR_UNSUPPORTED, "this call is not available on an index with 64-bit storage (count, cursors_for_many_queries "
                                       "and locate on host pointers are)");
    if (!ix || !ix->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "index handle is null");
    return *ix->impl;
}

// index_width 64 and a collection that does not fit 32-bit rows (or gdx_debug_force_wide): the 64-bit engine of wide.hip
std::atomic<int> g_force_wide{0};
bool wants_wide(const uint64_t *text_offsets, uint64_t n_texts, int index_width)
{
    if (index_width != 64 || !text_offsets || n_texts == 0) return false;
    const uint64_t n = text_offsets[n_texts] - text_offsets[0] + n_texts;
    return n > 0xffffffffull || g_force_wide.load() != 0;
}

hipStream_t as_stream(void *s) { return static_cast<hipStream_t>(s); }

// makes the handle's device current for the duration of a call and restores the caller's afterwards
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev)
    {
        GDX_HIP(hipGetDevice(&prev));
        if (prev != dev) {
            GDX_HIP(hipSetDevice(dev));
            changed = true;
        }
    }
    ~DeviceGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

gdx::BuildOptions make_build_options(const gdx_build_options_t *o)
{
    gdx::BuildOptions b;
    if (!o) return b;
    // struct_size lets the struct grow: a caller compiled against an older header passes a shorter struct, whose
    // missing tail takes the defaults; a longer one (newer caller) is read up to what this library knows
    gdx_build_options_t full;
    gdx_build_options_init(&full);
    if (o->struct_size < 2 * sizeof(uint32_t))
        gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_build_options_t.struct_size is %u (use gdx_build_options_init)", o->struct_size);
    std::memcpy(&full, o, o->struct_size < sizeof(full) ? o->struct_size : sizeof(full));
    o = &full;
    if (o->pair_lines < -1 || o->pair_lines > 1) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "pair_lines must be -1, 0 or 1");
    if (o->jump_entry_bytes != -1 && o->jump_entry_bytes != 0 && o->jump_entry_bytes != 8 && o->jump_entry_bytes != 16 &&
        o->jump_entry_bytes != 32)
        gdx::fail(GDX_ERR_INVALID_ARGUMENT, "jump_entry_bytes must be -1, 0, 8, 16 or 32");
    if (o->top_table_depth < -1 || o->top_table_depth > 16)
        gdx::fail(GDX_ERR_INVALID_ARGUMENT, "top_table_depth must be in -1..=16");
    b.pair_lines = o->pair_lines;
    b.jump_bytes = o->jump_entry_bytes;
    b.top_depth = o->top_table_depth;
    b.aux_budget_bytes = o->aux_budget_bytes;
    if (o->full_suffix_array < -1 || o->full_suffix_array > 1 || o->text_units < -1 || o->text_units > 1)
        gdx::fail(GDX_ERR_INVALID_ARGUMENT, "full_suffix_array / text_units must be -1, 0 or 1");
    b.full_sa = o->full_suffix_array;
    b.text_units = o->text_units;
    if (o->seed_symbols < -1 || o->seed_symbols > 24 || (o->seed_symbols > 1 && o->seed_symbols < 8))
        gdx::fail(GDX_ERR_INVALID_ARGUMENT, "seed_symbols must be -1, 0, 1 (automatic) or 8..24");
    if (o->seed_load_percent != 0 && (o->seed_load_percent < 20 || o->seed_load_percent > 100))
        gdx::fail(GDX_ERR_INVALID_ARGUMENT, "seed_load_percent must be 0 (default) or 20..100");
    b.seed_symbols = o->seed_symbols;
    b.seed_load_percent = o->seed_load_percent;
    if (o->reference_table_layout < -1 || o->reference_table_layout > 4)
        gdx::fail(GDX_ERR_INVALID_ARGUMENT, "reference_table_layout must be -1, 0 or 1..4");
    b.table_layout = o->reference_table_layout;
    if (o->inverse_suffix_array < -1 || o->inverse_suffix_array > 1) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "inverse_suffix_array must be -1, 0 or 1");
    b.inverse_sa = o->inverse_suffix_array;
    return b;
}

gdx::QueryOptions parse_query_options(const gdx_query_options_t *opts)
{
    gdx::QueryOptions q;
    gdx_query_options_t full;
    if (opts) {
        // (struct_size: as in make_build_options -- a shorter struct of an older caller keeps defaults for the rest)
        gdx_query_options_init(&full);
        if (opts->struct_size < 2 * sizeof(uint32_t))
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_options_t.struct_size is too small (use gdx_query_options_init)");
        std::memcpy(&full, opts, opts->struct_size < sizeof(full) ? opts->struct_size : sizeof(full));
        opts = &full;
        if (opts->search_kernel < -1 || opts->search_kernel > 2 ||
            (opts->search_lanes != 0 && opts->search_lanes != 4 && opts->search_lanes != 8) || opts->load_policy < -1 ||
            opts->load_policy > 3 || opts->length_schedule < -1 || opts->length_schedule > 1 ||
            opts->locate_kernel < -1 || opts->locate_kernel > 2)
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_options_t: field out of range");
        q.search_variant = opts->search_kernel;
        q.search_lanes = opts->search_lanes;
        q.load_policy = opts->load_policy;
        q.length_schedule = opts->length_schedule;
        q.locate_variant = opts->locate_kernel;
        if (opts->locate_jump_walk < -1 || opts->locate_jump_walk > 1)
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_options_t: field out of range");
        q.locate_jump_walk = opts->locate_jump_walk;
        if (opts->search_defer_after < -1 || opts->search_defer_after > 1000)
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_options_t: field out of range");
        q.search_defer_after = opts->search_defer_after;
        if (opts->search_fast < -1 || opts->search_fast > 2) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_options_t: field out of range");
        q.search_fast = opts->search_fast;
        if (opts->search_exact < -1 || opts->search_exact > 1) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_options_t: field out of range");
        q.search_exact = opts->search_exact;
        q.max_hits_per_query = opts->max_hits_per_query;
        if (opts->search_seed < -1 || opts->search_seed > 1) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_options_t: field out of range");
        q.search_seed = opts->search_seed;
    }
    return q;
}

}  // namespace

extern "C" {

const char *gdx_last_error(void) { return g_last_error.c_str(); }

int gdx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void gdx_build_options_init(gdx_build_options_t *opts)
{
    if (!opts) return;
    opts->struct_size = sizeof(gdx_build_options_t);
    opts->pair_lines = -1;
    opts->jump_entry_bytes = -1;
    opts->top_table_depth = -1;
    opts->aux_budget_bytes = 0;
    opts->full_suffix_array = -1;
    opts->text_units = -1;
    opts->seed_symbols = -1;
    opts->seed_load_percent = 0;
    opts->inverse_suffix_array = -1;
    opts->reference_table_layout = -1;
}

void gdx_query_options_init(gdx_query_options_t *opts)
{
    if (!opts) return;
    opts->struct_size = sizeof(gdx_query_options_t);
    opts->search_kernel = -1;
    opts->search_lanes = 0;
    opts->load_policy = -1;
    opts->length_schedule = -1;
    opts->locate_kernel = -1;
    opts->locate_jump_walk = -1;
    opts->search_defer_after = -1;
    opts->search_fast = -1;
    opts->search_exact = -1;
    opts->max_hits_per_query = 0;
    opts->search_seed = -1;
}

int gdx_index_build(const uint8_t *texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                    const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate, int lookup_depth,
                    int index_width, int device_id, gdx_index_t **out)
{
    return gdx_index_build_ex(texts_buf, text_offsets, n_texts, io_to_dense, sigma, n_searchable, sa_rate, lookup_depth,
                              index_width, device_id, nullptr, out);
}

int gdx_index_build_ex(const uint8_t *texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                       const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate, int lookup_depth,
                       int index_width, int device_id, const gdx_build_options_t *opts, gdx_index_t **out)
{
    return guarded([&] {
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        auto cfg = make_config(io_to_dense, sigma, n_searchable, sa_rate, lookup_depth, index_width, device_id, opts);
        if (wants_wide(text_offsets, n_texts, index_width)) {
            DeviceGuard guard(device_id);
            *out = new gdx_index{nullptr, gdx::WideIndex::construct_index(texts_buf, false, text_offsets, n_texts, cfg)};
            return (int)GDX_OK;
        }
        auto impl = gdx::FmIndex::construct_index(texts_buf, false, text_offsets, n_texts, cfg);
        *out = new gdx_index{std::move(impl), nullptr};
        return (int)GDX_OK;
    });
}

int gdx_index_build_dev(const void *d_texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                        const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate, int lookup_depth,
                        int index_width, int device_id, gdx_index_t **out)
{
    return gdx_index_build_dev_ex(d_texts_buf, text_offsets, n_texts, io_to_dense, sigma, n_searchable, sa_rate,
                                  lookup_depth, index_width, device_id, nullptr, out);
}

int gdx_index_build_dev_ex(const void *d_texts_buf, const uint64_t *text_offsets, uint64_t n_texts,
                           const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate, int lookup_depth,
                           int index_width, int device_id, const gdx_build_options_t *opts, gdx_index_t **out)
{
    return guarded([&] {
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        auto cfg = make_config(io_to_dense, sigma, n_searchable, sa_rate, lookup_depth, index_width, device_id, opts);
        if (wants_wide(text_offsets, n_texts, index_width)) {
            DeviceGuard guard(device_id);
            *out = new gdx_index{nullptr, gdx::WideIndex::construct_index(static_cast<const uint8_t *>(d_texts_buf), true,
                                                                          text_offsets, n_texts, cfg)};
            return (int)GDX_OK;
        }
        auto impl = gdx::FmIndex::construct_index(static_cast<const uint8_t *>(d_texts_buf), true, text_offsets,
                                                  n_texts, cfg);
        *out = new gdx_index{std::move(impl), nullptr};
        return (int)GDX_OK;
    });
}

int gdx_index_from_parts_ex(int table_kind, int block_bits, const uint64_t *count, const uint64_t *interleaved_blocks,
                            uint64_t n, const uint32_t *sa_samples, uint64_t sa_rate, const uint64_t *border_keys,
                            const uint64_t *border_vals, const uint64_t *sentinel_indices, uint64_t n_texts,
                            const uint8_t *io_to_dense, int sigma, int n_searchable, int lookup_depth,
                            int index_width, int device_id, gdx_index_t **out)
{
    return gdx_index_from_parts_ex2(table_kind, block_bits, count, interleaved_blocks, n, sa_samples, sa_rate, border_keys,
                                    border_vals, sentinel_indices, n_texts, io_to_dense, sigma, n_searchable, lookup_depth,
                                    index_width, device_id, nullptr, out);
}

int gdx_index_from_parts_ex2(int table_kind, int block_bits, const uint64_t *count, const uint64_t *interleaved_blocks,
                             uint64_t n, const uint32_t *sa_samples, uint64_t sa_rate, const uint64_t *border_keys,
                             const uint64_t *border_vals, const uint64_t *sentinel_indices, uint64_t n_texts,
                             const uint8_t *io_to_dense, int sigma, int n_searchable, int lookup_depth,
                             int index_width, int device_id, const gdx_build_options_t *opts, gdx_index_t **out)
{
    return guarded([&] {
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        auto cfg = make_config(io_to_dense, sigma, n_searchable, sa_rate, lookup_depth, index_width, device_id, opts);
        auto impl = gdx::FmIndex::from_parts(table_kind, block_bits, count, interleaved_blocks, n, sa_samples,
                                             border_keys, border_vals, sentinel_indices, n_texts, cfg);
        *out = new gdx_index{std::move(impl), nullptr};
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
    return gdx_index_load_ex(path, device_id, nullptr, out);
}

int gdx_index_load_ex(const char *path, int device_id, const gdx_build_options_t *opts, gdx_index_t **out)
{
    return guarded([&] {
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        auto impl = gdx::FmIndex::load(path, device_id, make_build_options(opts));
        *out = new gdx_index{std::move(impl), nullptr};
        return (int)GDX_OK;
    });
}

void gdx_index_free(gdx_index_t *ix)
{
    if (!ix) return;
    (void)guarded([&] {
        if (ix->impl) {
            DeviceGuard guard(ix->impl->config().device_id);  // the caller's current device is left as it was
            ix->impl.reset();
        }
        if (ix->wide) {
            DeviceGuard guard(ix->wide->config().device_id);
            ix->wide.reset();
        }
        delete ix;
        return (int)GDX_OK;
    });
}

int gdx_index_aux(const gdx_index_t *ix, gdx_index_aux_t *out)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        const gdx::IndexView &v = f.view();
        const gdx::AuxReport &r = f.aux_report();
        out->pair_lines = v.pair_lines != nullptr;
        out->jump_entry_bytes = v.jump ? static_cast<int32_t>(v.jump_bytes) : 0;
        out->top_table_depth = v.top ? static_cast<int32_t>(v.top_depth) : 0;
        out->wanted_jump_entry_bytes = static_cast<int32_t>(r.wanted_jump_bytes);
        out->wanted_top_table_depth = static_cast<int32_t>(r.wanted_top_depth);
        out->wide_permille = static_cast<int32_t>(r.wide_fraction * 1000.0 + 0.5);
        out->aux_bytes = r.aux_bytes;
        out->aux_budget_bytes = r.budget_bytes;
        return (int)GDX_OK;
    });
}

int gdx_index_set_query_options(gdx_index_t *ix, const gdx_query_options_t *opts)
{
    return guarded([&] {
        if (!ix || !ix->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "index handle is null");
        ix->impl->set_query_options(parse_query_options(opts));
        return (int)GDX_OK;
    });
}

int gdx_multi_set_query_options(gdx_multi_t *m, const gdx_query_options_t *opts)
{
    return guarded([&] {
        if (!m || !m->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "multi handle is null");
        const gdx::QueryOptions q = parse_query_options(opts);
        for (auto &rep : m->impl->replicas) rep->set_query_options(q);
        return (int)GDX_OK;
    });
}

int gdx_index_get_query_options(const gdx_index_t *ix, gdx_query_options_t *out)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        const gdx::QueryOptions q = f.query_options();
        out->struct_size = sizeof(gdx_query_options_t);
        out->search_kernel = q.search_variant;
        out->search_lanes = q.search_lanes;
        out->load_policy = q.load_policy;
        out->length_schedule = q.length_schedule;
        out->locate_kernel = q.locate_variant;
        out->locate_jump_walk = q.locate_jump_walk;
        out->search_defer_after = q.search_defer_after;
        out->search_fast = q.search_fast;
        out->search_exact = q.search_exact;
        out->max_hits_per_query = q.max_hits_per_query;
        out->search_seed = q.search_seed;
        return (int)GDX_OK;
    });
}

int gdx_index_seed_info(const gdx_index_t *ix, uint64_t out[8])
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        const gdx::AuxReport &r = f.aux_report();
        out[0] = f.view().seed ? r.seed_k : 0;
        out[1] = r.seed_buckets;
        out[2] = r.seed_single;
        out[3] = r.seed_multi;
        out[4] = r.seed_overflowed;
        out[5] = r.seed_max_disp;
        out[6] = r.seed_bytes;
        out[7] = f.view().seed_tag_bits;
        return (int)GDX_OK;
    });
}

int gdx_index_seed_records(const gdx_index_t *ix, uint64_t out[4])
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        const gdx::AuxReport &r = f.aux_report();
        out[0] = r.seed_pair_records;
        out[1] = r.seed_quad_records;
        out[2] = r.seed_pair_records * 32ull + r.seed_quad_records * 64ull;
        out[3] = 0;
        return (int)GDX_OK;
    });
}

int gdx_index_rebuild_aux(gdx_index_t *ix, const gdx_build_options_t *opts)
{
    return guarded([&] {
        if (!ix || !ix->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "index handle is null");
        DeviceGuard guard(ix->impl->config().device_id);
        ix->impl->rebuild_aux(make_build_options(opts));
        return (int)GDX_OK;
    });
}

int gdx_index_info(const gdx_index_t *ix, gdx_index_info_t *out)
{
    return guarded([&] {
        if (ix && ix->wide) {
            if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
            const gdx::WideIndex &w = *ix->wide;
            out->total_text_len = w.total_text_len();
            out->num_texts = w.num_texts();
            out->sigma = w.config().sigma;
            out->n_searchable = w.config().n_searchable;
            out->lookup_depth = w.config().lookup_depth;
            out->index_width = 64;
            out->sa_rate = w.config().sa_rate;
            out->device_bytes = w.device_bytes();
            out->device_id = w.config().device_id;
            out->table_layout = 0;
            return (int)GDX_OK;
        }
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
int gdx_index_export_bwt(const gdx_index_t *ix, uint8_t *bwt)
{
    if (ix && ix->wide)
        return guarded([&] {
            if (!bwt) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "bwt is null");
            DeviceGuard guard(ix->wide->config().device_id);
            ix->wide->export_bwt(bwt);
            return (int)GDX_OK;
        });
    GDX_EXPORT(bwt, f.export_bwt(bwt));
}
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

int gdx_index_export_reference_table(const gdx_index_t *ix, uint64_t *interleaved_blocks, uint64_t capacity_words,
                                     uint64_t *out_n_words, uint32_t *interleaved_superblock_offsets, uint64_t capacity_offsets,
                                     uint64_t *out_n_offsets)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (!out_n_words || !out_n_offsets) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_index_export_reference_table: null argument");
        f.export_reference_table(interleaved_blocks, capacity_words, out_n_words, interleaved_superblock_offsets, capacity_offsets,
                                 out_n_offsets);
        return (int)GDX_OK;
    });
}

int gdx_rank_many(const gdx_index_t *ix, const uint8_t *symbols, const uint64_t *idx, uint64_t m, uint64_t *out)
{
    return guarded([&] {
        if (ix && ix->wide) {
            if (m != 0 && !symbols) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
            DeviceGuard guard(ix->wide->config().device_id);
            return ix->wide->rank_or_symbol_many(symbols, idx, m, out);
        }
        return deref(ix).rank_many(symbols, idx, m, out);
    });
}

int gdx_symbol_at_many(const gdx_index_t *ix, const uint64_t *idx, uint64_t m, uint8_t *out)
{
    return guarded([&] {
        if (ix && ix->wide) {
            if (m != 0 && !out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
            DeviceGuard guard(ix->wide->config().device_id);
            std::vector<uint64_t> wide_out(m);
            const int rc = ix->wide->rank_or_symbol_many(nullptr, idx, m, wide_out.data());
            for (uint64_t i = 0; i < m; i++) out[i] = static_cast<uint8_t>(wide_out[i]);
            return rc;
        }
        return deref(ix).symbol_at_many(idx, m, out);
    });
}

int gdx_count_many(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                   uint64_t *out_counts, uint8_t *out_status)
{
    return guarded([&] {
        if (!out_counts && nq) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_counts is null");
        if (ix && ix->wide) {
            DeviceGuard guard(ix->wide->config().device_id);
            return ix->wide->cursors_for_many_queries(qbuf, qoff, nq, nullptr, nullptr, out_counts, out_status);
        }
        return deref(ix).cursors_for_many_queries(qbuf, qoff, nq, nullptr, nullptr, out_counts, out_status);
    });
}

int gdx_cursors_for_many_queries(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                 uint64_t *out_start, uint64_t *out_end, uint8_t *out_status)
{
    return guarded([&] {
        if ((!out_start || !out_end) && nq) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_start / out_end is null");
        if (ix && ix->wide) {
            DeviceGuard guard(ix->wide->config().device_id);
            return ix->wide->cursors_for_many_queries(qbuf, qoff, nq, out_start, out_end, nullptr, out_status);
        }
        return deref(ix).cursors_for_many_queries(qbuf, qoff, nq, out_start, out_end, nullptr, out_status);
    });
}

int gdx_locate_many(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                    uint64_t *out_hit_offsets, gdx_hit_t *hits, uint64_t hits_capacity, uint64_t *out_total,
                    uint8_t *out_status)
{
    return guarded([&] {
        if (ix && ix->wide) {
            DeviceGuard guard(ix->wide->config().device_id);
            return ix->wide->locate_many(qbuf, qoff, nq, out_hit_offsets, hits, hits ? hits_capacity : 0, out_total, out_status);
        }
        return deref(ix).locate_many(qbuf, qoff, nq, out_hit_offsets, hits, hits_capacity, out_total, out_status);
    });
}

int gdx_locate_many_alloc(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                          uint64_t *out_hit_offsets, gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status)
{
    return guarded([&] {
        if (ix && ix->wide) {
            DeviceGuard guard(ix->wide->config().device_id);
            return ix->wide->locate_many_alloc(qbuf, qoff, nq, out_hit_offsets, out_hits, out_total, out_status);
        }
        return deref(ix).locate_many_alloc(qbuf, qoff, nq, out_hit_offsets, out_hits, out_total, out_status);
    });
}

void gdx_free_hits(gdx_hit_t *hits) { gdx::recycle_hits(hits); }

int gdx_cursor_empty(const gdx_index_t *ix, uint64_t *start, uint64_t *end)
{
    return guarded([&] {
        if (!start || !end) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
        if (ix && ix->wide) {
            *start = 0;
            *end = ix->wide->total_text_len();
            return (int)GDX_OK;
        }
        const gdx::FmIndex &f = deref(ix);
        *start = 0;  // lib.rs:202-210
        *end = f.total_text_len();
        return (int)GDX_OK;
    });
}

int gdx_cursor_extend_front_many(const gdx_index_t *ix, uint64_t *start, uint64_t *end, const uint8_t *io_symbols,
                                 uint64_t m, uint8_t *out_status)
{
    return guarded([&] {
        if (ix && ix->wide) {
            DeviceGuard guard(ix->wide->config().device_id);
            return ix->wide->cursor_extend_front_many(start, end, io_symbols, m, out_status);
        }
        return deref(ix).cursor_extend_front_many(start, end, io_symbols, m, out_status);
    });
}

int gdx_cursor_locate_many(const gdx_index_t *ix, const uint64_t *start, const uint64_t *end, uint64_t m,
                           uint64_t *out_hit_offsets, gdx_hit_t *hits, uint64_t hits_capacity, uint64_t *out_total)
{
    return guarded([&] {
        if (ix && ix->wide) {
            DeviceGuard guard(ix->wide->config().device_id);
            return ix->wide->cursor_locate_many(start, end, m, out_hit_offsets, hits, hits_capacity, out_total);
        }
        return deref(ix).cursor_locate_many(start, end, m, out_hit_offsets, hits, hits_capacity, out_total);
    });
}

// ---- device-resident entry points ---------------------------------------------------------------------

int gdx_cursors_for_many_queries_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                     void *d_out_start, void *d_out_end, void *d_out_status, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if ((reinterpret_cast<uintptr_t>(d_qbuf) & 7u) != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_qbuf must be 8-byte aligned");
        DeviceGuard guard(f.config().device_id);
        gdx::launch_search(f.view(), static_cast<const uint8_t *>(d_qbuf), static_cast<const uint64_t *>(d_qoff), nq,
                           static_cast<uint32_t *>(d_out_start), static_cast<uint32_t *>(d_out_end), nullptr,
                           static_cast<uint8_t *>(d_out_status), as_stream(stream), nullptr, nullptr, f.query_options());
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
        DeviceGuard guard(f.config().device_id);
        gdx::launch_search(f.view(), static_cast<const uint8_t *>(d_qbuf), static_cast<const uint64_t *>(d_qoff), nq,
                           static_cast<uint32_t *>(d_out_start), static_cast<uint32_t *>(d_out_end), nullptr,
                           static_cast<uint8_t *>(d_out_status), as_stream(stream), nullptr,
                           static_cast<uint2 *>(d_hint), f.query_options());
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
        DeviceGuard guard(f.config().device_id);
        gdx::SearchCall c;  // counts only: the search may end with the lazy tail (mode 1)
        c.d_qbuf = static_cast<const uint8_t *>(d_qbuf);
        c.d_qbeg = static_cast<const uint64_t *>(d_qoff);
        c.d_qend = c.d_qbeg + 1;
        c.nq = nq;
        c.d_count = static_cast<uint32_t *>(d_out_counts);
        c.d_status = static_cast<uint8_t *>(d_out_status);
        c.mode = 1;
        gdx::launch_search_call(f.view(), c, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_cursor_extend_front_many_dev(const gdx_index_t *ix, void *d_start, void *d_end, const void *d_io_symbols,
                                     uint64_t m, void *d_out_status, void *stream)
{
    return guarded([&] {
        DeviceGuard guard(deref(ix).config().device_id);
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
        DeviceGuard guard(deref(ix).config().device_id);
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
        DeviceGuard guard(deref(ix).config().device_id);
        gdx::launch_locate(deref(ix).view(), static_cast<const uint32_t *>(d_start),
                           static_cast<const uint32_t *>(d_end), m, static_cast<const uint64_t *>(d_hit_offsets),
                           total_hits, d_hits, false, d_workspace, as_stream(stream), nullptr, nullptr,
                           deref(ix).query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_locate_intervals_hint_dev(const gdx_index_t *ix, const void *d_start, const void *d_end, uint64_t m,
                                  const void *d_hit_offsets, uint64_t total_hits, void *d_hits, void *d_workspace,
                                  const void *d_hint, void *stream)
{
    return guarded([&] {
        DeviceGuard guard(deref(ix).config().device_id);
        gdx::launch_locate(deref(ix).view(), static_cast<const uint32_t *>(d_start),
                           static_cast<const uint32_t *>(d_end), m, static_cast<const uint64_t *>(d_hit_offsets),
                           total_hits, d_hits, false, d_workspace, as_stream(stream), nullptr,
                           static_cast<const uint2 *>(d_hint), deref(ix).query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

namespace {
__global__ __launch_bounds__(256) void unpack_records_kernel(const uint4 *__restrict__ rec, uint64_t nq,
                                                             uint32_t *__restrict__ counts, uint8_t *__restrict__ status)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; q < nq; q += stride) {
        const uint4 r = rec[q];
        if (counts) counts[q] = r.y - r.x;
        if (status) status[q] = static_cast<uint8_t>(r.w >> 24);
    }
}

void check_records(const void *d_records)
{
    if (!d_records || (reinterpret_cast<uintptr_t>(d_records) & 15u) != 0)
        gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_records must be a non-null 16-byte aligned device pointer");
}
}  // namespace

int gdx_locate_many_search_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                               void *d_records, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if ((reinterpret_cast<uintptr_t>(d_qbuf) & 7u) != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_qbuf must be 8-byte aligned");
        check_records(d_records);
        DeviceGuard guard(f.config().device_id);
        gdx::SearchCall c;
        c.d_qbuf = static_cast<const uint8_t *>(d_qbuf);
        c.d_qbeg = static_cast<const uint64_t *>(d_qoff);
        c.d_qend = c.d_qbeg + 1;
        c.nq = nq;
        c.d_rec = static_cast<uint4 *>(d_records);
        c.mode = 1;
        gdx::launch_search_call(f.view(), c, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_locate_many_offsets_dev(const gdx_index_t *ix, const void *d_records, uint64_t nq, void *d_hit_offsets,
                                void *stream)
{
    return gdx_locate_many_offsets_capped_dev(ix, d_records, nq, 0, d_hit_offsets, stream);
}

int gdx_locate_many_offsets_capped_dev(const gdx_index_t *ix, const void *d_records, uint64_t nq, uint32_t max_hits,
                                       void *d_hit_offsets, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        DeviceGuard guard(f.config().device_id);
        const size_t tb = gdx::hit_offsets_rec_temp_bytes(nq);
        void *temp = gdx::stream_scratch(as_stream(stream), 10, tb ? tb : 1);
        gdx::launch_hit_offsets_rec(static_cast<const uint4 *>(d_records), nq, static_cast<uint64_t *>(d_hit_offsets),
                                    temp, tb, as_stream(stream), max_hits);
        return (int)GDX_OK;
    });
}

int gdx_locate_many_hits_dev(const gdx_index_t *ix, const void *d_records, uint64_t nq, const void *d_hit_offsets,
                             uint64_t total_hits, void *d_hits, void *d_workspace, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        DeviceGuard guard(f.config().device_id);
        gdx::launch_locate(f.view(), nullptr, nullptr, nq, static_cast<const uint64_t *>(d_hit_offsets), total_hits,
                           d_hits, false, d_workspace, as_stream(stream), nullptr, nullptr, f.query_options(),
                           static_cast<const uint4 *>(d_records));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_locate_many_unpack_dev(const gdx_index_t *ix, const void *d_records, uint64_t nq, void *d_out_counts,
                               void *d_out_status, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        if (nq == 0) return (int)GDX_OK;
        DeviceGuard guard(f.config().device_id);
        const uint64_t blocks = (nq + 255) / 256;
        hipLaunchKernelGGL(unpack_records_kernel, dim3(static_cast<unsigned>(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                           as_stream(stream), static_cast<const uint4 *>(d_records), nq,
                           static_cast<uint32_t *>(d_out_counts), static_cast<uint8_t *>(d_out_status));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

// ---- the same with compact results beside the records (gdx.h) -------------------------------------------------

int gdx_locate_many_search_compact_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                       void *d_records, void *d_compact, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if ((reinterpret_cast<uintptr_t>(d_qbuf) & 7u) != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_qbuf must be 8-byte aligned");
        check_records(d_records);
        if (!d_compact && nq != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_compact is null");
        DeviceGuard guard(f.config().device_id);
        gdx::SearchCall c;
        c.d_qbuf = static_cast<const uint8_t *>(d_qbuf);
        c.d_qbeg = static_cast<const uint64_t *>(d_qoff);
        c.d_qend = c.d_qbeg + 1;
        c.nq = nq;
        c.d_rec = static_cast<uint4 *>(d_records);
        c.d_compact = static_cast<uint32_t *>(d_compact);
        c.mode = 1;
        gdx::launch_search_call(f.view(), c, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_locate_many_offsets_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                        uint32_t max_hits, void *d_hit_offsets, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        DeviceGuard guard(f.config().device_id);
        const size_t tb = gdx::hit_offsets_rec_temp_bytes(nq);
        void *temp = gdx::stream_scratch(as_stream(stream), 10, tb ? tb : 1);
        gdx::launch_hit_offsets_rec(static_cast<const uint4 *>(d_records), nq, static_cast<uint64_t *>(d_hit_offsets),
                                    temp, tb, as_stream(stream), max_hits, false, static_cast<const uint32_t *>(d_compact));
        return (int)GDX_OK;
    });
}

int gdx_locate_many_hits_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                     const void *d_hit_offsets, uint64_t total_hits, void *d_hits, void *d_workspace, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        DeviceGuard guard(f.config().device_id);
        gdx::launch_locate(f.view(), nullptr, nullptr, nq, static_cast<const uint64_t *>(d_hit_offsets), total_hits,
                           d_hits, false, d_workspace, as_stream(stream), nullptr, nullptr, f.query_options(),
                           static_cast<const uint4 *>(d_records), false, static_cast<const uint32_t *>(d_compact));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

uint64_t gdx_locate_many_totals_workspace_bytes(uint64_t nq) { return gdx::scan_totals_workspace_bytes(nq); }

int gdx_locate_many_totals_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                       uint32_t max_hits, void *d_scan_workspace, void *d_totals, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        if (!d_scan_workspace || !d_totals) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_locate_many_totals_compact_dev: null argument");
        DeviceGuard guard(f.config().device_id);
        gdx::launch_scan_totals(static_cast<const uint4 *>(d_records), static_cast<const uint32_t *>(d_compact), nq, max_hits,
                                false, d_scan_workspace, static_cast<unsigned long long *>(d_totals), as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

static int offsets_hits_compact(bool narrow, const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                             uint32_t max_hits, const void *d_scan_workspace, void *d_hit_offsets,
                                             uint64_t total_hits, uint64_t rest_hits, void *d_hits, void *d_workspace, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        if (!d_scan_workspace || !d_hit_offsets || (total_hits != 0 && !d_hits))
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_locate_many_offsets_hits_compact_dev: null argument");
        if (narrow && total_hits >= (1ull << 32)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "32-bit hit offsets need fewer than 2^32 hits");
        if ((rest_hits != 0 || d_compact == nullptr) && total_hits != 0 && !d_workspace)
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_locate_many_offsets_hits_compact_dev: d_workspace is null");
        DeviceGuard guard(f.config().device_id);
        gdx::launch_offsets_hits(f.view(), static_cast<const uint4 *>(d_records), static_cast<const uint32_t *>(d_compact), nq, max_hits,
                                 false, d_scan_workspace, d_hit_offsets, narrow, total_hits, rest_hits, d_hits, d_workspace,
                                 as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_locate_many_offsets_hits_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                             uint32_t max_hits, const void *d_scan_workspace, void *d_hit_offsets,
                                             uint64_t total_hits, uint64_t rest_hits, void *d_hits, void *d_workspace, void *stream)
{
    return offsets_hits_compact(false, ix, d_records, d_compact, nq, max_hits, d_scan_workspace, d_hit_offsets, total_hits, rest_hits,
                                d_hits, d_workspace, stream);
}

int gdx_locate_many_offsets32_hits_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                               uint32_t max_hits, const void *d_scan_workspace, void *d_hit_offsets32,
                                               uint64_t total_hits, uint64_t rest_hits, void *d_hits, void *d_workspace, void *stream)
{
    return offsets_hits_compact(true, ix, d_records, d_compact, nq, max_hits, d_scan_workspace, d_hit_offsets32, total_hits, rest_hits,
                                d_hits, d_workspace, stream);
}

int gdx_locate_many_unpack_compact_dev(const gdx_index_t *ix, const void *d_records, const void *d_compact, uint64_t nq,
                                       void *d_out_counts, void *d_out_status, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        DeviceGuard guard(f.config().device_id);
        gdx::launch_unpack_records(static_cast<const uint4 *>(d_records), nq, static_cast<uint32_t *>(d_out_counts),
                                   static_cast<uint8_t *>(d_out_status), as_stream(stream), static_cast<const uint32_t *>(d_compact));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_compact_split_hits_dev(const gdx_index_t *ix, const void *d_compact, uint64_t nq, void *d_out_text_ids,
                               void *d_out_positions, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (d_compact == nullptr || d_out_text_ids == nullptr || d_out_positions == nullptr)
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_compact_split_hits_dev: null buffer");
        if ((reinterpret_cast<uintptr_t>(d_compact) | reinterpret_cast<uintptr_t>(d_out_positions)) % 16 != 0 ||
            reinterpret_cast<uintptr_t>(d_out_text_ids) % 4 != 0)
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_compact_split_hits_dev: buffers must be 16-byte (text ids: 4-byte) aligned");
        if (f.view().n_texts > 256u)
            gdx::fail(GDX_ERR_UNSUPPORTED, "gdx_compact_split_hits_dev: text ids as bytes need a collection of at most 256 texts");
        DeviceGuard guard(f.config().device_id);
        gdx::launch_compact_split(f.view(), static_cast<const uint32_t *>(d_compact), nq, static_cast<uint8_t *>(d_out_text_ids),
                                  static_cast<int32_t *>(d_out_positions), as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_compact_exceptions_dev(const gdx_index_t *ix, const void *d_compact, uint64_t nq, void *d_out_queries,
                               uint64_t capacity, void *d_out_n, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (d_compact == nullptr || d_out_n == nullptr || (d_out_queries == nullptr && capacity != 0))
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_compact_exceptions_dev: null buffer");
        if (reinterpret_cast<uintptr_t>(d_compact) % 16 != 0 || reinterpret_cast<uintptr_t>(d_out_n) % 8 != 0)
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_compact_exceptions_dev: d_compact must be 16-byte, d_out_n 8-byte aligned");
        if (nq > 0xffffffffull) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_compact_exceptions_dev: more than 2^32 - 1 queries");
        DeviceGuard guard(f.config().device_id);
        gdx::launch_compact_exceptions(static_cast<const uint32_t *>(d_compact), nq, static_cast<uint32_t *>(d_out_queries),
                                       capacity, static_cast<unsigned long long *>(d_out_n), as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

uint64_t gdx_wire_bitmap_bytes(uint64_t nq) { return (nq + 2047) / 2048 * 256; }
uint64_t gdx_wire_pack_workspace_bytes(uint64_t nq) { return gdx::wire_pack_workspace_bytes(nq); }

int gdx_wire_pack_dev(const gdx_index_t *ix, const void *d_compact, const void *d_hit_offsets, uint32_t offsets_width,
                      const void *d_hits, uint64_t nq, void *d_bitmap, void *d_tile_found, void *d_found_pos, uint64_t found_capacity,
                      void *d_exc_queries, void *d_exc_counts, uint64_t exc_capacity, void *d_exc_text_ids, void *d_exc_positions,
                      uint64_t exc_hits_capacity, void *d_meta, void *d_workspace, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (!d_tile_found || !d_meta || !d_workspace || (nq != 0 && (!d_compact || !d_hit_offsets || !d_bitmap)) ||
            (found_capacity != 0 && !d_found_pos) || (exc_capacity != 0 && (!d_exc_queries || !d_exc_counts)) ||
            (exc_hits_capacity != 0 && (!d_exc_text_ids || !d_exc_positions || !d_hits)))
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_wire_pack_dev: null buffer");
        if (offsets_width != 32u && offsets_width != 64u) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "offsets_width must be 32 or 64");
        if (reinterpret_cast<uintptr_t>(d_compact) % 16 != 0 || reinterpret_cast<uintptr_t>(d_workspace) % 8 != 0 ||
            (reinterpret_cast<uintptr_t>(d_tile_found) | reinterpret_cast<uintptr_t>(d_found_pos) | reinterpret_cast<uintptr_t>(d_exc_queries) |
             reinterpret_cast<uintptr_t>(d_exc_counts) | reinterpret_cast<uintptr_t>(d_exc_positions) | reinterpret_cast<uintptr_t>(d_meta)) % 4 != 0)
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_wire_pack_dev: d_compact must be 16-byte, d_workspace 8-byte, the u32 / i32 arrays 4-byte aligned");
        if (nq > 0xffffffffull) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_wire_pack_dev: more than 2^32 - 1 queries");
        if (f.view().n_texts > 256u) gdx::fail(GDX_ERR_UNSUPPORTED, "gdx_wire_pack_dev: text ids as bytes need a collection of at most 256 texts");
        DeviceGuard guard(f.config().device_id);
        gdx::launch_wire_pack(static_cast<const uint32_t *>(d_compact), d_hit_offsets, offsets_width == 32u,
                              static_cast<const gdx_hit32_t *>(d_hits), nq, static_cast<uint8_t *>(d_bitmap),
                              static_cast<uint32_t *>(d_tile_found), static_cast<uint32_t *>(d_found_pos), found_capacity,
                              static_cast<uint32_t *>(d_exc_queries), static_cast<uint32_t *>(d_exc_counts), exc_capacity,
                              static_cast<uint8_t *>(d_exc_text_ids), static_cast<int32_t *>(d_exc_positions), exc_hits_capacity,
                              static_cast<uint32_t *>(d_meta), d_workspace, as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_wire_split_dev(const gdx_index_t *ix, const void *d_bitmap, const void *d_tile_found, const void *d_found_pos,
                       uint64_t found_capacity, uint64_t nq, const void *d_exc_queries, const void *d_meta, uint64_t exc_capacity,
                       void *d_out_text_ids, void *d_out_positions, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (!d_tile_found || !d_meta || (nq != 0 && (!d_bitmap || !d_out_text_ids || !d_out_positions)) ||
            (found_capacity != 0 && !d_found_pos) || (exc_capacity != 0 && !d_exc_queries))
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_wire_split_dev: null buffer");
        if (reinterpret_cast<uintptr_t>(d_out_positions) % 16 != 0 || reinterpret_cast<uintptr_t>(d_out_text_ids) % 8 != 0 ||
            (reinterpret_cast<uintptr_t>(d_tile_found) | reinterpret_cast<uintptr_t>(d_found_pos) | reinterpret_cast<uintptr_t>(d_exc_queries) |
             reinterpret_cast<uintptr_t>(d_meta)) % 4 != 0)
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_wire_split_dev: d_out_positions must be 16-byte, d_out_text_ids 8-byte, the u32 arrays 4-byte aligned");
        if (f.view().n_texts > 256u) gdx::fail(GDX_ERR_UNSUPPORTED, "gdx_wire_split_dev: text ids as bytes need a collection of at most 256 texts");
        DeviceGuard guard(f.config().device_id);
        gdx::launch_wire_split(f.view(), static_cast<const uint8_t *>(d_bitmap), static_cast<const uint32_t *>(d_tile_found),
                               static_cast<const uint32_t *>(d_found_pos), found_capacity, nq, static_cast<const uint32_t *>(d_exc_queries),
                               static_cast<const uint32_t *>(d_meta), exc_capacity, static_cast<uint8_t *>(d_out_text_ids),
                               static_cast<int32_t *>(d_out_positions), as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

// ---- packed queries ------------------------------------------------------------------------------------

uint64_t gdx_packed_bytes(uint64_t n_symbols) { return (n_symbols + 3) / 4 / 8 * 8 + 16; }

int gdx_pack_queries(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint8_t *out_packed,
                     uint64_t *out_exceptions, uint64_t exceptions_capacity, uint64_t *out_n_exceptions)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if (exceptions_capacity && !out_exceptions) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_exceptions is null");
        const uint64_t n = f.pack_queries_host(qbuf, qoff, nq, out_packed, out_exceptions, exceptions_capacity);
        if (out_n_exceptions) *out_n_exceptions = n;
        return n > exceptions_capacity ? (int)GDX_ERR_CAPACITY : (int)GDX_OK;
    });
}

int gdx_pack_queries_table(const uint8_t *io_to_dense, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint8_t *out_packed,
                           uint64_t *out_exceptions, uint64_t exceptions_capacity, uint64_t *out_n_exceptions)
{
    return guarded([&] {
        if (exceptions_capacity && !out_exceptions) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_exceptions is null");
        const uint64_t n = gdx::pack_queries_with_table(io_to_dense, qbuf, qoff, nq, out_packed, out_exceptions, exceptions_capacity);
        if (out_n_exceptions) *out_n_exceptions = n;
        return n > exceptions_capacity ? (int)GDX_ERR_CAPACITY : (int)GDX_OK;
    });
}

int gdx_pack_queries_dev(const gdx_index_t *ix, const void *d_qbuf, uint64_t n_symbols, void *d_packed, void *d_bad_flags,
                         void *d_bad_symbols, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        DeviceGuard guard(f.config().device_id);
        gdx::launch_pack_queries(f.view(), static_cast<const uint8_t *>(d_qbuf), n_symbols, static_cast<uint8_t *>(d_packed),
                                 static_cast<uint8_t *>(d_bad_flags), static_cast<unsigned long long *>(d_bad_symbols),
                                 as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_count_many_packed(const gdx_index_t *ix, const uint8_t *packed, const uint64_t *qoff, uint64_t nq,
                          uint64_t *out_counts, uint8_t *out_status)
{
    return guarded([&] {
        if (!out_counts && nq) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_counts is null");
        return deref(ix).cursors_for_many_queries(packed, qoff, nq, nullptr, nullptr, out_counts, out_status, true);
    });
}

int gdx_cursors_for_many_queries_packed(const gdx_index_t *ix, const uint8_t *packed, const uint64_t *qoff, uint64_t nq,
                                        uint64_t *out_start, uint64_t *out_end, uint8_t *out_status)
{
    return guarded([&] {
        if ((!out_start || !out_end) && nq) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_start / out_end is null");
        return deref(ix).cursors_for_many_queries(packed, qoff, nq, out_start, out_end, nullptr, out_status, true);
    });
}

// mode 0 = intervals (start, end), 1 = counts, 2 = records
static int packed_search_dev(const gdx_index_t *ix, const void *d_packed, const void *d_qoff, uint64_t nq, int what,
                             void *d_a, void *d_b, void *d_status, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if ((reinterpret_cast<uintptr_t>(d_packed) & 1u) != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_packed must be 2-byte aligned");
        DeviceGuard guard(f.config().device_id);
        gdx::SearchCall c;
        c.d_qbuf = static_cast<const uint8_t *>(d_packed);
        c.d_qbeg = static_cast<const uint64_t *>(d_qoff);
        c.d_qend = c.d_qbeg + 1;
        c.nq = nq;
        c.packed = true;
        c.d_status = static_cast<uint8_t *>(d_status);
        if (what == 0) {
            c.d_start = static_cast<uint32_t *>(d_a);
            c.d_end = static_cast<uint32_t *>(d_b);
            c.mode = 0;
        } else if (what == 1) {
            c.d_count = static_cast<uint32_t *>(d_a);
            c.mode = 1;
        } else {
            check_records(d_a);
            c.d_rec = static_cast<uint4 *>(d_a);
            c.mode = 1;
        }
        gdx::launch_search_call(f.view(), c, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_cursors_for_many_queries_packed_dev(const gdx_index_t *ix, const void *d_packed, const void *d_qoff, uint64_t nq,
                                            void *d_out_start, void *d_out_end, void *d_out_status, void *stream)
{
    return packed_search_dev(ix, d_packed, d_qoff, nq, 0, d_out_start, d_out_end, d_out_status, stream);
}

int gdx_count_many_packed_dev(const gdx_index_t *ix, const void *d_packed, const void *d_qoff, uint64_t nq,
                              void *d_out_counts, void *d_out_status, void *stream)
{
    return packed_search_dev(ix, d_packed, d_qoff, nq, 1, d_out_counts, nullptr, d_out_status, stream);
}

int gdx_locate_many_search_packed_dev(const gdx_index_t *ix, const void *d_packed, const void *d_qoff, uint64_t nq,
                                      void *d_records, void *stream)
{
    return packed_search_dev(ix, d_packed, d_qoff, nq, 2, d_records, nullptr, nullptr, stream);
}

// ---- query layouts: packed and / or uniform batches through one description (gdx.h) ------------------------------

void gdx_query_layout_init(gdx_query_layout_t *layout)
{
    if (!layout) return;
    std::memset(layout, 0, sizeof(*layout));
    layout->struct_size = sizeof(*layout);
}

namespace {
// fills the query side of a SearchCall from a layout (null = ASCII with offsets: the plain calls)
void apply_layout(gdx::SearchCall &c, const void *d_qbuf, const void *d_qoff, uint64_t nq, const gdx_query_layout_t *layout)
{
    gdx_query_layout_t l;
    gdx_query_layout_init(&l);
    if (layout) {
        if (layout->struct_size < 16 || layout->struct_size > sizeof(l)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_layout_t.struct_size");
        std::memcpy(&l, layout, layout->struct_size);
    }
    if (l.packed != 0 && l.packed != 1) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_layout_t.packed must be 0 or 1");
    if (l.uniform_len >= (1ull << 21)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_layout_t.uniform_len must be below 2^21");
    const uintptr_t align = l.packed ? 1u : 7u;
    if ((reinterpret_cast<uintptr_t>(d_qbuf) & align) != 0)
        gdx::fail(GDX_ERR_INVALID_ARGUMENT, l.packed ? "d_qbuf (packed) must be 2-byte aligned" : "d_qbuf must be 8-byte aligned");
    if (l.uniform_len == 0 && !d_qoff && nq != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_qoff is null and the layout is not uniform");
    c.d_qbuf = static_cast<const uint8_t *>(d_qbuf);
    c.nq = nq;
    c.packed = l.packed != 0;
    c.uniform_len = static_cast<uint32_t>(l.uniform_len);
    if (l.uniform_len == 0) {
        c.d_qbeg = static_cast<const uint64_t *>(d_qoff);
        c.d_qend = c.d_qbeg + 1;
    }
}
}  // namespace

// host-pointer calls on a batch in a layout (the chunked pipeline of host_api.hip)
namespace {
void host_layout(const gdx_query_layout_t *layout, bool &packed, uint64_t &uniform_len)
{
    gdx_query_layout_t l;
    gdx_query_layout_init(&l);
    if (layout) {
        if (layout->struct_size < 16 || layout->struct_size > sizeof(l)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_layout_t.struct_size");
        std::memcpy(&l, layout, layout->struct_size);
    }
    if (l.packed != 0 && l.packed != 1) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_layout_t.packed must be 0 or 1");
    if (l.uniform_len >= (1ull << 21)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_query_layout_t.uniform_len must be below 2^21");
    packed = l.packed != 0;
    uniform_len = l.uniform_len;
}
}  // namespace

int gdx_count_many_layout(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                          const gdx_query_layout_t *layout, uint64_t *out_counts, uint8_t *out_status)
{
    return guarded([&] {
        bool packed;
        uint64_t ulen;
        host_layout(layout, packed, ulen);
        if (!out_counts && nq != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_counts is null");
        return deref(ix).cursors_for_many_queries(qbuf, qoff, nq, nullptr, nullptr, out_counts, out_status, packed, ulen);
    });
}

int gdx_cursors_for_many_queries_layout(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                        const gdx_query_layout_t *layout, uint64_t *out_start, uint64_t *out_end,
                                        uint8_t *out_status)
{
    return guarded([&] {
        bool packed;
        uint64_t ulen;
        host_layout(layout, packed, ulen);
        if ((!out_start || !out_end) && nq != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_start / out_end is null");
        return deref(ix).cursors_for_many_queries(qbuf, qoff, nq, out_start, out_end, nullptr, out_status, packed, ulen);
    });
}

int gdx_locate_many_alloc_layout(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                 const gdx_query_layout_t *layout, uint64_t *out_hit_offsets, gdx_hit_t **out_hits,
                                 uint64_t *out_total, uint8_t *out_status)
{
    return guarded([&] {
        bool packed;
        uint64_t ulen;
        host_layout(layout, packed, ulen);
        return deref(ix).locate_many_alloc(qbuf, qoff, nq, out_hit_offsets, out_hits, out_total, out_status, packed, ulen);
    });
}

int gdx_locate_many_alloc_layout32(const gdx_index_t *ix, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                   const gdx_query_layout_t *layout, gdx_hits32_t *out_results, uint8_t *out_status)
{
    return guarded([&] {
        bool packed;
        uint64_t ulen;
        host_layout(layout, packed, ulen);
        return deref(ix).locate_many_alloc32(qbuf, qoff, nq, out_results, out_status, packed, ulen);
    });
}

void gdx_free_hits32(gdx_hits32_t *results) { gdx::recycle_hits32(results); }
void gdx_release_cached_hits(void) { gdx::release_cached_hits(); }

int gdx_locate_many_search_compact_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                              const gdx_query_layout_t *layout, void *d_records, void *d_compact, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        if (!d_compact && nq != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_compact is null");
        DeviceGuard guard(f.config().device_id);
        gdx::SearchCall c;
        apply_layout(c, d_qbuf, d_qoff, nq, layout);
        c.d_rec = static_cast<uint4 *>(d_records);
        c.d_compact = static_cast<uint32_t *>(d_compact);
        c.mode = 1;
        gdx::launch_search_call(f.view(), c, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_locate_many_search_totals_compact_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                                     const gdx_query_layout_t *layout, uint32_t max_hits, void *d_records,
                                                     void *d_compact, void *d_scan_workspace, void *d_totals, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        if (!d_compact && nq != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_compact is null");
        if (!d_scan_workspace || !d_totals) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_locate_many_search_totals_compact_layout_dev: null argument");
        DeviceGuard guard(f.config().device_id);
        unsigned long long *totals = static_cast<unsigned long long *>(d_totals);
        GDX_HIP(hipMemsetAsync(totals, 0, 2 * sizeof(unsigned long long), as_stream(stream)));
        gdx::SearchCall c;
        apply_layout(c, d_qbuf, d_qoff, nq, layout);
        c.d_rec = static_cast<uint4 *>(d_records);
        c.d_compact = static_cast<uint32_t *>(d_compact);
        c.mode = 1;
        bool folded = false;
        c.d_tile_sums = static_cast<unsigned long long *>(d_scan_workspace);
        c.d_tile_rest = totals + 1;
        c.tile_max_hits = max_hits;
        c.tile_sums_done = &folded;
        gdx::launch_search_call(f.view(), c, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        if (folded)  // the search counted the hits per tile itself: only the scan of the tile sums is left
            gdx::launch_scan_totals_finish(d_scan_workspace, nq, totals, as_stream(stream));
        else
            gdx::launch_scan_totals(static_cast<const uint4 *>(d_records), static_cast<const uint32_t *>(d_compact), nq, max_hits,
                                    false, d_scan_workspace, totals, as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_locate_many_step_compact_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                            const gdx_query_layout_t *layout, uint32_t max_hits, void *d_records, void *d_compact,
                                            void *d_scan_workspace, void *d_totals, void *d_hit_offsets, uint32_t offsets_width,
                                            void *d_hits, uint64_t hits_capacity, void *d_workspace, void *event_after_search,
                                            void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        if (!d_scan_workspace || !d_totals || !d_hit_offsets || (hits_capacity != 0 && (!d_hits || !d_workspace)))
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_locate_many_step_compact_layout_dev: null argument");
        if (offsets_width != 32u && offsets_width != 64u) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "offsets_width must be 32 or 64");
        const bool narrow = offsets_width == 32u;
        if (narrow && hits_capacity >= (1ull << 32)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "32-bit hit offsets need fewer than 2^32 hits");
        if ((reinterpret_cast<uintptr_t>(d_totals) & 7u) != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_totals must be 8-byte aligned");
        DeviceGuard guard(f.config().device_id);
        gdx::LocateStep step;
        apply_layout(step.call, d_qbuf, d_qoff, nq, layout);
        step.call.d_rec = static_cast<uint4 *>(d_records);
        step.call.d_compact = static_cast<uint32_t *>(d_compact);
        step.max_hits = max_hits;
        step.d_scan_workspace = d_scan_workspace;
        step.d_totals = static_cast<unsigned long long *>(d_totals);
        step.d_hit_offsets = d_hit_offsets;
        step.narrow = narrow;
        step.d_hits = d_hits;
        step.hits_capacity = hits_capacity;
        step.d_workspace = d_workspace;
        step.event_after_search = static_cast<hipEvent_t>(event_after_search);
        gdx::launch_locate_step(f.view(), step, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_locate_many_search_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                      const gdx_query_layout_t *layout, void *d_records, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        check_records(d_records);
        DeviceGuard guard(f.config().device_id);
        gdx::SearchCall c;
        apply_layout(c, d_qbuf, d_qoff, nq, layout);
        c.d_rec = static_cast<uint4 *>(d_records);
        c.mode = 1;
        gdx::launch_search_call(f.view(), c, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_count_many_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                              const gdx_query_layout_t *layout, void *d_out_counts, void *d_out_status, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        DeviceGuard guard(f.config().device_id);
        gdx::SearchCall c;
        apply_layout(c, d_qbuf, d_qoff, nq, layout);
        c.d_count = static_cast<uint32_t *>(d_out_counts);
        c.d_status = static_cast<uint8_t *>(d_out_status);
        c.mode = 1;
        gdx::launch_search_call(f.view(), c, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_cursors_for_many_queries_layout_dev(const gdx_index_t *ix, const void *d_qbuf, const void *d_qoff, uint64_t nq,
                                            const gdx_query_layout_t *layout, void *d_out_start, void *d_out_end,
                                            void *d_out_status, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        DeviceGuard guard(f.config().device_id);
        gdx::SearchCall c;
        apply_layout(c, d_qbuf, d_qoff, nq, layout);
        c.d_start = static_cast<uint32_t *>(d_out_start);
        c.d_end = static_cast<uint32_t *>(d_out_end);
        c.d_status = static_cast<uint8_t *>(d_out_status);
        c.mode = 0;
        gdx::launch_search_call(f.view(), c, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_cursor_extend_front_strings_dev(const gdx_index_t *ix, void *d_start, void *d_end, const void *d_qbuf,
                                        const void *d_qbeg, const void *d_qend, uint64_t m, void *d_status,
                                        const void *d_active_in, const void *d_n_active_in, void *d_active_out,
                                        void *d_n_active_out, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if ((reinterpret_cast<uintptr_t>(d_qbuf) & 7u) != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_qbuf must be 8-byte aligned");
        if (m && (!d_start || !d_end || !d_qbeg || !d_qend)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
        if ((d_active_out == nullptr) != (d_n_active_out == nullptr) || (d_active_in != nullptr && d_n_active_in == nullptr))
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "active lists need their counters");
        if (m >= 0xffffffffull) gdx::fail(GDX_ERR_UNSUPPORTED, "more than 2^32-2 cursors in one call");
        DeviceGuard guard(f.config().device_id);
        if (d_n_active_out) GDX_HIP(hipMemsetAsync(d_n_active_out, 0, sizeof(uint32_t), as_stream(stream)));
        gdx::SearchCall c;
        c.d_qbuf = static_cast<const uint8_t *>(d_qbuf);
        c.d_qbeg = static_cast<const uint64_t *>(d_qbeg);
        c.d_qend = static_cast<const uint64_t *>(d_qend);
        c.nq = m;
        c.d_start = static_cast<uint32_t *>(d_start);
        c.d_end = static_cast<uint32_t *>(d_end);
        c.d_status = static_cast<uint8_t *>(d_status);
        c.mode = 2;
        c.cursors.active_in = static_cast<const uint32_t *>(d_active_in);
        c.cursors.n_active_in = static_cast<const uint32_t *>(d_n_active_in);
        c.cursors.active_out = static_cast<uint32_t *>(d_active_out);
        c.cursors.n_active_out = static_cast<uint32_t *>(d_n_active_out);
        gdx::launch_search_call(f.view(), c, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_cursor_extend_front_chunk_dev(const gdx_index_t *ix, void *d_start, void *d_end, const void *d_qbuf,
                                      const void *d_qoff, uint64_t m, uint32_t chunk_symbols, uint32_t chunk_index,
                                      void *d_status, const void *d_active_in, const void *d_n_active_in,
                                      void *d_active_out, void *d_n_active_out, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        if ((reinterpret_cast<uintptr_t>(d_qbuf) & 7u) != 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "d_qbuf must be 8-byte aligned");
        if (m && (!d_start || !d_end || !d_qoff)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
        if (chunk_symbols == 0) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "chunk_symbols must be positive");
        if ((d_active_out == nullptr) != (d_n_active_out == nullptr) || (d_active_in != nullptr && d_n_active_in == nullptr))
            gdx::fail(GDX_ERR_INVALID_ARGUMENT, "active lists need their counters");
        if (m >= 0xffffffffull) gdx::fail(GDX_ERR_UNSUPPORTED, "more than 2^32-2 cursors in one call");
        DeviceGuard guard(f.config().device_id);
        if (d_n_active_out) GDX_HIP(hipMemsetAsync(d_n_active_out, 0, sizeof(uint32_t), as_stream(stream)));
        gdx::SearchCall c;
        c.d_qbuf = static_cast<const uint8_t *>(d_qbuf);
        c.d_qbeg = static_cast<const uint64_t *>(d_qoff);
        c.d_qend = static_cast<const uint64_t *>(d_qoff) + 1;
        c.nq = m;
        c.d_start = static_cast<uint32_t *>(d_start);
        c.d_end = static_cast<uint32_t *>(d_end);
        c.d_status = static_cast<uint8_t *>(d_status);
        c.mode = 2;
        c.cursors.active_in = static_cast<const uint32_t *>(d_active_in);
        c.cursors.n_active_in = static_cast<const uint32_t *>(d_n_active_in);
        c.cursors.active_out = static_cast<uint32_t *>(d_active_out);
        c.cursors.n_active_out = static_cast<uint32_t *>(d_n_active_out);
        c.cursors.chunk_symbols = chunk_symbols;
        c.cursors.chunk_index = chunk_index;
        gdx::launch_search_call(f.view(), c, as_stream(stream), f.query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_cursor_extend_front_strings(const gdx_index_t *ix, uint64_t *start, uint64_t *end, const uint8_t *qbuf,
                                    const uint64_t *qoff, uint64_t m, uint8_t *status)
{
    return guarded([&] { return deref(ix).cursor_extend_front_strings(start, end, qbuf, qoff, m, status); });
}

int gdx_rank_many_dev(const gdx_index_t *ix, const void *d_symbols, const void *d_idx, uint64_t m, void *d_out,
                      void *d_error, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        DeviceGuard guard(f.config().device_id);
        uint32_t *d_err = static_cast<uint32_t *>(d_error);
        if (!d_err) d_err = static_cast<uint32_t *>(gdx::stream_scratch(as_stream(stream), 9, sizeof(uint32_t)));
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

int gdx_debug_set_host_chunking(uint64_t queries, uint64_t bytes)
{
    gdx::set_host_chunking(queries, bytes);
    return GDX_OK;
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
        DeviceGuard guard(deref(ix).config().device_id);
        gdx::launch_search(deref(ix).view(), static_cast<const uint8_t *>(d_qbuf),
                           static_cast<const uint64_t *>(d_qoff), nq, nullptr, nullptr, nullptr, nullptr,
                           as_stream(stream), static_cast<unsigned long long *>(d_steps), nullptr,
                           deref(ix).query_options());
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_bench_lf_walk_dev(const gdx_index_t *ix, const void *d_rows, uint64_t m, uint32_t steps, void *d_symbols,
                          void *d_end_rows, void *stream)
{
    return guarded([&] {
        DeviceGuard guard(deref(ix).config().device_id);
        if (m != 0 && (!d_rows || !d_symbols)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_bench_lf_walk_dev: null argument");
        gdx::launch_lf_walk(deref(ix).view(), static_cast<const uint32_t *>(d_rows), m, steps,
                            static_cast<uint8_t *>(d_symbols), static_cast<uint32_t *>(d_end_rows), as_stream(stream));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_debug_force_wide(int on)
{
    g_force_wide.store(on != 0 ? 1 : 0);
    return GDX_OK;
}

int gdx_index_aux_info(const gdx_index_t *ix, uint32_t out[4])
{
    return guarded([&] {
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        const gdx::IndexView &v = deref(ix).view();
        out[0] = v.pair_lines != nullptr;
        out[1] = v.jump ? v.jump_bytes : 0u;
        out[2] = v.top ? v.top_depth : 0u;
        out[3] = (v.sa_full ? 1u : 0u) | (v.text_units ? 2u : 0u) | (v.isa ? 4u : 0u) | (deref(ix).aux_report().default_shape ? 8u : 0u);
        return (int)GDX_OK;
    });
}

int gdx_locate_step_stats_dev(const gdx_index_t *ix, const void *d_start, const void *d_end, uint64_t m,
                              const void *d_hit_offsets, uint64_t total_hits, void *d_hits, void *d_workspace,
                              void *d_steps, void *stream)
{
    return guarded([&] {
        DeviceGuard guard(deref(ix).config().device_id);
        gdx::launch_locate(deref(ix).view(), static_cast<const uint32_t *>(d_start),
                           static_cast<const uint32_t *>(d_end), m, static_cast<const uint64_t *>(d_hit_offsets),
                           total_hits, d_hits, false, d_workspace, as_stream(stream),
                           static_cast<unsigned long long *>(d_steps), nullptr, deref(ix).query_options(), nullptr,
                           true);
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

int gdx_locate_many_hits_stats_dev(const gdx_index_t *ix, const void *d_records, uint64_t nq, const void *d_hit_offsets,
                                   uint64_t total_hits, void *d_hits, void *d_workspace, void *d_steps, void *stream)
{
    return guarded([&] {
        const gdx::FmIndex &f = deref(ix);
        DeviceGuard guard(f.config().device_id);
        gdx::launch_locate(f.view(), nullptr, nullptr, nq, static_cast<const uint64_t *>(d_hit_offsets), total_hits,
                           d_hits, false, d_workspace, as_stream(stream), static_cast<unsigned long long *>(d_steps),
                           nullptr, f.query_options(), static_cast<const uint4 *>(d_records));
        GDX_HIP(hipGetLastError());
        return (int)GDX_OK;
    });
}

// ---- replicas on several GPUs behind one handle -----------------------------------------------------------

int gdx_multi_from_indexes(gdx_index_t **replicas, int n_replicas, gdx_multi_t **out)
{
    return guarded([&] {
        if (!out || !replicas || n_replicas < 1) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_multi_from_indexes: bad argument");
        *out = nullptr;
        for (int r = 0; r < n_replicas; r++) {
            if (!replicas[r] || !replicas[r]->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "replica %d is null", r);
            for (int k = 0; k < r; k++)
                if (replicas[k] == replicas[r]) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "replica %d is the same handle as replica %d", r, k);
            const gdx::FmIndex &a = *replicas[0]->impl, &b = *replicas[r]->impl;
            if (a.total_text_len() != b.total_text_len() || a.num_texts() != b.num_texts() ||
                a.config().sa_rate != b.config().sa_rate || a.config().sigma != b.config().sigma ||
                a.config().n_searchable != b.config().n_searchable || a.config().lookup_depth != b.config().lookup_depth ||
                std::memcmp(a.config().io_to_dense, b.config().io_to_dense, 256) != 0)
                gdx::fail(GDX_ERR_INVALID_ARGUMENT, "replica %d was not built from the same texts / configuration", r);
        }
        auto m = std::make_unique<gdx::Multi>();
        for (int r = 0; r < n_replicas; r++) {
            m->replicas.push_back(std::move(replicas[r]->impl));
            delete replicas[r];  // the handle is consumed
            replicas[r] = nullptr;
        }
        m->start_workers();
        *out = new gdx_multi{std::move(m)};
        return (int)GDX_OK;
    });
}

int gdx_multi_build(const uint8_t *texts_buf, const uint64_t *text_offsets, uint64_t n_texts, const uint8_t *io_to_dense,
                    int sigma, int n_searchable, uint64_t sa_rate, int lookup_depth, int index_width,
                    const int *device_ids, int n_devices, const gdx_build_options_t *opts, gdx_multi_t **out)
{
    return guarded([&] {
        if (!out || !device_ids || n_devices < 1) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_multi_build: bad argument");
        *out = nullptr;
        auto m = std::make_unique<gdx::Multi>();
        m->replicas.resize(n_devices);
        std::vector<std::string> errors(n_devices);
        std::vector<int> status(n_devices, GDX_OK);
        std::vector<std::thread> threads;
        for (int r = 0; r < n_devices; r++) {  // every device builds its own replica (the builder is deterministic)
            threads.emplace_back([&, r] {
                try {
                    auto cfg = make_config(io_to_dense, sigma, n_searchable, sa_rate, lookup_depth, index_width, device_ids[r], opts);
                    m->replicas[r] = gdx::FmIndex::construct_index(texts_buf, false, text_offsets, n_texts, cfg);
                } catch (const gdx::Error &e) {
                    status[r] = e.status;
                    errors[r] = e.what();
                } catch (const std::exception &e) {
                    status[r] = GDX_ERR_DEVICE;
                    errors[r] = e.what();
                }
            });
        }
        for (auto &t : threads) t.join();
        for (int r = 0; r < n_devices; r++)
            if (status[r] != GDX_OK) gdx::fail(status[r], "device %d: %s", device_ids[r], errors[r].c_str());
        m->start_workers();
        *out = new gdx_multi{std::move(m)};
        return (int)GDX_OK;
    });
}

void gdx_multi_free(gdx_multi_t *m)
{
    if (!m) return;
    (void)guarded([&] {
        if (m->impl) m->impl->workers.clear();  // joins the worker threads (their buffer caches go with them)
        if (m->impl)
            for (auto &rep : m->impl->replicas)
                if (rep) {
                    DeviceGuard guard(rep->config().device_id);
                    rep.reset();
                }
        delete m;
        return (int)GDX_OK;
    });
}

int gdx_multi_replicas(const gdx_multi_t *m) { return (m && m->impl) ? static_cast<int>(m->impl->replicas.size()) : 0; }

int gdx_multi_count_many(const gdx_multi_t *m, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_counts,
                         uint8_t *out_status)
{
    return guarded([&] {
        if (!m || !m->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "multi handle is null");
        if (!out_counts && nq) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_counts is null");
        return gdx::multi_cursors(*m->impl, qbuf, qoff, nq, nullptr, nullptr, out_counts, out_status);
    });
}

int gdx_multi_cursors_for_many_queries(const gdx_multi_t *m, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                       uint64_t *out_start, uint64_t *out_end, uint8_t *out_status)
{
    return guarded([&] {
        if (!m || !m->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "multi handle is null");
        if ((!out_start || !out_end) && nq) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out_start / out_end is null");
        return gdx::multi_cursors(*m->impl, qbuf, qoff, nq, out_start, out_end, nullptr, out_status);
    });
}

int gdx_multi_locate_many_gather_dev(gdx_multi_t *m, const gdx_device_shard_t *shards, int n_shards, int root,
                                     gdx_gathered_t *out)
{
    return guarded([&] {
        if (!m || !m->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "multi handle is null");
        if (!shards || !out || n_shards < 1) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "gdx_multi_locate_many_gather_dev: bad argument");
        int prev = 0;
        GDX_HIP(hipGetDevice(&prev));
        std::vector<gdx::DeviceShard> sh(n_shards);
        for (int r = 0; r < n_shards; r++)
            sh[r] = gdx::DeviceShard{static_cast<const uint8_t *>(shards[r].d_qbuf), static_cast<const uint64_t *>(shards[r].d_qoff), shards[r].nq};
        gdx::Gathered g{};
        try {
            gdx::multi_locate_gather_dev(*m->impl, sh.data(), n_shards, root, &g);
        } catch (...) {
            (void)hipSetDevice(prev);
            throw;
        }
        (void)hipSetDevice(prev);
        out->d_counts = g.d_counts;
        out->d_hit_offsets = g.d_hit_offsets;
        out->d_hits = g.d_hits;
        out->d_status = g.d_status;
        out->nq = g.nq;
        out->total_hits = g.total_hits;
        out->device_id = g.device_id;
        out->used_rccl = g.used_rccl;
        return (int)GDX_OK;
    });
}

int gdx_multi_locate_many_alloc(const gdx_multi_t *m, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                uint64_t *out_hit_offsets, gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status)
{
    return guarded([&] {
        if (!m || !m->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "multi handle is null");
        return gdx::multi_locate_alloc(*m->impl, qbuf, qoff, nq, out_hit_offsets, out_hits, out_total, out_status);
    });
}

// ---- partitioned index (parts.hip) ---------------------------------------------------------------------------------

int gdx_parts_build(const void *texts_buf, int texts_on_device, const uint64_t *text_offsets, uint64_t n_texts,
                    const uint8_t *io_to_dense, int sigma, int n_searchable, uint64_t sa_rate, int lookup_depth,
                    int device_id, uint64_t max_part_symbols, const gdx_build_options_t *opts, gdx_parts_t **out)
{
    return guarded([&] {
        if (!out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        auto cfg = make_config(io_to_dense, sigma, n_searchable, sa_rate, lookup_depth, 32, device_id, opts);
        DeviceGuard guard(device_id);
        auto impl = gdx::Parts::build(static_cast<const uint8_t *>(texts_buf), texts_on_device != 0, text_offsets, n_texts, cfg,
                                      max_part_symbols);
        *out = new gdx_parts{std::move(impl)};
        return (int)GDX_OK;
    });
}

void gdx_parts_free(gdx_parts_t *p)
{
    if (!p) return;
    (void)guarded([&] {
        if (p->impl && !p->impl->parts.empty()) {
            DeviceGuard guard(p->impl->cfg.device_id);
            p->impl->parts.clear();
        }
        delete p;
        return (int)GDX_OK;
    });
}

int gdx_parts_info(const gdx_parts_t *p, uint64_t out[4])
{
    return guarded([&] {
        if (!p || !p->impl || !out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "null argument");
        out[0] = p->impl->parts.size();
        out[1] = p->impl->total_len;
        out[2] = p->impl->first_text.back();
        out[3] = 0;
        for (auto &ix : p->impl->parts) out[3] += ix->device_bytes();
        return (int)GDX_OK;
    });
}

int gdx_parts_set_query_options(gdx_parts_t *p, const gdx_query_options_t *opts)
{
    return guarded([&] {
        if (!p || !p->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "parts handle is null");
        const gdx::QueryOptions q = parse_query_options(opts);
        for (auto &ix : p->impl->parts) ix->set_query_options(q);
        return (int)GDX_OK;
    });
}

int gdx_parts_count_many(const gdx_parts_t *p, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                         uint64_t *out_counts, uint8_t *out_status)
{
    return guarded([&] {
        if (!p || !p->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "parts handle is null");
        DeviceGuard guard(p->impl->cfg.device_id);
        return p->impl->count_many(qbuf, qoff, nq, out_counts, out_status);
    });
}

int gdx_parts_locate_many_alloc(const gdx_parts_t *p, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                uint64_t *out_hit_offsets, gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status)
{
    return guarded([&] {
        if (!p || !p->impl) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "parts handle is null");
        DeviceGuard guard(p->impl->cfg.device_id);
        return p->impl->locate_many_alloc(qbuf, qoff, nq, out_hit_offsets, out_hits, out_total, out_status);
    });
}

// ---- FASTA / FASTQ ingestion (host only) ------------------------------------------------------------------

int gdx_fastx_open(const char *path, gdx_fastx_t **out)
{
    return guarded([&] {
        if (!path || !out) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "path / out is null");
        auto r = std::make_unique<gdx_fastx>();
        // GDX_FASTX_THREADS: parser threads of the mapped reader (default: the CPUs the process may use, at most 32); 0 = the
        // streaming reader
        unsigned threads = gdx::fastx_default_threads();
        if (const char *e = getenv("GDX_FASTX_THREADS")) threads = static_cast<unsigned>(std::max(0, atoi(e)));
        if (threads != 0) r->mapped.reset(gdx::FastxMappedReader::open(path, threads));
        if (!r->mapped) r->impl = std::make_unique<gdx::FastxReader>(path);
        *out = r.release();
        return (int)GDX_OK;
    });
}

int gdx_fastx_next_batch_ex(gdx_fastx_t *reader, uint8_t *qbuf, uint64_t qbuf_capacity, uint64_t *qoff,
                            uint64_t max_records, uint64_t *n_out, uint64_t *out_uniform_len)
{
    return guarded([&] {
        if (!reader || (!reader->impl && !reader->mapped)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "reader handle is null");
        if (!qoff || !n_out || (!qbuf && qbuf_capacity)) gdx::fail(GDX_ERR_INVALID_ARGUMENT, "output pointer is null");
        if (reader->mapped) {
            *n_out = reader->mapped->next_batch(qbuf, qbuf_capacity, qoff, max_records, out_uniform_len);
            return (int)GDX_OK;
        }
        const uint64_t n = reader->impl->next_batch(qbuf, qbuf_capacity, qoff, max_records);
        *n_out = n;
        if (out_uniform_len) {
            uint64_t len = n ? qoff[1] - qoff[0] : 0;
            for (uint64_t i = 1; i < n && len != 0; i++)
                if (qoff[i + 1] - qoff[i] != len) len = 0;
            *out_uniform_len = len;
        }
        return (int)GDX_OK;
    });
}

int gdx_fastx_next_batch(gdx_fastx_t *reader, uint8_t *qbuf, uint64_t qbuf_capacity, uint64_t *qoff,
                         uint64_t max_records, uint64_t *n_out)
{
    return gdx_fastx_next_batch_ex(reader, qbuf, qbuf_capacity, qoff, max_records, n_out, nullptr);
}

void gdx_fastx_close(gdx_fastx_t *reader) { delete reader; }

}  // extern "C"
