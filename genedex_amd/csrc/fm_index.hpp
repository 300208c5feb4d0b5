// fm_index.hpp -- host-side FM index object living in HBM.
//
// Mirrors the reference's FmIndex<I, CondensedTextWithRankSupport<I, Block64>> (lib.rs:89-100):
// same members (alphabet table, count, occurrence table, sampled suffix array, text ids, lookup
// tables), same query entry points (count_many / cursors_for_many_queries / locate_many, the cursor
// operations), batched by construction.  The C ABI in include/gdx.h is a thin shim over this class.
#pragma once

#include <atomic>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <vector>

#include "build.hpp"
#include "common.hpp"
#include "kernels.hpp"
#include "layout.hpp"

namespace gdx {

// Build-time choices of this implementation that the reference has no counterpart for (gdx_build_options_t).
// Negative / zero = default.  The GDX_* environment variables named in fm_index.hip only override fields left
// at their default (debugging aid).
struct BuildOptions {
    // (all of jump_bytes / full_sa / text_units / seed_symbols / inverse_sa at -1 and pair lines not switched off: THE DEFAULT
    // SHAPE -- seed table + text units + full and inverse suffix array + pair lines + top table of depth <= 14, no jump table
    // -- when it fits the budget; FmIndex::build_aux)
    int pair_lines = -1;            // -1 default (on for sigma <= 8), 0 off, 1 on
    int jump_bytes = -1;            // -1 default (none in the default shape, else 32), 0 none, 8, 16, 32
    int top_depth = -1;             // -1 default (largest even D <= 16 with 4^D <= 2n; <= 14 in the default shape), 0 none, 1..16
    uint64_t aux_budget_bytes = 0;  // cap for jump + top table; 0 = min(free HBM - reserve, half of the HBM)
    int full_sa = -1;               // 1: SA[row] of every row as its own array
    int text_units = -1;            // 1: the text itself, 16 bytes per 32 symbols (layout.hpp)
    int seed_symbols = -1;          // -1 / 0 none, 1 = seed table with k chosen from the text length, 8..24 = that k
                                    // (implies text_units)
    int seed_load_percent = 0;      // slots filled on average, 0 = default (70), 20..100
    int table_layout = -1;          // -1 / 0: this implementation's own occurrence table (rank lines for sigma <= 8); 1..4: the
                                    // reference's Condensed64 / Condensed512 / Flat64 / Flat512 layout, queried as it is
    int inverse_sa = -1;            // 1: ISA[position] = row as its own array (4 bytes per symbol): exact intervals through the
                                    // seed table
};

// What the aux build decided (gdx_index_aux_t)
struct AuxReport {
    uint32_t wanted_jump_bytes = 0, wanted_top_depth = 0;  // before the budget was applied
    uint64_t budget_bytes = 0, aux_bytes = 0;
    double wide_fraction = 0.0;  // share of text positions whose top-table interval is wider than 4 rows
    bool default_shape = false;  // the options were left at their defaults and the default shape fitted the budget (fm_index.hip)
    // seed table: k, buckets, entries of kind 0 / kind 1, buckets that turned an entry away, largest displacement
    uint64_t seed_k = 0, seed_buckets = 0, seed_single = 0, seed_multi = 0, seed_overflowed = 0, seed_max_disp = 0, seed_bytes = 0, seed_pair_records = 0, seed_quad_records = 0;
};

struct IndexConfig {
    uint8_t io_to_dense[256];  // alphabet.rs:24-28
    int sigma = 0;             // num_dense_symbols (incl. sentinel)
    int n_searchable = 0;      // num_searchable_dense_symbols
    uint64_t sa_rate = 4;      // config.rs:75 suffix_array_sampling_rate
    int lookup_depth = 0;      // config.rs:76 lookup_table_depth
    int index_width = 32;      // 32 = u32, -32 = i32, 64 = i64
    int device_id = 0;
    BuildOptions build;
};

struct Narrow32Sink;  // (below)

class FmIndex {
public:
    // FmIndexConfig::construct_index (config.rs:63-69).  texts_buf: concatenated IO symbols of all
    // texts, on the host or already on the device.
    static std::unique_ptr<FmIndex> construct_index(const uint8_t *texts_buf, bool texts_on_device,
                                                    const uint64_t *text_offsets, uint64_t n_texts,
                                                    const IndexConfig &cfg);
    // import of the reference's logical arrays (host pointers)
    // table_kind 0 = condensed, 1 = flat; block_bits 64 | 512 (the reference's four table variants)
    static std::unique_ptr<FmIndex> from_parts(int table_kind, int block_bits, const uint64_t *count,
                                               const uint64_t *interleaved_blocks, uint64_t n,
                                               const uint32_t *sa_samples, const uint64_t *border_keys,
                                               const uint64_t *border_vals, const uint64_t *sentinel_indices,
                                               uint64_t n_texts, const IndexConfig &cfg);
    // own file format around the reference's logical arrays (lib.rs:296-327 save_to_file / load_from_file)
    void save(const char *path) const;
    static std::unique_ptr<FmIndex> load(const char *path, int device_id, const BuildOptions &build = BuildOptions());
    ~FmIndex();

    const IndexView &view() const { return view_; }
    const IndexConfig &config() const { return cfg_; }
    const BuildStats &build_stats() const { return stats_; }
    uint64_t total_text_len() const { return n_; }  // lib.rs:291-294
    uint64_t num_texts() const { return n_texts_; }  // lib.rs:287-289
    uint64_t device_bytes() const;
    void make_current() const;  // hipSetDevice(device_id)
    const AuxReport &aux_report() const { return aux_report_; }
    // kernel choices of the query calls on this handle (gdx_query_options_t); safe against concurrent queries
    void set_query_options(const QueryOptions &q);
    QueryOptions query_options() const;
    // drops the pair lines / jump table / top table and builds them again with other options (bench ladder;
    // not safe against concurrent queries)
    void rebuild_aux(const BuildOptions &opts);
    // search -> scan -> locate of a device-resident batch on this replica's device, results left in the given buffers
    // (grown as needed): counts u32[nq], status u8[nq], hits gdx_hit32_t[total]; returns the number of hits
    uint64_t locate_shard_dev(const uint8_t *d_qbuf, const uint64_t *d_qoff, uint64_t nq, DeviceBuffer<uint32_t> &counts,
                              DeviceBuffer<uint8_t> &status, DeviceBuffer<gdx_hit32_t> &hits, hipStream_t stream) const;

    // ---- host-pointer query API (each call uploads, runs, downloads, synchronises) -------------
    // packed: qbuf holds 2-bit codes and qoff counts symbols (include/gdx.h "packed queries")
    int cursors_for_many_queries(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_start,
                                 uint64_t *out_end, uint64_t *out_count, uint8_t *out_status, bool packed = false,
                                 uint64_t uniform_len = 0) const;
    uint64_t pack_queries_host(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint8_t *out_packed,
                               uint64_t *out_exc, uint64_t capacity) const;
    int locate_many(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets,
                    gdx_hit_t *hits, uint64_t hits_capacity, uint64_t *out_total, uint8_t *out_status) const;
    // the same as locate_many with a hit buffer the library allocates (malloc; the caller frees it): one pass
    int locate_many_alloc(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets,
                          gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status, bool packed = false,
                          uint64_t uniform_len = 0) const;
    // narrow results (u32 offsets, 8-byte hits) in pinned memory the library owns: the device writes them there (host_api.hip)
    int locate_many_alloc32(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, gdx_hits32_t *out, uint8_t *out_status,
                            bool packed = false, uint64_t uniform_len = 0) const;
    int cursor_extend_front_many(uint64_t *start, uint64_t *end, const uint8_t *io_symbols, uint64_t m,
                                 uint8_t *out_status) const;
    int cursor_extend_front_strings(uint64_t *start, uint64_t *end, const uint8_t *qbuf, const uint64_t *qoff, uint64_t m,
                                    uint8_t *status) const;
    int cursor_locate_many(const uint64_t *start, const uint64_t *end, uint64_t m, uint64_t *out_hit_offsets,
                           gdx_hit_t *hits, uint64_t hits_capacity, uint64_t *out_total) const;
    int rank_many(const uint8_t *symbols, const uint64_t *idx, uint64_t m, uint64_t *out) const;
    int symbol_at_many(const uint64_t *idx, uint64_t m, uint8_t *out) const;

    // ---- exports in the reference's logical layout ------------------------------------------------
    void export_count(uint64_t *count) const;
    void export_bwt(uint8_t *bwt) const;
    void export_sa_samples(uint32_t *samples) const;
    void export_borders(uint64_t *keys, uint64_t *vals) const;
    void export_sentinel_indices(uint64_t *out) const;
    void export_lookup_table(int depth, uint32_t *pairs) const;
    void export_condensed_table(uint64_t *blocks, uint16_t *block_offsets, uint32_t *superblock_offsets) const;
    void export_reference_table(uint64_t *blocks, uint64_t capacity_words, uint64_t *n_words, uint32_t *superblock_offsets,
                                uint64_t capacity_sb, uint64_t *n_sb) const;

private:
    FmIndex() = default;
    // chunked, three-deep pipeline behind the host-pointer query calls (host_api.hip)
    int host_pipeline(int kind, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_a, uint64_t *out_b,
                      uint8_t *out_status, gdx_hit_t *hits, uint64_t hits_capacity, uint64_t *out_total,
                      const std::function<gdx_hit_t *(uint64_t, uint64_t *)> *grow_hits, bool packed = false,
                      uint64_t uniform_len = 0, Narrow32Sink *narrow = nullptr) const;
    void finish_from_bwt(const uint8_t *d_bwt_padded, hipStream_t stream);  // table + lookup + view
    void build_aux(const uint8_t *d_bwt_padded, hipStream_t stream);        // pair lines, jump table, top table
    void build_seed_table(const uint32_t *d_sa, uint32_t k, uint32_t load_percent, hipStream_t stream, uint64_t room_bytes);
    void locate_device(const uint32_t *d_start, const uint32_t *d_end, uint64_t m, uint64_t *out_hit_offsets,
                       gdx_hit_t *hits, uint64_t hits_capacity, uint64_t *out_total, int *rc,
                       const uint4 *d_rec = nullptr) const;

    IndexConfig cfg_;
    BuildStats stats_;
    AuxReport aux_report_;
    std::atomic<int> q_search_variant_{-1}, q_search_lanes_{0}, q_load_policy_{-1}, q_schedule_{-1}, q_locate_variant_{-1}, q_locate_jump_walk_{-1}, q_defer_after_{-1}, q_fast_{-1}, q_exact_{-1}, q_seed_{-1};
    std::atomic<uint32_t> q_max_hits_{0};
    uint64_t n_ = 0, n_texts_ = 0;
    std::vector<uint64_t> count_host_;      // sigma+1
    std::vector<uint64_t> sentinels_host_;  // n_texts
    std::vector<uint64_t> border_keys_host_, border_vals_host_;
    std::vector<uint64_t> lookup_off_host_;

    DeviceBuffer<u32x4> lines_;
    DeviceBuffer<uint32_t> sb_offsets_;
    DeviceBuffer<u32x4> pair_lines_;
    DeviceBuffer<uint32_t> jump_;
    DeviceBuffer<uint2> top_;
    DeviceBuffer<uint32_t> sa_full_;
    DeviceBuffer<u32x4> text_units_;
    DeviceBuffer<u32x4> seed_, seed_pairs_, seed_quads_;
    DeviceBuffer<uint32_t> isa_;
    DeviceBuffer<uint64_t> g_planes_;
    DeviceBuffer<uint16_t> g_block_off_;
    DeviceBuffer<uint32_t> count_;
    DeviceBuffer<uint8_t> io_to_dense_;
    DeviceBuffer<uint32_t> sa_samples_;
    DeviceBuffer<uint32_t> border_keys_, border_vals_, sentinels_;
    DeviceBuffer<uint2> lookup_;
    IndexView view_{};
};

// ---- multi.hip: replicas on several devices behind one handle (gdx_multi_*) ---------------------------------
// One long-lived worker thread per replica: the host-pointer pipeline keeps its pinned staging and device buffers in
// per-thread caches (host_api.hip), so the threads have to outlive the calls.
class ReplicaWorker {
public:
    ReplicaWorker();
    ~ReplicaWorker();
    ReplicaWorker(const ReplicaWorker &) = delete;
    ReplicaWorker &operator=(const ReplicaWorker &) = delete;
    void submit(std::function<void()> job);  // runs on the worker thread
    void wait();                             // until every submitted job is done

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

struct Multi {
    std::vector<std::unique_ptr<FmIndex>> replicas;
    std::vector<std::unique_ptr<ReplicaWorker>> workers;  // one per replica (start_workers)
    std::recursive_mutex call_mutex;  // calls on one handle run one after the other: a second concurrent call queues here
    // gdx_multi_locate_many_gather_dev: RCCL communicators (one per replica, created on first use) and the root's
    // result buffers
    std::vector<void *> comms;
    DeviceBuffer<uint32_t> g_counts;
    DeviceBuffer<uint64_t> g_offsets;
    DeviceBuffer<gdx_hit32_t> g_hits;
    DeviceBuffer<uint8_t> g_status, g_scan;
    int g_device = -1;
    ~Multi();
    void start_workers();   // once, when the replicas are in place
    ReplicaWorker &worker(size_t r) { return *workers[r]; }
};
int multi_cursors(Multi &m, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_start,
                  uint64_t *out_end, uint64_t *out_count, uint8_t *out_status);
struct DeviceShard {
    const uint8_t *d_qbuf;
    const uint64_t *d_qoff;
    uint64_t nq;
};
struct Gathered {
    void *d_counts, *d_hit_offsets, *d_hits, *d_status;
    uint64_t nq, total_hits;
    int device_id, used_rccl;
};
void multi_locate_gather_dev(Multi &m, const DeviceShard *shards, int n_shards, int root, Gathered *out);
// FmIndex half of it: search -> scan -> locate of a device-resident shard on this replica's device (multi.hip)
int multi_locate_alloc(Multi &m, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets,
                       gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status);

void launch_build_lines(const uint8_t *d_bwt_padded, uint64_t n, uint64_t n_lines, u32x4 *d_lines, uint32_t *d_sb_totals,
                        uint64_t n_sb, hipStream_t stream);

// Index storage beyond 32 bits (wide.hip): rows and text positions are 64-bit; the reference's own arrays only (rank lines,
// u64 superblock offsets, sampled suffix array, borders) and one-lane-per-query kernels -- the functional equivalent
// of `IndexStorage for i64` (construction/mod.rs:225-252), not a fast path.
class WideIndex {
public:
    static std::unique_ptr<WideIndex> construct_index(const uint8_t *texts_buf, bool texts_on_device,
                                                      const uint64_t *text_offsets, uint64_t n_texts, const IndexConfig &cfg);
    ~WideIndex();
    const IndexConfig &config() const;
    uint64_t total_text_len() const;
    uint64_t num_texts() const;
    uint64_t device_bytes() const;
    int cursors_for_many_queries(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_start, uint64_t *out_end,
                                 uint64_t *out_count, uint8_t *out_status) const;
    int locate_many(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets, gdx_hit_t *hits,
                    uint64_t hits_capacity, uint64_t *out_total, uint8_t *out_status) const;
    int locate_many_alloc(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets,
                          gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status) const;
    int cursor_extend_front_many(uint64_t *start, uint64_t *end, const uint8_t *io_symbols, uint64_t m, uint8_t *out_status) const;
    int cursor_locate_many(const uint64_t *start, const uint64_t *end, uint64_t m, uint64_t *out_hit_offsets, gdx_hit_t *hits,
                           uint64_t hits_capacity, uint64_t *out_total) const;
    // rank(symbols[i], idx[i]) (symbols != null) or symbol_at(idx[i]) as u64 (symbols == null)
    int rank_or_symbol_many(const uint8_t *symbols, const uint64_t *idx, uint64_t m, uint64_t *out) const;
    void export_bwt(uint8_t *bwt) const;

private:
    WideIndex();
    struct Impl;
    std::unique_ptr<Impl> p_;
};

// A collection beyond 2^32 - 1 symbols as several 32-bit indexes cut at text borders (parts.hip, gdx_parts_*)
struct Parts {
    std::vector<std::unique_ptr<FmIndex>> parts;
    std::vector<uint64_t> first_text;  // [parts + 1]: number of texts before every part
    uint64_t total_len = 0;            // symbols incl. sentinels
    IndexConfig cfg;
    ~Parts();
    static std::unique_ptr<Parts> build(const uint8_t *texts_buf, bool texts_on_device, const uint64_t *text_offsets,
                                        uint64_t n_texts, const IndexConfig &cfg, uint64_t max_part_symbols);
    int count_many(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_counts, uint8_t *out_status) const;
    int locate_many_alloc(const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq, uint64_t *out_hit_offsets,
                          gdx_hit_t **out_hits, uint64_t *out_total, uint8_t *out_status) const;
};

// ASCII -> 2-bit codes on the host with nothing but the alphabet's table (gdx_pack_queries_table; host_api.hip)
uint64_t pack_queries_with_table(const uint8_t *io_to_dense, const uint8_t *qbuf, const uint64_t *qoff, uint64_t nq,
                                 uint8_t *out_packed, uint64_t *out_exc, uint64_t capacity);
// gdx_free_hits: the array goes back to the library, which keeps one for the next gdx_locate_many_alloc (host_api.hip)
void recycle_hits(gdx_hit_t *hits);
void recycle_hits32(gdx_hits32_t *results);  // gdx_free_hits32: the pinned arrays go back to the library
void release_cached_hits();                  // gdx_release_cached_hits
// where the narrow host locate (FmIndex::locate_many_alloc32) puts its results: pinned arrays the device copies into
struct Narrow32Sink {
    uint32_t *offsets = nullptr;  // nq + 1
    gdx_hit32_t *hits = nullptr;
    uint64_t cap = 0;             // hits the array holds
    // a larger array holding the first `keep` hits of the old one (called with no copy into the old one on its way)
    std::function<gdx_hit32_t *(uint64_t need, uint64_t keep, uint64_t *new_cap)> grow;
};
// parser threads of the mapped FASTA / FASTQ reader (fastx.hpp): the CPUs the process may use, at most 32 (host_api.hip)
unsigned fastx_default_threads();
// chunk size of the host-pointer pipeline (host_api.hip); 0 = default.  Tests use small chunks.
void set_host_chunking(uint64_t queries, uint64_t bytes);

}  // namespace gdx
