// index_file.hpp -- header of the index file format of gdx_index_save / gdx_index_load (FmIndex::save_to_file /
// load_from_file, lib.rs:296-327) and its validation.  No HIP: the same code is compiled into the sanitised CPU
// checks (tests/host_checks.cpp), because the header comes from an untrusted file.
//
// File = FileHeader, then count u64[sigma + 1], sentinel_indices u64[n_texts], border keys u64[n_texts], border
// values u64[n_texts], suffix-array samples u32[n_samples], bit planes u64[n_plane_words] (condensed.rs:24-30).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "errors.hpp"

namespace gdx {

constexpr char kIndexFileMagic[8] = {'G', 'D', 'X', 'I', 'D', 'X', '0', '1'};

struct FileHeader {
    char magic[8];
    uint64_t n, n_texts, sa_rate, n_plane_words, n_samples;
    int32_t sigma, n_searchable, lookup_depth, index_width;
    uint8_t io_to_dense[256];
};

inline int plane_bits(uint64_t sigma)  // condensed.rs:417-419
{
    int bits = 0;
    while ((1ull << bits) < sigma) bits++;
    return bits;
}

struct IndexFile {
    FILE *f;
    IndexFile(const char *path, const char *mode) : f(std::fopen(path, mode))
    {
        if (!f) fail(GDX_ERR_INVALID_ARGUMENT, "cannot open %s", path);
    }
    ~IndexFile()
    {
        if (f) std::fclose(f);
    }
    IndexFile(const IndexFile &) = delete;
    IndexFile &operator=(const IndexFile &) = delete;
    void write(const void *p, size_t bytes)
    {
        if (bytes && std::fwrite(p, 1, bytes, f) != bytes) fail(GDX_ERR_DEVICE, "short write");
    }
    void read(void *p, size_t bytes)
    {
        if (bytes && std::fread(p, 1, bytes, f) != bytes) fail(GDX_ERR_INVALID_ARGUMENT, "index file is truncated");
    }
};

// Reads the header and checks everything that later sizes an allocation or indexes an array: the magic, the scalar
// ranges, that the section sizes follow from n / sigma / sa_rate, and that the file holds exactly the payload the
// header promises (before anything is allocated).
inline FileHeader read_index_header(IndexFile &in, const char *path)
{
    FileHeader h;
    in.read(&h, sizeof(h));
    if (std::memcmp(h.magic, kIndexFileMagic, 8) != 0) fail(GDX_ERR_INVALID_ARGUMENT, "%s is not a gdx index file", path);
    if (h.sigma < 2 || h.sigma > 256 || h.n_searchable < 1 || h.n_searchable >= h.sigma || h.n_texts == 0 ||
        h.n_texts > h.n || h.n > 0xffffffffull || h.sa_rate == 0 || h.sa_rate > 0xffffffffull || h.lookup_depth < 0 ||
        h.lookup_depth > 24 || (h.index_width != 32 && h.index_width != -32 && h.index_width != 64) ||
        h.n_samples != div_ceil_u64(h.n, h.sa_rate) ||
        h.n_plane_words != div_ceil_u64(h.n + 1, 64) * static_cast<uint64_t>(plane_bits(static_cast<uint64_t>(h.sigma))))
        fail(GDX_ERR_INVALID_ARGUMENT, "index file header is inconsistent");
    for (int b = 0; b < 256; b++)  // a dense code indexes count[] and the superblock offsets at query time
        if (h.io_to_dense[b] >= h.sigma) fail(GDX_ERR_INVALID_ARGUMENT, "index file header: io_to_dense[%d] is not a dense symbol", b);
    const uint64_t payload = (static_cast<uint64_t>(h.sigma) + 1 + 3 * h.n_texts + h.n_plane_words) * sizeof(uint64_t) +
                             h.n_samples * sizeof(uint32_t);
    const long at = std::ftell(in.f);
    if (at < 0 || std::fseek(in.f, 0, SEEK_END) != 0) fail(GDX_ERR_INVALID_ARGUMENT, "cannot seek in %s", path);
    const long size = std::ftell(in.f);
    if (size < 0 || std::fseek(in.f, at, SEEK_SET) != 0) fail(GDX_ERR_INVALID_ARGUMENT, "cannot seek in %s", path);
    if (static_cast<uint64_t>(size - at) != payload)
        fail(GDX_ERR_INVALID_ARGUMENT, "index file is truncated or has trailing bytes (%lld payload bytes, header says %llu)",
             static_cast<long long>(size - at), static_cast<unsigned long long>(payload));
    return h;
}

}  // namespace gdx
