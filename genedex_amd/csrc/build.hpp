// build.hpp -- index construction on the GPU (plumbing for the query path)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace gdx {

struct BuildStats {
    uint64_t sa_initial_order = 0;       // symbols fixed by the first key sort
    uint64_t sa_pending_after_sort = 0;  // suffixes still in groups of size > 1 after it
    uint64_t sa_rounds = 0;              // doubling rounds that followed
    double seconds_encode = 0, seconds_sa = 0, seconds_bwt = 0, seconds_table = 0, seconds_lookup = 0, seconds_pairs = 0;
};

// d_sa[0..n) = suffix array of d_text[0..n) (symbols < sigma); freq[c] = occurrences of symbol c
void build_suffix_array(const uint8_t *d_text, uint64_t n, int sigma, const std::vector<uint64_t> &freq,
                        uint32_t *d_sa, hipStream_t stream, BuildStats *stats);

}  // namespace gdx
