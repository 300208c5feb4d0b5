// synth.hpp -- synthetic workloads and roofline micro-benchmarks (include/gdx_bench.h)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace gdx {

void launch_synth_text(uint8_t *d_out, uint64_t n, uint64_t seed, uint32_t n_per_million, hipStream_t stream);
void synth_queries(const uint8_t *d_io_text, const uint64_t *d_text_offsets, uint64_t n_texts, uint64_t nq,
                   uint32_t len_min, uint32_t len_max, uint32_t sampled_per_million, uint64_t seed, uint64_t *d_qoff,
                   uint8_t *d_qbuf, uint64_t qbuf_capacity, uint64_t *out_total_bytes, hipStream_t stream);
void launch_stream_copy(void *d_dst, const void *d_src, uint64_t bytes, hipStream_t stream);
void launch_stream_read(const void *d_src, uint64_t bytes, uint32_t *d_sink, hipStream_t stream);
void launch_random_gather(const void *d_src, uint64_t n_lines, uint32_t line_bytes, uint64_t n_accesses,
                          uint64_t seed, uint32_t mode, uint32_t *d_sink, hipStream_t stream);

}  // namespace gdx
