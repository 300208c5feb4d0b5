// fastx_reader_bench -- the mapped FASTA / FASTQ reader alone (no GPU): reads per second over a file, best of `reps`.
//   g++ -std=c++17 -O3 -pthread -I genedex_amd/csrc -I include -o /tmp/fastx_reader_bench tools/fastx_reader_bench.cpp
//   /tmp/fastx_reader_bench <file> <records per batch> <threads> [symbols per record = 50] [reps = 4]
// GDX_FASTX_TIMING=1 prints where every batch's time went.
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>
#ifndef FASTX_HEADER
#define FASTX_HEADER "fastx.hpp"
#endif
#include FASTX_HEADER

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: fastx_reader_bench <file> <records per batch> <threads> [symbols per record] [reps]\n");
        return 2;
    }
    const uint64_t max_records = std::strtoull(argv[2], nullptr, 10);
    const unsigned threads = static_cast<unsigned>(std::atoi(argv[3]));
    const uint64_t cap = max_records * (argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 50);
    const int reps = argc > 5 ? std::atoi(argv[5]) : 4;
    std::vector<uint8_t> qbuf(cap);
    std::vector<uint64_t> qoff(max_records + 1);
    std::memset(qbuf.data(), 1, cap);  // (the buffers' pages exist before anything is timed, as a caller's reused buffers do)
    std::memset(qoff.data(), 1, qoff.size() * 8);
    double best = 0.0;
    for (int rep = 0; rep < reps; rep++) {
        const double t0 = now();
        std::unique_ptr<gdx::FastxMappedReader> reader(gdx::FastxMappedReader::open(argv[1], threads));
        if (!reader) return 3;
        uint64_t records = 0, checksum = 0;
        for (;;) {
            uint64_t ulen = 0;
            const uint64_t n = reader->next_batch(qbuf.data(), cap, qoff.data(), max_records, &ulen);
            if (n == 0) break;
            records += n;
            checksum += qoff[n] + qbuf[qoff[n] - 1];
        }
        const double dt = now() - t0;
        std::printf("rep %d: %" PRIu64 " records in %.1f ms -> %.1f M records/s (checksum %" PRIu64 ")\n", rep, records, dt * 1e3,
                    static_cast<double>(records) / dt / 1e6, checksum);
        best = std::max(best, static_cast<double>(records) / dt / 1e6);
    }
    std::printf("best %.1f M records/s on %u threads\n", best, threads);
    return 0;
}
