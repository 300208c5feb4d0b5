// host_checks.cpp -- the parts of libgdx.so's host code that parse untrusted bytes, compiled for the CPU with
// AddressSanitizer + UBSan (tests/test_host_sanitized.py builds and runs this; GPU sanitizers are not available):
//   host_checks fastx <file> <max_records> <buffer_bytes>   reads every batch; prints "ok <records> <symbols> <checksum>"
//   host_checks header <file>                                validates an index file header; prints "ok n=.. texts=.."
// A malformed input must end in "error: <message>" (exit code 3), never in a sanitizer report.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../genedex_amd/csrc/fastx.hpp"
#include "../../genedex_amd/csrc/index_file.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: host_checks fastx <file> <max_records> <buffer_bytes> | header <file>\n");
        return 2;
    }
    const std::string mode = argv[1];
    try {
        if (mode == "fastx") {
            const uint64_t max_records = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1000;
            const uint64_t cap = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : (1u << 20);
            gdx::FastxReader reader(argv[2]);
            std::vector<uint8_t> qbuf(cap ? cap : 1);
            std::vector<uint64_t> qoff(max_records + 1);
            uint64_t records = 0, symbols = 0, checksum = 1469598103934665603ull;
            for (;;) {
                const uint64_t n = reader.next_batch(qbuf.data(), cap, qoff.data(), max_records);
                if (n == 0) break;
                records += n;
                symbols += qoff[n];
                for (uint64_t i = 0; i < qoff[n]; i++) checksum = (checksum ^ qbuf[i]) * 1099511628211ull;
                for (uint64_t i = 0; i < n; i++)
                    if (qoff[i + 1] < qoff[i] || qoff[i + 1] > cap) return 4;  // the offsets stay inside the buffer
            }
            std::printf("ok %" PRIu64 " %" PRIu64 " %" PRIu64 "\n", records, symbols, checksum);
        } else if (mode == "header") {
            gdx::IndexFile in(argv[2], "rb");
            const gdx::FileHeader h = gdx::read_index_header(in, argv[2]);
            std::printf("ok n=%" PRIu64 " texts=%" PRIu64 " sigma=%d\n", h.n, h.n_texts, h.sigma);
        } else {
            return 2;
        }
    } catch (const gdx::Error &e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
    return 0;
}
