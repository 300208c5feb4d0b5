// host_checks.cpp -- the parts of libgdx.so's host code that parse untrusted bytes, compiled for the CPU with
// AddressSanitizer + UBSan (tests/test_host_sanitized.py builds and runs this; GPU sanitizers are not available):
//   host_checks fastx <file> <max_records> <buffer_bytes>   reads every batch; prints "ok <records> <symbols> <checksum>"
//   host_checks header <file>                                validates an index file header; prints "ok n=.. texts=.."
//   host_checks pack <file>                                  packs the file's bytes as queries (pack_host.hpp: the AVX2 path and the
//                                                            byte loop) in exactly sized buffers; prints "ok <cases> <exceptions>"
// A malformed input must end in "error: <message>" (exit code 3), never in a sanitizer report.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "../../genedex_amd/csrc/fastx.hpp"
#include "../../genedex_amd/csrc/index_file.hpp"
#include "../../genedex_amd/csrc/pack_host.hpp"

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
        } else if (mode == "pack") {
            // the file's bytes as a query buffer, packed in pieces whose borders fall everywhere relative to the 32- and
            // 128-symbol steps of the vector loop; source and destination are heap blocks of exactly the size the contract
            // names, so a read or write one byte outside is a sanitizer report.  Compared with the table applied byte by byte.
            std::vector<uint8_t> data;
            {
                gdx::IndexFile in(argv[2], "rb");
                uint8_t chunk[4096];
                for (;;) {
                    const size_t n = std::fread(chunk, 1, sizeof(chunk), in.f);
                    if (n == 0) break;
                    data.insert(data.end(), chunk, chunk + n);
                }
            }
            uint8_t dna[256] = {0}, clash[256] = {0};
            dna[(int)'A'] = dna[(int)'a'] = 1, dna[(int)'C'] = dna[(int)'c'] = 2, dna[(int)'G'] = dna[(int)'g'] = 3;
            dna[(int)'T'] = dna[(int)'t'] = 4, dna[(int)'N'] = dna[(int)'n'] = 5;
            clash[0x41] = 1, clash[0x51] = 2, clash[0x43] = 3, clash[0x47] = 4;  // no nibble plan: the byte loop
            uint64_t cases = 0, exceptions = 0;
            for (const uint8_t *tab : {dna, clash}) {
                const gdx::PackPlan plan = gdx::make_pack_plan(tab);
                if ((tab == dna) != plan.fast) return 5;
                for (uint64_t first : {uint64_t(0), uint64_t(1), uint64_t(31), uint64_t(33), uint64_t(130)}) {
                    for (uint64_t cut : {uint64_t(0), uint64_t(1), uint64_t(29), uint64_t(127)}) {
                        if (first + cut > data.size()) continue;
                        const uint64_t n_sym = data.size() - cut;
                        if (first > n_sym) continue;
                        const uint64_t n_bytes = (n_sym + 3) / 4;
                        std::unique_ptr<uint8_t[]> src(new uint8_t[n_sym ? n_sym : 1]), out(new uint8_t[n_bytes ? n_bytes : 1]);
                        std::copy(data.begin(), data.begin() + n_sym, src.get());
                        // two halves on a 64-byte border of the output, as the worker threads split it
                        const uint64_t mid = std::min(n_bytes, (n_bytes / 2 + 63) / 64 * 64);
                        std::vector<uint64_t> bad;
                        gdx::pack_range(plan, tab, src.get(), first, n_sym, 0, mid, out.get(), [&](uint64_t j) { bad.push_back(j); });
                        gdx::pack_range(plan, tab, src.get(), first, n_sym, mid, n_bytes, out.get(), [&](uint64_t j) { bad.push_back(j); });
                        std::vector<uint64_t> want_bad;
                        for (uint64_t b = 0; b < n_bytes; b++) {
                            uint32_t w = 0;
                            for (uint32_t k = 0; k < 4; k++) {
                                const uint64_t j = 4 * b + k;
                                if (j < first || j >= n_sym) continue;
                                const uint32_t d = tab[src[j]];
                                if (d - 1u < 4u) w |= (d - 1u) << (2u * k);
                                else want_bad.push_back(j);
                            }
                            if (out[b] != w) return 6;
                        }
                        if (bad != want_bad) return 7;
                        cases++;
                        exceptions += bad.size();
                    }
                }
            }
            std::printf("ok %" PRIu64 " %" PRIu64 "\n", cases, exceptions);
        } else {
            return 2;
        }
    } catch (const gdx::Error &e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
    return 0;
}
