// host_checks.cpp -- the parts of libgdx.so's host code that parse untrusted bytes, compiled for the CPU with
// AddressSanitizer + UBSan (tests/test_host_sanitized.py builds and runs this; GPU sanitizers are not available):
//   host_checks fastx <file> <max_records> <buffer_bytes>   reads every batch; prints "ok <records> <symbols> <checksum>"
//   host_checks header <file>                                validates an index file header; prints "ok n=.. texts=.."
//   host_checks pack <file>                                  packs the file's bytes as queries (pack_host.hpp: the AVX2 path and the
//                                                            byte loop) in exactly sized buffers; prints "ok <cases> <exceptions>"
//   host_checks wire <seed> <reads> <texts> [<found of ten> <exceptions: one in>]  expands a random chunk's "found bitmap" wire (wire_host.hpp) in pieces,
//                                                            exactly sized buffers; prints "ok <hits> <exceptions>"
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
#include "../../genedex_amd/csrc/wire_host.hpp"

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
        } else if (mode == "fastxmap") {
            // the memory-mapped reader, blocks parsed by `threads` threads (GDX_FASTX_BLOCK_BYTES: the smallest block): the same
            // line as "fastx" prints for the same file, or the same error
            const uint64_t max_records = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1000;
            const uint64_t cap = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : (1u << 20);
            const unsigned threads = argc > 5 ? static_cast<unsigned>(std::strtoul(argv[5], nullptr, 10)) : 4u;
            std::unique_ptr<gdx::FastxMappedReader> reader(gdx::FastxMappedReader::open(argv[2], threads));
            if (!reader) {  // (an empty or missing file: the streaming reader decides)
                gdx::FastxReader plain(argv[2]);
                std::printf("ok 0 0 %" PRIu64 "\n", static_cast<uint64_t>(1469598103934665603ull));
                return 0;
            }
            std::vector<uint8_t> qbuf(cap ? cap : 1);
            std::vector<uint64_t> qoff(max_records + 1);
            uint64_t records = 0, symbols = 0, checksum = 1469598103934665603ull;
            for (;;) {
                uint64_t ulen = 0;
                const uint64_t n = reader->next_batch(qbuf.data(), cap, qoff.data(), max_records, &ulen);
                if (n == 0) break;
                records += n;
                symbols += qoff[n];
                for (uint64_t i = 0; i < qoff[n]; i++) checksum = (checksum ^ qbuf[i]) * 1099511628211ull;
                for (uint64_t i = 0; i < n; i++) {
                    if (qoff[i + 1] < qoff[i] || qoff[i + 1] > cap) return 4;
                    if (ulen != 0 && qoff[i + 1] - qoff[i] != ulen) return 5;
                }
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
        } else if (mode == "wire") {
            // a chunk of reads with random outcomes -- one hit (nine in ten, or one in ten: argv[5]), none, or an exception with
            // 0..40 hits -- over a collection of `texts` texts; the wire as wire_pack_kernel lays it out, in heap blocks of
            // exactly the sizes the host pipeline copies; expanded by 1, 3 and 7 workers and compared with the plain construction
            uint64_t x = std::strtoull(argv[2], nullptr, 10) * 2654435761ull + 88172645463325252ull;
            auto rnd = [&]() {
                x ^= x << 13, x ^= x >> 7, x ^= x << 17;
                return x;
            };
            const uint64_t nq = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 5000;
            const uint64_t n_texts = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 3;
            const uint64_t found_in_ten = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 9;
            const uint64_t exc_one_in = argc > 6 ? std::strtoull(argv[6], nullptr, 10) : 3;  // of the reads that are not found
            const uint64_t n = 1000 + n_texts * 700;
            std::vector<uint64_t> sentinels(n_texts);
            for (uint64_t t = 0; t < n_texts; t++) sentinels[t] = (t + 1) * (n / n_texts) - 1 - (t + 1 < n_texts ? rnd() % 300 : 0);
            sentinels[n_texts - 1] = n - 1;
            auto split = [&](uint32_t g) {
                uint32_t t = 0;
                while (sentinels[t] < g) t++;
                gdx_hit32_t h;
                h.text_id = t;
                h.position = t == 0 ? g : g - static_cast<uint32_t>(sentinels[t - 1]) - 1u;
                return h;
            };
            const uint64_t tiles = (nq + gdx::kHostWireTile - 1) / gdx::kHostWireTile;
            std::vector<uint8_t> bitmap(tiles * 256, 0);
            std::vector<uint32_t> tile_found(tiles + 1), tile_off(tiles + 1), found_pos, exc_q, exc_cnt, want_off(nq + 1);
            std::vector<uint8_t> found_ids;
            std::vector<gdx_hit32_t> exc_hits, want_hits;
            const uint32_t base = 777;
            want_off[0] = base;
            for (uint64_t q = 0; q < nq; q++) {
                if (q % gdx::kHostWireTile == 0) {
                    tile_found[q / gdx::kHostWireTile] = static_cast<uint32_t>(found_pos.size());
                    tile_off[q / gdx::kHostWireTile] = static_cast<uint32_t>(want_hits.size());
                }
                const uint64_t r = rnd() % 100;
                if (r < found_in_ten * 10) {
                    uint32_t g;
                    do g = static_cast<uint32_t>(rnd() % n); while (std::find(sentinels.begin(), sentinels.end(), g) != sentinels.end());
                    bitmap[q / 8] |= static_cast<uint8_t>(1u << (q % 8));
                    found_pos.push_back(split(g).position);
                    found_ids.push_back(static_cast<uint8_t>(split(g).text_id));
                    want_hits.push_back(split(g));
                } else if (rnd() % exc_one_in == 0) {
                    const uint32_t cnt = static_cast<uint32_t>(rnd() % 41);
                    exc_q.push_back(static_cast<uint32_t>(q));
                    exc_cnt.push_back(cnt);
                    for (uint32_t i = 0; i < cnt; i++) {
                        gdx_hit32_t h;
                        h.text_id = static_cast<uint32_t>(rnd() % n_texts), h.position = static_cast<uint32_t>(rnd() % 1000);
                        exc_hits.push_back(h);
                        want_hits.push_back(h);
                    }
                }
                want_off[q + 1] = base + static_cast<uint32_t>(want_hits.size());
            }
            tile_found[tiles] = static_cast<uint32_t>(found_pos.size());
            tile_off[tiles] = static_cast<uint32_t>(want_hits.size());
            found_pos.push_back(0xffffffffu);  // (the one element past the found reads that HostWire asks to be readable)
            found_ids.push_back(0xffu);
            auto exact = [](const auto &v) {  // a heap block of exactly the vector's bytes (at least one)
                using T = typename std::decay<decltype(v[0])>::type;
                std::unique_ptr<T[]> p(new T[v.size() ? v.size() : 1]);
                std::copy(v.begin(), v.end(), p.get());
                return p;
            };
            const auto b_bitmap = exact(bitmap);
            const auto b_tf = exact(tile_found), b_to = exact(tile_off), b_fp = exact(found_pos), b_eq = exact(exc_q), b_ec = exact(exc_cnt);
            const auto b_eh = exact(exc_hits);
            const auto b_fi = exact(found_ids);
            const gdx::HostWire w{b_bitmap.get(), b_tf.get(), b_to.get(), b_fp.get(), n_texts > 1 ? b_fi.get() : nullptr,
                                  b_eq.get(), b_ec.get(), b_eh.get(), exc_q.size()};
            for (uint64_t workers : {uint64_t(1), uint64_t(3), uint64_t(7)}) {
                const uint64_t per = (tiles + workers - 1) / workers;
                // the narrow form (u32 offsets, 8-byte hits) ...
                std::unique_ptr<uint32_t[]> off(new uint32_t[nq + 1]);
                std::unique_ptr<gdx_hit32_t[]> hits(new gdx_hit32_t[base + want_hits.size() + 1]);
                std::fill(off.get(), off.get() + nq + 1, 0xdeadbeefu);
                for (uint64_t k = workers; k-- > 0;)  // (in reverse: no piece relies on the one before it)
                    gdx::wire_expand_tiles<uint32_t, gdx_hit32_t>(w, nq, std::min(tiles, per * k), std::min(tiles, per * k + per), base, off.get(),
                                                                  hits.get());
                for (uint64_t q = 0; q <= nq; q++)
                    if (off[q] != want_off[q] && nq != 0) return 8;
                for (uint64_t i = 0; i < want_hits.size(); i++)
                    if (hits[base + i].text_id != want_hits[i].text_id || hits[base + i].position != want_hits[i].position) return 9;
                // ... the wide form (u64 offsets, 16-byte hits), and offsets alone beyond 2^32 (the sizing pass of gdx_locate_many)
                std::unique_ptr<uint64_t[]> off64(new uint64_t[nq + 1]), off_only(new uint64_t[nq + 1]);
                std::unique_ptr<gdx_hit_t[]> hits64(new gdx_hit_t[base + want_hits.size() + 1]);
                const uint64_t far = 5'000'000'000ull;
                for (uint64_t k = workers; k-- > 0;) {
                    gdx::wire_expand_tiles<uint64_t, gdx_hit_t>(w, nq, std::min(tiles, per * k), std::min(tiles, per * k + per), base, off64.get(),
                                                                hits64.get());
                    gdx::wire_expand_tiles<uint64_t, gdx_hit_t>(w, nq, std::min(tiles, per * k), std::min(tiles, per * k + per), far, off_only.get(),
                                                                nullptr);
                    gdx::wire_expand_tiles<uint64_t, gdx_hit_t>(w, nq, std::min(tiles, per * k), std::min(tiles, per * k + per), base, nullptr,
                                                                hits64.get());
                }
                for (uint64_t q = 0; q <= nq; q++)
                    if (nq != 0 && (off64[q] != want_off[q] || off_only[q] != want_off[q] - base + far)) return 10;
                for (uint64_t i = 0; i < want_hits.size(); i++)
                    if (hits64[base + i].text_id != want_hits[i].text_id || hits64[base + i].position != want_hits[i].position) return 11;
            }
            // the staging copy and the widening of counts, on exactly sized blocks at every alignment of source and destination
            for (uint64_t len : {uint64_t(0), uint64_t(1), uint64_t(31), uint64_t(4095), uint64_t(4096), uint64_t(4229), uint64_t(10000)})
                for (uint64_t shift = 0; shift < 40; shift += 13) {
                    std::unique_ptr<uint8_t[]> src(new uint8_t[len + shift + 1]), dst(new uint8_t[len + 40 - shift + 1]);
                    for (uint64_t i = 0; i < len; i++) src[shift + i] = static_cast<uint8_t>(rnd());
                    gdx::stream_copy(dst.get() + (40 - shift), src.get() + shift, len);
                    if (len != 0 && std::memcmp(dst.get() + (40 - shift), src.get() + shift, len) != 0) return 12;
                    const uint64_t n32 = len / 4;
                    std::unique_ptr<uint64_t[]> wide(new uint64_t[n32 + 1]);
                    std::unique_ptr<uint32_t[]> narrow(new uint32_t[n32 + 1]);
                    for (uint64_t i = 0; i < n32; i++) narrow[i] = static_cast<uint32_t>(rnd());
                    gdx::widen_u32(narrow.get() + (shift & 1), n32 - (n32 ? (shift & 1) : 0), wide.get() + (shift & 1));
                    for (uint64_t i = (shift & 1); i < n32; i++)
                        if (wide[i] != narrow[i]) return 13;
                }
            std::printf("ok %zu %zu\n", want_hits.size(), exc_q.size());
        } else {
            return 2;
        }
    } catch (const gdx::Error &e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
    return 0;
}
