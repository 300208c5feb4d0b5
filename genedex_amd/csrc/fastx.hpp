// fastx.hpp -- streaming FASTA / FASTQ reader that fills the ABI's query layout (qbuf + qoff) batch by batch.
// Host-only.  SURVEY.md section 8f row 3: the reference leaves reading to its callers and notes that it can
// cost more than searching (ROADMAP.md:35-37); this is the ingestion side of gdx_count_many / gdx_locate_many
// and of gdx_index_build (texts).
//
// Accepted: FASTA ('>' header lines, sequences over any number of lines) and FASTQ ('@' header, sequence lines,
// '+' line, as many quality characters as sequence symbols, also over several lines).  '\r' is dropped, empty
// lines are skipped, sequence bytes are copied as they are (the alphabet table decides what is valid).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string>
#include <vector>

#include "errors.hpp"

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define GDX_FASTX_AVX2 1
#endif

namespace gdx {

class FastxReader {
public:
    explicit FastxReader(const char *path) : file_(std::fopen(path, "rb")), buf_(1 << 20)
    {
        if (!file_) fail(GDX_ERR_INVALID_ARGUMENT, "cannot open %s", path);
    }
    ~FastxReader()
    {
        if (file_) std::fclose(file_);
    }
    FastxReader(const FastxReader &) = delete;
    FastxReader &operator=(const FastxReader &) = delete;

    // Appends up to max_records sequences to qbuf (capacity bytes) and their offsets to qoff[0 .. n] (qoff[0] = 0).
    // Stops before a record that does not fit (a record longer than the whole buffer is an error).  Returns the
    // number of records read; 0 at the end of the file.
    uint64_t next_batch(uint8_t *qbuf, uint64_t capacity, uint64_t *qoff, uint64_t max_records)
    {
        uint64_t n = 0, used = 0;
        qoff[0] = 0;
        while (n < max_records) {
            // a FASTQ record whose four lines lie in the read buffer as they are goes straight into qbuf (no line strings, no
            // pending copy): the reader ran at a few million reads per second, far below everything behind it
            if (!have_pending_ && !have_line_) {
                const size_t got = fastq_record_in_place(qbuf + used, capacity - used);
                if (got != kNoFastPath) {
                    used += got;
                    qoff[++n] = used;
                    records_++;
                    continue;
                }
            }
            if (!have_pending_ && !read_record()) break;
            if (pending_.size() > capacity - used) {
                if (n == 0) fail(GDX_ERR_CAPACITY, "record %llu has %zu symbols, the buffer holds %llu",
                                 static_cast<unsigned long long>(records_), pending_.size(),
                                 static_cast<unsigned long long>(capacity));
                break;  // stays pending for the next batch
            }
            if (!pending_.empty()) std::memcpy(qbuf + used, pending_.data(), pending_.size());
            used += pending_.size();
            qoff[++n] = used;
            have_pending_ = false;
            records_++;
        }
        return n;
    }

private:
    static constexpr size_t kNoFastPath = ~static_cast<size_t>(0);
    // The next record if it is a plain four-line FASTQ record that lies completely in the buffer: '@' header, ONE sequence
    // line, '+' line, ONE quality line of the same length, no '\r', all four newlines present.  Copies the sequence to out
    // (if it fits `room`) and returns its length; kNoFastPath = anything else (the general path decides: nothing is consumed).
    size_t fastq_record_in_place(uint8_t *out, uint64_t room)
    {
        if (pos_ >= len_ || buf_[pos_] != '@') return kNoFastPath;
        const char *p = buf_.data() + pos_, *end = buf_.data() + len_;
        const char *h = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        if (!h) return kNoFastPath;
        const char *seq = h + 1;
        const char *s_end = seq < end ? static_cast<const char *>(std::memchr(seq, '\n', static_cast<size_t>(end - seq))) : nullptr;
        if (!s_end || s_end == seq || s_end[-1] == '\r' || s_end + 1 >= end || s_end[1] != '+') return kNoFastPath;
        const char *plus_end = static_cast<const char *>(std::memchr(s_end + 1, '\n', static_cast<size_t>(end - (s_end + 1))));
        if (!plus_end) return kNoFastPath;
        const size_t n_sym = static_cast<size_t>(s_end - seq);
        const char *q = plus_end + 1;
        if (static_cast<size_t>(end - q) < n_sym + 1 || q[n_sym] != '\n') return kNoFastPath;  // (a '\n' inside the quality line: not this shape)
        if (std::memchr(q, '\n', n_sym) != nullptr || (n_sym && q[n_sym - 1] == '\r')) return kNoFastPath;
        if (n_sym > room) return kNoFastPath;  // (the general path reports or defers it)
        std::memcpy(out, seq, n_sym);
        pos_ = static_cast<size_t>(q + n_sym + 1 - buf_.data());
        return n_sym;
    }

    // next line without its terminator ('\n', optional '\r' before it); false at the end of the file
    bool next_line(std::string &line)
    {
        line.clear();
        bool any = false;
        for (;;) {
            if (pos_ == len_) {
                if (!eof_) {
                    len_ = std::fread(buf_.data(), 1, buf_.size(), file_);
                    pos_ = 0;
                    if (len_ == 0) {
                        if (std::ferror(file_)) fail(GDX_ERR_INVALID_ARGUMENT, "read error");
                        eof_ = true;
                    }
                }
                if (eof_) break;  // a last line without '\n'
            }
            const char *start = buf_.data() + pos_;
            const char *nl = static_cast<const char *>(std::memchr(start, '\n', len_ - pos_));
            const size_t take = nl ? static_cast<size_t>(nl - start) : len_ - pos_;
            line.append(start, take);
            pos_ += take + (nl ? 1 : 0);
            any = true;
            if (nl) break;
        }
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return any;
    }
    bool peek_line()
    {
        if (!have_line_) have_line_ = next_line(line_);
        return have_line_;
    }
    void consume_line() { have_line_ = false; }

    bool read_record()
    {
        while (peek_line() && line_.empty()) consume_line();
        if (!peek_line()) return false;
        pending_.clear();
        if (line_[0] == '>') {
            consume_line();
            while (peek_line() && (line_.empty() || line_[0] != '>')) {
                pending_.insert(pending_.end(), line_.begin(), line_.end());
                consume_line();
            }
        } else if (line_[0] == '@') {
            consume_line();
            while (peek_line() && (line_.empty() || line_[0] != '+')) {
                pending_.insert(pending_.end(), line_.begin(), line_.end());
                consume_line();
            }
            if (!peek_line()) fail(GDX_ERR_INVALID_ARGUMENT, "FASTQ record %llu has no '+' line", static_cast<unsigned long long>(records_));
            consume_line();
            size_t quality = 0;  // quality strings may start with '@' or '+': count characters, not lines
            while (quality < pending_.size() && peek_line()) {
                quality += line_.size();
                consume_line();
            }
            if (quality != pending_.size())
                fail(GDX_ERR_INVALID_ARGUMENT, "FASTQ record %llu: %zu quality characters for %zu symbols",
                     static_cast<unsigned long long>(records_), quality, pending_.size());
        } else {
            fail(GDX_ERR_INVALID_ARGUMENT, "record %llu starts with '%c' (expected '>' or '@')", static_cast<unsigned long long>(records_),
                 line_[0]);
        }
        have_pending_ = true;
        return true;
    }

    std::FILE *file_;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false;
    std::string line_;
    bool have_line_ = false;
    std::vector<uint8_t> pending_;
    bool have_pending_ = false;
    uint64_t records_ = 0;
};

// ---- the same reader over a memory-mapped file, records parsed by several threads (round 6) ----------------------------------
// One thread parsing a FASTQ file delivers 60 M reads a second; the calls behind it take 1-3 G (ROADMAP.md:35-37 of the reference
// names reading the queries as a cost of its own).  Here a batch is made from a window of the mapped file cut into tiles of 2 MB
// at guessed record starts -- FASTA: a line that starts with '>'; FASTQ: a line that starts with '@' from which two whole records
// parse (a quality line may start with '@' too) -- which the threads take in file order:
//   parse   a tile with the SAME rules as FastxReader::read_record into descriptors {record start, first sequence byte, symbols};
//           plain four-line FASTQ records are read off an index of the tile's newlines (one vector pass per 16 KB), everything
//           else goes through the line loops.  A tile must end exactly where the next one starts: a guess that was wrong (or a
//           malformed record) shows there, and the window is parsed again by one thread from its first byte, which is the
//           authority for errors and their messages;
//   decide  where a tile's records go in qbuf / qoff and how many fit the caller's limits needs the totals of the tiles in front:
//           whichever thread finishes a parse moves that chain on as far as the parsed tiles reach (a few stores per tile);
//   copy    by the tile's own thread, as a rule right after its parse, when the tile's bytes and descriptors are still in the
//           core's caches; a thread whose tile is not decided yet keeps it and parses on (a few tiles, then it waits).
// Results are FastxReader's, byte for byte (tests/test_fastx.py runs both on the same files).  On the bench host: 330 M reads/s
// with one block per thread and two passes (parse all, copy all), 800 M with the tiles (16 threads, 50-symbol reads).
class FastxMappedReader {
public:
    // null when the file cannot be mapped (a pipe, an empty file): the caller falls back to FastxReader
    static FastxMappedReader *open(const char *path, unsigned threads);
    ~FastxMappedReader();
    FastxMappedReader(const FastxMappedReader &) = delete;
    FastxMappedReader &operator=(const FastxMappedReader &) = delete;
    // as FastxReader::next_batch; *uniform_len (optional) = the common length of the batch's records, 0 when they differ
    uint64_t next_batch(uint8_t *qbuf, uint64_t capacity, uint64_t *qoff, uint64_t max_records, uint64_t *uniform_len);

private:
    struct Rec {
        uint64_t start;    // file offset of the record's first byte
        uint32_t seq;      // its first sequence line starts this many bytes further (a header line of 4 GB is refused)
        uint32_t one_line; // 1: the sequence is one line without '\r' -- its symbols are the bytes from there on
        uint64_t symbols;
    };
    struct Block {
        std::vector<Rec> recs;
        uint64_t end = 0;      // where the parse stopped (the start of the record it did not take, or the end of the file)
        uint64_t symbols = 0, min_len = ~0ull, max_len = 0;
        bool ok = true;
        std::string error;     // (authoritative parse only)
        std::vector<uint32_t> newlines;  // (FASTQ tiles: the offsets of the tile's '\n' bytes, see parse_block)
    };
    FastxMappedReader() = default;
    // one line from pos: [*b, *e) without '\n' and one trailing '\r'; returns the position behind it; pos == size_: no line
    // the next '\n' at or behind pos, or size_.  Lines of reads are a few dozen bytes: two 32-byte compares in line find most of
    // them in 2-3 ns, where a memchr call spends 10 on its prologue -- four lines per record, a third of the parse
    uint64_t newline_at(uint64_t pos) const
    {
#ifdef GDX_FASTX_AVX2
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) return newline_at_avx2(pos);
#endif
        const char *nl = static_cast<const char *>(std::memchr(data_ + pos, '\n', size_ - pos));
        return nl ? static_cast<uint64_t>(nl - data_) : size_;
    }
#ifdef GDX_FASTX_AVX2
    __attribute__((target("avx2"))) uint64_t newline_at_avx2(uint64_t pos) const
    {
        const __m256i nl = _mm256_set1_epi8('\n');
        while (pos + 32 <= size_) {
            const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(data_ + pos));
            const uint32_t m = static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nl)));
            if (m != 0u) return pos + static_cast<uint32_t>(__builtin_ctz(m));
            pos += 32;
        }
        while (pos < size_ && data_[pos] != '\n') pos++;
        return pos;
    }
#endif
    uint64_t line_at(uint64_t pos, uint64_t *b, uint64_t *e) const
    {
        const uint64_t at = newline_at(pos);
        const char *nl = at < size_ ? data_ + at : nullptr;
        const uint64_t end = at;
        *b = pos;
        *e = (end > pos && data_[end - 1] == '\r') ? end - 1 : end;
        return nl ? end + 1 : size_;
    }
    // FastxReader::read_record on the mapped bytes: the record at or behind `pos` (blank lines skipped); false at the end of the
    // file; a malformed record: `error` set (when given), *bad = true
    bool record_at(uint64_t pos, Rec *r, uint64_t *next, bool *bad, std::string *error, uint64_t number) const;
    void parse_block(uint64_t from, uint64_t stop, uint64_t max_records, uint64_t max_symbols, Block *out, bool authoritative,
                     uint64_t first_number) const;
    uint64_t guess_record_start(uint64_t from, uint64_t limit) const;
#ifdef GDX_FASTX_AVX2
    static uint64_t newline_index_avx2(const char *p, uint64_t at, uint64_t len, std::vector<uint32_t> &out, uint64_t k);
    void copy_records_avx2(const Rec *recs, uint64_t count, uint8_t *qbuf, uint64_t *qoff, uint64_t at) const;
#endif
    void copy_record(const Rec &r, uint8_t *out) const;

    int fd_ = -1;
    const char *data_ = nullptr;
    uint64_t size_ = 0, cursor_ = 0, records_ = 0;
    unsigned threads_ = 1;
    char kind_ = 0;  // '>' or '@': what the file's first record starts with
    double bytes_per_record_ = 0.0;
    std::vector<Block> scratch_;  // a few per thread: a tile's descriptors (kept between tiles and batches: no fresh pages)
};

}  // namespace gdx

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <thread>

namespace gdx {

inline FastxMappedReader *FastxMappedReader::open(const char *path, unsigned threads)
{
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) {
        ::close(fd);
        return nullptr;
    }
    void *m = mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
        ::close(fd);
        return nullptr;
    }
    (void)madvise(m, static_cast<size_t>(st.st_size), MADV_SEQUENTIAL);
    FastxMappedReader *r = new FastxMappedReader();
    r->fd_ = fd;
    r->data_ = static_cast<const char *>(m);
    r->size_ = static_cast<uint64_t>(st.st_size);
    r->threads_ = threads < 1 ? 1 : threads;
    return r;
}

inline FastxMappedReader::~FastxMappedReader()
{
    if (data_) munmap(const_cast<char *>(data_), size_);
    if (fd_ >= 0) ::close(fd_);
}

inline bool FastxMappedReader::record_at(uint64_t pos, Rec *r, uint64_t *next, bool *bad, std::string *error, uint64_t number) const
{
    uint64_t b, e, nx;
    *bad = false;
    // blank lines between records
    for (;;) {
        if (pos >= size_) return false;
        nx = line_at(pos, &b, &e);
        if (e > b) break;
        pos = nx;
    }
    r->start = pos;
    r->symbols = 0;
    r->one_line = 0;
    if (data_[b] == '@') {
        // the plain four-line record -- header, ONE sequence line, '+' line, ONE quality line of the same length, no '\r' -- without
        // the loops below (FastxReader::fastq_record_in_place); anything else falls through to them
        const uint64_t h_end = nx - 1;                       // the header's '\n' (nx == size_: no newline, not this shape)
        if (nx < size_ && data_[h_end] == '\n') {
            const uint64_t s_end = newline_at(nx);
            if (s_end < size_ && s_end > nx && data_[s_end - 1] != '\r' && s_end + 1 < size_ && data_[s_end + 1] == '+') {
                const uint64_t p_end = newline_at(s_end + 1);
                const uint64_t n_sym = s_end - nx, q = p_end + 1;
                if (p_end < size_ && q + n_sym < size_ && data_[q + n_sym] == '\n' && data_[q + n_sym - 1] != '\r' &&
                    newline_at(q) == q + n_sym && nx - pos <= 0xffffffffull) {
                    r->seq = static_cast<uint32_t>(nx - pos);
                    r->symbols = n_sym;
                    r->one_line = 1;
                    *next = q + n_sym + 1;
                    return true;
                }
            }
        }
    }
    uint64_t seq_lines = 0, first_b = 0, first_e = 0;  // sequence lines seen, the first one's bytes
    auto fail_with = [&](const char *fmt, unsigned long long a1, unsigned long long a2, unsigned long long a3) {
        *bad = true;
        if (error) {
            char buf[160];
            std::snprintf(buf, sizeof(buf), fmt, a1, a2, a3);
            *error = buf;
        }
        return false;
    };
    const char c = data_[b];
    auto seq_line = [&](uint64_t lb, uint64_t le) {
        if (le > lb && seq_lines++ == 0) first_b = lb, first_e = le;
        r->symbols += le - lb;
    };
    auto finish = [&] {
        // (one line and nothing stripped from it: the copy is a memcpy)
        r->one_line = (seq_lines == 1 && first_e - first_b == r->symbols && first_b == r->start + r->seq) ? 1u : 0u;
    };
    if (c == '>' || c == '@') {
        if (nx - r->start > 0xffffffffull) return fail_with("record %llu: a header line of more than 4 GB", number, 0, 0);
        r->seq = static_cast<uint32_t>(nx - r->start);
    }
    if (c == '>') {
        pos = nx;
        while (pos < size_) {
            nx = line_at(pos, &b, &e);
            if (e > b && data_[b] == '>') break;
            seq_line(b, e);
            pos = nx;
        }
        finish();
        *next = pos;
        return true;
    }
    if (c == '@') {
        pos = nx;
        bool plus = false;
        while (pos < size_) {
            nx = line_at(pos, &b, &e);
            if (e > b && data_[b] == '+') {
                plus = true;
                pos = nx;
                break;
            }
            seq_line(b, e);
            pos = nx;
        }
        if (!plus) return fail_with("FASTQ record %llu has no '+' line", number, 0, 0);
        uint64_t quality = 0;  // quality strings may start with '@' or '+': count characters, not lines
        while (quality < r->symbols && pos < size_) {
            nx = line_at(pos, &b, &e);
            quality += e - b;
            pos = nx;
        }
        if (quality != r->symbols)
            return fail_with("FASTQ record %llu: %llu quality characters for %llu symbols", number, quality, r->symbols);
        finish();
        *next = pos;
        return true;
    }
    *bad = true;
    if (error) {
        char buf[96];
        std::snprintf(buf, sizeof(buf), "record %llu starts with '%c' (expected '>' or '@')", static_cast<unsigned long long>(number), c);
        *error = buf;
    }
    return false;
}

#ifdef GDX_FASTX_AVX2
// the offsets of every '\n' in p[at, len), len < 2^32, appended to out[0, k) (out.size() is the room, grown as needed); returns the new
// k.  64 bytes a step, the set bits of the compare mask written without a branch per bit (eight stores whatever the mask holds, the
// cursor moves by its population count)
__attribute__((target("avx2,bmi,popcnt"))) inline uint64_t FastxMappedReader::newline_index_avx2(const char *p, uint64_t at, uint64_t len,
                                                                                                  std::vector<uint32_t> &out, uint64_t k)
{
    const __m256i nl = _mm256_set1_epi8('\n');
    for (; at + 64 <= len; at += 64) {
        _mm_prefetch(p + at + 16384, _MM_HINT_T0);  // (the next 16 KB arrive while this one's records are checked)
        const uint64_t m0 = static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + at)), nl)));
        const uint64_t m1 = static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + at + 32)), nl)));
        uint64_t m = m0 | (m1 << 32);
        if (k + 64 > out.size()) out.resize(out.size() * 2);
        uint32_t *o = out.data() + k;
        const uint32_t base = static_cast<uint32_t>(at);
        const unsigned c = static_cast<unsigned>(_mm_popcnt_u64(m));
        for (int j = 0; j < 8; j++) {
            o[j] = base + static_cast<uint32_t>(_tzcnt_u64(m));
            m = _blsr_u64(m);
        }
        for (unsigned j = 8; j < c; j++) {
            o[j] = base + static_cast<uint32_t>(_tzcnt_u64(m));
            m = _blsr_u64(m);
        }
        k += c;
    }
    for (; at < len; at++)
        if (p[at] == '\n') {
            if (k + 1 > out.size()) out.resize(out.size() * 2);
            out[k++] = static_cast<uint32_t>(at);
        }
    return k;
}
#endif

inline void FastxMappedReader::parse_block(uint64_t from, uint64_t stop, uint64_t max_records, uint64_t max_symbols, Block *out,
                                           bool authoritative, uint64_t first_number) const
{
    uint64_t pos = from;
    out->recs.clear();
    if (bytes_per_record_ > 0.0 && stop > from && stop != ~0ull)  // (no growing, copying vector: a batch is millions of records)
        out->recs.reserve(static_cast<size_t>(std::min<double>(static_cast<double>(max_records),
                                                               static_cast<double>((stop < size_ ? stop : size_) - from) / bytes_per_record_ * 1.05)) + 16);
    out->symbols = 0;
    out->min_len = ~0ull;
    out->max_len = 0;
    out->ok = true;
    // A tile of a FASTQ file: all its newlines first (one vector pass), then a plain four-line record is four consecutive entries
    // of that index and the conditions of record_at's own short cut, checked on them -- the same record, without four searches.
    // Whatever is not that shape (blank lines, '\r', folded sequences, the file's last line without '\n') goes to record_at.
    std::vector<uint32_t> &nl = out->newlines;
    uint64_t n_nl = 0, k = 0, indexed = 0;  // nl[k, n_nl): the newlines of [pos, from + indexed) not yet used
    const uint64_t tile_len = stop - from;    // (only read when `wide`)
    bool wide = false;
    (void)wide, (void)tile_len, (void)indexed, (void)k, (void)n_nl;
#ifdef GDX_FASTX_AVX2
    static const bool can = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi") && __builtin_cpu_supports("popcnt");
    const char *index_env = getenv("GDX_FASTX_NEWLINE_INDEX");  // (tests and experiments: 0 = every record by the line loops)
    wide = can && (index_env == nullptr || atoi(index_env) != 0) && !authoritative && kind_ == '@' && stop <= size_ && stop > from && stop - from < 0xffffffffull;
    if (wide && nl.size() < 4096) nl.resize(4096);
#endif
    while (out->recs.size() < max_records && out->symbols <= max_symbols) {  // (the record that overflows is the last one parsed)
        // (blank lines in front of a block's border belong to the record behind them: stop at the border itself)
        if (pos >= stop) break;
        Rec r;
        uint64_t next = pos;
#ifdef GDX_FASTX_AVX2
        if (wide && k + 4 > n_nl && indexed < tile_len) {
            // the next 16 KB of the tile into the index (what the records behind it left unused moves to the front): the bytes are
            // in the first-level cache when the record checks below read them
            std::memmove(nl.data(), nl.data() + k, (n_nl - k) * sizeof(uint32_t));
            n_nl -= k;
            k = 0;
            while (n_nl < 4 && indexed < tile_len) {
                const uint64_t upto = std::min<uint64_t>(tile_len, indexed + 16384);
                n_nl = newline_index_avx2(data_ + from, indexed, upto, nl, n_nl);
                indexed = upto;
            }
        }
#endif
        if (k + 4 <= n_nl && data_[pos] == '@') {
            const uint64_t nx = from + nl[k] + 1, s_end = from + nl[k + 1], p_end = from + nl[k + 2], q_end = from + nl[k + 3];
            const uint64_t n_sym = s_end - nx, q = p_end + 1;
            if (s_end > nx && data_[s_end - 1] != '\r' && data_[s_end + 1] == '+' && q_end == q + n_sym && q + n_sym < size_ &&
                data_[q_end - 1] != '\r' && nx - pos <= 0xffffffffull) {
                r.start = pos;
                r.seq = static_cast<uint32_t>(nx - pos);
                r.one_line = 1;
                r.symbols = n_sym;
                out->recs.push_back(r);
                out->symbols += n_sym;
                out->min_len = n_sym < out->min_len ? n_sym : out->min_len;
                out->max_len = n_sym > out->max_len ? n_sym : out->max_len;
                pos = q_end + 1;
                k += 4;
                continue;
            }
        }
        bool bad = false;
        if (!record_at(pos, &r, &next, &bad, authoritative ? &out->error : nullptr, first_number + out->recs.size())) {
            if (bad) out->ok = false;
            else pos = size_;  // only blank lines were left
            break;
        }
        if (r.start >= stop) {  // (blank lines led across the border)
            pos = r.start;
            break;
        }
        if (!authoritative && kind_ != data_[r.start]) {  // (a file that mixes FASTA and FASTQ: one thread decides)
            out->ok = false;
            break;
        }
        out->recs.push_back(r);
        out->symbols += r.symbols;
        out->min_len = r.symbols < out->min_len ? r.symbols : out->min_len;
        out->max_len = r.symbols > out->max_len ? r.symbols : out->max_len;
        pos = next;
        while (k < n_nl && from + nl[k] < pos) k++;  // (the index follows the record that went the long way)
        if (k == n_nl && pos > from + indexed) indexed = std::min<uint64_t>(tile_len, pos - from);
    }
    out->end = pos;
}

// the first offset in [from, limit) at which a record seems to start; `limit` if none
inline uint64_t FastxMappedReader::guess_record_start(uint64_t from, uint64_t limit) const
{
    uint64_t pos = from;
    if (pos > 0 && data_[pos - 1] != '\n') {  // to the start of the next line
        const char *nl = static_cast<const char *>(std::memchr(data_ + pos, '\n', size_ - pos));
        if (!nl) return limit;
        pos = static_cast<uint64_t>(nl - data_) + 1;
    }
    while (pos < limit) {
        if (data_[pos] == kind_) {
            if (kind_ == '>') return pos;
            // FASTQ: two records in a row must parse from here, and what follows them must be a record start or the end
            Rec r;
            uint64_t p1, p2;
            bool bad;
            if (record_at(pos, &r, &p1, &bad, nullptr, 0) && r.start == pos) {
                Rec r2;
                const bool second = record_at(p1, &r2, &p2, &bad, nullptr, 0);
                if ((second && data_[r2.start] == '@') || (!second && !bad)) return pos;
            }
        }
        const char *nl = static_cast<const char *>(std::memchr(data_ + pos, '\n', size_ - pos));
        if (!nl) return limit;
        pos = static_cast<uint64_t>(nl - data_) + 1;
    }
    return limit;
}

#ifdef GDX_FASTX_AVX2
// a tile's records into qbuf from symbol `at`, their ends into qoff[1..count] (qoff points at the tile's first entry): one-line
// records in 32-byte pieces that may run up to 31 bytes past the record -- into the room of the tile's next records, which are
// written after it, never past the tile's own share of qbuf (the next tile's thread may be there already) or the mapping
__attribute__((target("avx2"))) inline void FastxMappedReader::copy_records_avx2(const Rec *recs, uint64_t count, uint8_t *qbuf, uint64_t *qoff,
                                                                                 uint64_t at) const
{
    uint64_t total = 0;
    for (uint64_t j = 0; j < count; j++) total += recs[j].symbols;
    const uint64_t dst_end = at + total;
    for (uint64_t j = 0; j < count; j++) {
        const Rec &r = recs[j];
        const uint64_t src = r.start + r.seq, whole = (r.symbols + 31) & ~31ull;
        if (r.one_line && at + whole <= dst_end && src + whole <= size_) {
            for (uint64_t b = 0; b < whole; b += 32)
                _mm256_storeu_si256(reinterpret_cast<__m256i *>(qbuf + at + b),
                                    _mm256_loadu_si256(reinterpret_cast<const __m256i *>(data_ + src + b)));
        } else {
            copy_record(r, qbuf + at);
        }
        at += r.symbols;
        qoff[j + 1] = at;
    }
}
#endif

inline void FastxMappedReader::copy_record(const Rec &r, uint8_t *out) const
{
    uint64_t pos = r.start + r.seq, left = r.symbols;
    if (r.one_line) {
        std::memcpy(out, data_ + pos, left);
        return;
    }
    while (left > 0) {
        uint64_t b, e;
        pos = line_at(pos, &b, &e);
        std::memcpy(out, data_ + b, e - b);
        out += e - b;
        left -= e - b;
    }
}

inline uint64_t FastxMappedReader::next_batch(uint8_t *qbuf, uint64_t capacity, uint64_t *qoff, uint64_t max_records, uint64_t *uniform_len)
{
    qoff[0] = 0;
    if (uniform_len) *uniform_len = 0;
    if (max_records == 0 || cursor_ >= size_) return 0;
    const double t_enter = getenv("GDX_FASTX_TIMING") != nullptr
                               ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
    if (kind_ == 0) {  // the first record names the kind of file
        uint64_t p = cursor_;
        while (p < size_ && (data_[p] == '\n' || data_[p] == '\r')) p++;
        kind_ = p < size_ ? data_[p] : '>';
    }
    if (bytes_per_record_ <= 0.0) {  // the first batch: the file's first records say how long a record is
        uint64_t p = cursor_, seen = 0;
        Rec r;
        bool bad = false;
        while (seen < 64 && record_at(p, &r, &p, &bad, nullptr, 0)) seen++;
        if (seen != 0 && !bad) bytes_per_record_ = static_cast<double>(p - cursor_) / static_cast<double>(seen);
    }
    // the window: about max_records records, never less than 1 MB
    const double per = bytes_per_record_ > 0.0 ? bytes_per_record_ : 256.0;
    double want = per * static_cast<double>(max_records) * 1.02 + 65536.0;
    if (bytes_per_record_ <= 0.0 && want > 3.0 * static_cast<double>(capacity) + (1 << 20)) want = 3.0 * static_cast<double>(capacity) + (1 << 20);
    uint64_t w_end = size_ - cursor_ > static_cast<uint64_t>(want) ? cursor_ + static_cast<uint64_t>(want) : size_;
    if (w_end < size_) w_end = guess_record_start(w_end, size_);
    const uint64_t w_bytes = w_end - cursor_;
    // tiles of the window, taken in file order by whichever thread is free (tests: GDX_FASTX_BLOCK_BYTES lowers the size, so that
    // small files are cut into tiles too).  Two megabytes: a tile's bytes and its descriptors are still in the core's caches when
    // they are copied, and the mapping's lock is taken once per tile (MADV_POPULATE_READ), not once per fault
    uint64_t tile_bytes = 2u << 20;
    if (const char *e = getenv("GDX_FASTX_BLOCK_BYTES")) tile_bytes = static_cast<uint64_t>(std::max(1L, atol(e)));
    const uint64_t n_tiles = std::max<uint64_t>(1, (w_bytes + tile_bytes - 1) / tile_bytes);
    const unsigned nt = static_cast<unsigned>(std::min<uint64_t>(threads_, n_tiles));
    constexpr unsigned kHeld = 4;  // parsed tiles a thread keeps while the tile in front of its oldest one is not through
    if (scratch_.size() < nt * kHeld) scratch_.resize(nt * kHeld);
    // A tile's share of the batch -- where its records go, how many of them fit the caller's limits -- is decided when every tile
    // in front of it has been parsed: whichever thread finishes a parse moves that chain on (under the lock: a few stores per
    // tile).  The tile's own thread then copies its records; if the decision is not there yet -- a thread in front was slow, the
    // pipeline behind this reader takes the same CPUs -- it keeps the tile and parses the next one, up to kHeld of them.
    struct Tile {
        const Block *bk = nullptr;
        uint64_t b = 0;                      // the border the tile must end at
        bool parsed = false;                 // (under the lock)
        uint64_t base_rec = 0, base_sym = 0, take = 0;
        std::atomic<uint32_t> decided{0};
    };
    std::unique_ptr<Tile[]> tiles(new Tile[n_tiles]);
    std::mutex lock;
    uint64_t next_tile = 0, next_border = cursor_, chain = 0;  // (under the lock) tiles handed out, tiles decided
    bool stopped = false;                                      // (under the lock) a tile met the limits, or was inconsistent
    std::atomic<bool> stop{false};
    bool inconsistent = false, too_long = false;               // (under the lock, read after the join)
    uint64_t n = 0, used = 0, stop_at = w_end, lo = ~0ull, hi = 0, too_long_symbols = 0;
    const bool timing = getenv("GDX_FASTX_TIMING") != nullptr;  // (debug: where a batch's time goes)
    const char *populate_env = getenv("GDX_FASTX_POPULATE");
    const bool populate = populate_env == nullptr || atoi(populate_env) != 0;
    (void)populate;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a0 = timing ? now() : 0.0;
    double t_stage[4] = {0.0, 0.0, 0.0, 0.0};  // (timing: thread-seconds mapping pages, parsing, waiting for a decision, copying)
    auto move_chain_on = [&] {  // (the caller holds the lock)
        while (chain < n_tiles && tiles[chain].parsed) {
            Tile &tl = tiles[chain++];
            const Block &bk = *tl.bk;
            tl.base_rec = n, tl.base_sym = used, tl.take = 0;
            if (stopped) {
                // (nothing more is taken)
            } else if (!bk.ok || bk.end != tl.b) {
                // a tile must end exactly where the next one starts: a guess that was wrong, or a malformed record
                inconsistent = stopped = true;
            } else if (n + bk.recs.size() <= max_records && used + bk.symbols <= capacity) {  // the whole tile
                tl.take = bk.recs.size();
                n += tl.take, used += bk.symbols, stop_at = bk.end;
            } else {  // what fits the caller's limits
                for (const Rec &r : bk.recs) {
                    if (n == max_records || r.symbols > capacity - used) {
                        if (n == 0 && r.symbols > capacity) too_long = true, too_long_symbols = r.symbols;
                        stop_at = r.start;
                        break;
                    }
                    tl.take++, n++;
                    used += r.symbols;
                }
                stopped = true;
            }
            if (stopped) stop.store(true, std::memory_order_relaxed);
            tl.decided.store(1, std::memory_order_release);
        }
    };
    auto worker = [&](unsigned me) {
        uint64_t held[kHeld];
        unsigned n_held = 0, head = 0;  // held[(head + i) % kHeld], i < n_held: oldest first
        uint64_t my_lo = ~0ull, my_hi = 0;
        double my_t[4] = {0.0, 0.0, 0.0, 0.0}, t0 = 0.0, t1 = 0.0;
        auto copy_tile = [&](const Tile &tl) {
            const Block &bk = *tl.bk;
            if (tl.take == 0) return;
            if (timing) t0 = now();
            if (tl.take == bk.recs.size()) {
                my_lo = bk.min_len < my_lo ? bk.min_len : my_lo;
                my_hi = bk.max_len > my_hi ? bk.max_len : my_hi;
            } else {
                for (uint64_t j = 0; j < tl.take; j++) {
                    my_lo = bk.recs[j].symbols < my_lo ? bk.recs[j].symbols : my_lo;
                    my_hi = bk.recs[j].symbols > my_hi ? bk.recs[j].symbols : my_hi;
                }
            }
#ifdef GDX_FASTX_AVX2
            static const bool avx2 = __builtin_cpu_supports("avx2");
            if (avx2) {
                copy_records_avx2(bk.recs.data(), tl.take, qbuf, qoff + tl.base_rec, tl.base_sym);
            } else
#endif
            {
                uint64_t at = tl.base_sym;
                for (uint64_t j = 0; j < tl.take; j++) {
                    copy_record(bk.recs[j], qbuf + at);
                    at += bk.recs[j].symbols;
                    qoff[tl.base_rec + j + 1] = at;
                }
            }
            if (timing) my_t[3] += now() - t0;
        };
        bool no_more = false;
        for (;;) {
            while (n_held != 0 && tiles[held[head]].decided.load(std::memory_order_acquire) != 0u) {
                copy_tile(tiles[held[head]]);
                head = (head + 1) % kHeld;
                n_held--;
            }
            if (no_more && n_held == 0) break;
            if (no_more || n_held == kHeld) {  // nothing else to do: wait for the oldest tile's decision
                if (timing) t0 = now();
                for (unsigned spins = 0; tiles[held[head]].decided.load(std::memory_order_acquire) == 0u; spins++)
                    if (spins > 64) std::this_thread::yield();
                if (timing) my_t[2] += now() - t0;
                continue;
            }
            if (stop.load(std::memory_order_relaxed)) {
                no_more = true;
                continue;
            }
            uint64_t t, a, b;
            {  // the next tile and its borders: a guessed record start behind every multiple of the tile size
                std::lock_guard<std::mutex> g(lock);
                if (next_tile >= n_tiles) {
                    no_more = true;
                    continue;
                }
                t = next_tile++;
                a = next_border;
                const uint64_t at = cursor_ + tile_bytes * (t + 1);
                b = (t + 1 == n_tiles || at >= w_end) ? w_end : (at <= a ? a : guess_record_start(at, w_end));
                next_border = b;
            }
            if (timing) t0 = now();
#ifdef MADV_POPULATE_READ
            // the tile's pages into this process's page table in one call (Linux 5.14+; an older kernel says EINVAL and the parse
            // faults them in 64 KB at a time)
            if (b > a && populate) {
                const uint64_t a0 = a & ~4095ull;
                (void)madvise(const_cast<char *>(data_) + a0, b - a0, MADV_POPULATE_READ);
            }
#endif
            if (timing) t1 = now(), my_t[0] += t1 - t0;
            Block *bk = &scratch_[me * kHeld + (head + n_held) % kHeld];
            parse_block(a, b, ~0ull, ~0ull, bk, false, 0);
            if (timing) my_t[1] += now() - t1;
            {
                std::lock_guard<std::mutex> g(lock);
                tiles[t].bk = bk, tiles[t].b = b, tiles[t].parsed = true;
                move_chain_on();
            }
            held[(head + n_held) % kHeld] = t;
            n_held++;
        }
        std::lock_guard<std::mutex> g(lock);
        lo = my_lo < lo ? my_lo : lo;
        hi = my_hi > hi ? my_hi : hi;
        for (int k = 0; k < 4; k++) t_stage[k] += my_t[k];
    };
    {
        std::vector<std::thread> th;
        for (unsigned i = 1; i < nt; i++) th.emplace_back([&, i] { worker(i); });
        worker(0);
        for (auto &t : th) t.join();
    }
    const double t_a1 = timing ? now() : 0.0;
    bool authority_failed = false;
    if (inconsistent) {  // one thread, from the window's first byte: the authority
        Block &bk = scratch_[0];
        parse_block(cursor_, size_, max_records, capacity, &bk, true, records_);
        if (!bk.ok && bk.recs.empty()) fail(GDX_ERR_INVALID_ARGUMENT, "%s", bk.error.c_str());
        // (records in front of a malformed one are delivered; the error comes with the batch that starts at it)
        authority_failed = !bk.ok;
        n = 0, used = 0, stop_at = bk.end, lo = ~0ull, hi = 0;
        too_long = false;
        for (const Rec &r : bk.recs) {
            if (n == max_records || r.symbols > capacity - used) {
                if (n == 0 && r.symbols > capacity) too_long_symbols = r.symbols, too_long = true;
                stop_at = r.start;
                break;
            }
            copy_record(r, qbuf + used);
            used += r.symbols;
            qoff[++n] = used;
            lo = r.symbols < lo ? r.symbols : lo;
            hi = r.symbols > hi ? r.symbols : hi;
        }
    }
    if (too_long)
        fail(GDX_ERR_CAPACITY, "record %llu has %llu symbols, the buffer holds %llu", static_cast<unsigned long long>(records_),
             static_cast<unsigned long long>(too_long_symbols), static_cast<unsigned long long>(capacity));
    if (timing)
        std::fprintf(stderr, "gdx fastx batch: %llu tiles on %u threads, window %.4f s, tiles %.4f s (thread-seconds: map %.3f, parse %.3f, "
                     "wait %.3f, copy %.3f), authority %.4f s\n", static_cast<unsigned long long>(n_tiles), nt, t_a0 - t_enter, t_a1 - t_a0,
                     t_stage[0], t_stage[1], t_stage[2], t_stage[3], now() - t_a1);
    if (uniform_len && n != 0 && lo == hi) *uniform_len = lo;
    if (n != 0) bytes_per_record_ = static_cast<double>(stop_at - cursor_) / static_cast<double>(n);
    if (n == 0 && authority_failed) fail(GDX_ERR_INVALID_ARGUMENT, "%s", scratch_[0].error.c_str());
    cursor_ = stop_at;
    records_ += n;
    return n;
}

}  // namespace gdx
