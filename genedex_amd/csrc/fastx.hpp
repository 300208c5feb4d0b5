// fastx.hpp -- streaming FASTA / FASTQ reader that fills the ABI's query layout (qbuf + qoff) batch by batch.
// Host-only.  SURVEY.md section 8f row 3: the reference leaves reading to its callers and notes that it can
// cost more than searching (ROADMAP.md:35-37); this is the ingestion side of gdx_count_many / gdx_locate_many
// and of gdx_index_build (texts).
//
// Accepted: FASTA ('>' header lines, sequences over any number of lines) and FASTQ ('@' header, sequence lines,
// '+' line, as many quality characters as sequence symbols, also over several lines).  '\r' is dropped, empty
// lines are skipped, sequence bytes are copied as they are (the alphabet table decides what is valid).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "errors.hpp"

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

}  // namespace gdx
