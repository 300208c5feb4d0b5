"""FASTA / FASTQ ingestion (include/gdx.h gdx_fastx_*): host-only code of libgdx.so, runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from genedex_amd import GdxError, fastx


def naive_fasta(text: str):
    seqs, cur = [], None
    for line in text.replace("\r", "").split("\n"):
        if line.startswith(">"):
            if cur is not None:
                seqs.append(cur)
            cur = ""
        elif cur is not None:
            cur += line
    if cur is not None:
        seqs.append(cur)
    return [s.encode() for s in seqs]


def test_fasta_multiline_crlf_and_no_trailing_newline(tmp_path):
    rng = np.random.default_rng(5)
    parts = []
    for i in range(300):
        seq = "".join(rng.choice(list("ACGTN"), int(rng.integers(0, 500))))
        width = int(rng.integers(1, 90))
        lines = [seq[j:j + width] for j in range(0, len(seq), width)]
        eol = "\r\n" if i % 3 == 0 else "\n"
        parts.append(f">seq{i} some description{eol}" + eol.join(lines) + (eol if lines else "") + ("\n" if i % 7 == 0 else ""))
    text = "".join(parts).rstrip("\n")  # the last line has no terminator
    path = tmp_path / "a.fa"
    path.write_bytes(text.encode())
    want = naive_fasta(text)
    assert fastx.read_sequences(str(path)) == want
    # small batches: records that do not fit stay pending for the next call
    got = []
    for qbuf, qoff in fastx.read_batches(str(path), max_records=7, buffer_bytes=1200):
        raw = qbuf.tobytes()
        assert qoff[0] == 0 and qoff.size - 1 <= 7 and int(qoff[-1]) <= 1200
        got += [raw[int(qoff[i]):int(qoff[i + 1])] for i in range(qoff.size - 1)]
    assert got == want


def test_fastq_with_quality_lines_that_look_like_headers(tmp_path):
    rng = np.random.default_rng(6)
    want, parts = [], []
    for i in range(500):
        seq = "".join(rng.choice(list("ACGT"), int(rng.integers(1, 160))))
        qual = "".join(rng.choice(list("@+>IJK#!"), len(seq)))  # '@' / '+' / '>' are legal quality characters
        if i % 5 == 0:  # multi-line record
            half = len(seq) // 2
            parts.append(f"@r{i}\n{seq[:half]}\n{seq[half:]}\n+r{i}\n{qual[:half]}\n{qual[half:]}\n")
        else:
            parts.append(f"@r{i}\n{seq}\n+\n{qual}\n")
        want.append(seq.encode())
    path = tmp_path / "a.fq"
    path.write_bytes("".join(parts).encode())
    assert fastx.read_sequences(str(path)) == want


def test_errors_are_reported(tmp_path):
    with pytest.raises(GdxError):
        fastx.read_sequences(str(tmp_path / "missing.fa"))
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@r\nACGT\n+\nII\n")  # too few quality characters
    with pytest.raises(GdxError):
        fastx.read_sequences(str(bad))
    junk = tmp_path / "junk.txt"
    junk.write_bytes(b"ACGT\n")
    with pytest.raises(GdxError):
        fastx.read_sequences(str(junk))
    big = tmp_path / "big.fa"
    big.write_bytes(b">x\n" + b"A" * 5000 + b"\n")
    with pytest.raises(GdxError):  # one record larger than the whole buffer
        list(fastx.read_batches(str(big), max_records=4, buffer_bytes=1000))
    empty = tmp_path / "empty.fa"
    empty.write_bytes(b"")
    assert fastx.read_sequences(str(empty)) == []


@pytest.mark.gpu
def test_index_and_queries_from_files(tmp_path):
    """Texts from a FASTA file, reads from a FASTQ file, through the same ABI as everything else."""
    from genedex_amd import FmIndexConfig, alphabet

    rng = np.random.default_rng(8)
    texts = ["".join(rng.choice(list("ACGTN"), int(rng.integers(500, 5000)), p=[.24, .24, .24, .24, .04])) for _ in range(6)]
    fa = tmp_path / "ref.fa"
    fa.write_text("".join(f">chr{i}\n" + "\n".join(t[j:j + 60] for j in range(0, len(t), 60)) + "\n" for i, t in enumerate(texts)))
    reads = []
    for _ in range(400):
        t = texts[int(rng.integers(0, 6))]
        pos = int(rng.integers(0, len(t) - 40))
        reads.append(t[pos:pos + int(rng.integers(10, 40))])
    fq = tmp_path / "reads.fq"
    fq.write_text("".join(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n" for i, r in enumerate(reads)))
    index = FmIndexConfig("u32").construct_index(fastx.read_sequences(str(fa)), alphabet.ascii_dna_with_n())
    total = 0
    for qbuf, qoff in fastx.read_batches(str(fq), max_records=150):
        raw = qbuf.tobytes()
        seqs = [raw[int(qoff[i]):int(qoff[i + 1])] for i in range(qoff.size - 1)]
        for s, c in zip(seqs, index.count_many(seqs)):
            want = sum(sum(1 for k in range(len(t) - len(s) + 1) if t.startswith(s.decode(), k)) for t in texts)
            assert int(c) == want
        total += len(seqs)
    assert total == len(reads)
    # the same file as packed batches (2-bit codes; reads with N are the exceptions and go through the plain call), and a file
    # of reads of one length as packed + uniform batches without offsets: the counts of the plain calls
    a = alphabet.ascii_dna_with_n()
    for path, lens in ((fq, None), (tmp_path / "uniform.fq", 36)):
        if lens is not None:
            uni = []
            for _ in range(500):
                t = texts[int(rng.integers(0, 6))]
                pos = int(rng.integers(0, len(t) - lens))
                uni.append(t[pos:pos + lens])
            path.write_text("".join(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n" for i, r in enumerate(uni)))
        for b in fastx.read_packed_batches(str(path), a, max_records=170):
            want, _ = index.count_raw(np.concatenate([b["qbuf"], np.zeros(8, np.uint8)]), b["qoff"])
            got, _ = index.count_layout_raw(b["packed"], None if b["uniform_len"] else b["qoff"], b["nq"], packed=True,
                                            uniform_len=b["uniform_len"])
            keep = np.ones(b["nq"], dtype=bool)
            keep[b["exceptions"].astype(np.int64)] = False
            assert got[keep].tolist() == want[keep].tolist() and (lens is None or b["uniform_len"] == lens)
            assert len(b["exceptions"]) > 0 or not any(b"N" in bytes(b["qbuf"][int(b["qoff"][i]):int(b["qoff"][i + 1])]) for i in range(b["nq"]))


def test_packed_batches_of_a_fastq_file(tmp_path):
    """fastx.read_packed_batches (gdx_fastx_next_batch -> gdx_pack_queries_table, host only): a FASTQ file of reads of one
    length comes out as 2-bit codes with `uniform_len` set and the reads with an N listed as exceptions; a file of mixed
    lengths keeps its offsets.  The codes are the dense symbols minus one, symbol j in bits 2 (j & 3) of byte j >> 2."""
    from genedex_amd import alphabet

    a = alphabet.ascii_dna_with_n()
    dense = a.io_to_dense_table
    rng = np.random.default_rng(8)

    def write(path, reads):
        path.write_bytes("".join(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n" for i, r in enumerate(reads)).encode())

    def check(batches, reads):
        seen = 0
        for b in batches:
            n = b["nq"]
            mine = reads[seen:seen + n]
            seen += n
            joined = "".join(mine).encode()
            want_exc = [i for i, r in enumerate(mine) if any(not 1 <= dense[ord(c)] <= 4 for c in r)]
            assert b["exceptions"].tolist() == want_exc
            bits = np.zeros(len(joined), dtype=np.uint8)
            for j, c in enumerate(joined):
                d = int(dense[c])
                bits[j] = d - 1 if 1 <= d <= 4 else 0
            got = b["packed"][: (len(joined) + 3) // 4]
            unpacked = np.stack([(got >> (2 * k)) & 3 for k in range(4)], axis=1).reshape(-1)[: len(joined)]
            assert unpacked.tolist() == bits.tolist()
            lens = {len(r) for r in mine}
            assert b["uniform_len"] == (lens.pop() if len(lens) == 1 else 0)
            assert b["qoff"].tolist() == np.concatenate([[0], np.cumsum([len(r) for r in mine])]).tolist()
            assert b["qbuf"].tobytes() == joined
        assert seen == len(reads)

    uniform = ["".join(rng.choice(list("ACGTacgt"), 50)) for _ in range(900)]
    for i in range(0, 900, 37):
        uniform[i] = uniform[i][:20] + "N" + uniform[i][21:]
    write(tmp_path / "u.fq", uniform)
    check(fastx.read_packed_batches(str(tmp_path / "u.fq"), a, max_records=256, buffer_bytes=1 << 16), uniform)
    mixed = ["".join(rng.choice(list("ACGT"), int(rng.integers(1, 160)))) for _ in range(500)]
    write(tmp_path / "m.fq", mixed)
    check(fastx.read_packed_batches(str(tmp_path / "m.fq"), a, max_records=100, buffer_bytes=1 << 14), mixed)


def _pack_reference(table, qbuf, qoff):
    """2-bit packing restated with numpy: (packed bytes, sorted exception queries)"""
    n_sym = int(qoff[-1])
    d = table[qbuf[:n_sym]].astype(np.int64)
    ok = (d >= 1) & (d <= 4)
    ok[: int(qoff[0])] = True  # symbols before the first query are not looked at
    code = np.where(ok, d - 1, 0)
    code[: int(qoff[0])] = 0
    pad = np.zeros((n_sym + 3) // 4 * 4, dtype=np.int64)
    pad[:n_sym] = code
    packed = (pad[0::4] | (pad[1::4] << 2) | (pad[2::4] << 4) | (pad[3::4] << 6)).astype(np.uint8)
    bad_pos = np.flatnonzero(~ok)
    exc = np.unique(np.searchsorted(qoff, bad_pos, side="right") - 1)
    return packed, exc


@pytest.mark.parametrize("name", ["ascii_dna", "ascii_dna_with_n", "ascii_dna_iupac", "ascii_dna_iupac_as_dna_with_n", "u8_until_4",
                                  "random_table", "low_nibble_clash"])
def test_host_packer_equals_the_byte_loop_on_every_table(name):
    """gdx_pack_queries_table (pack_host.hpp: 32 symbols per step through nibble look-ups when the alphabet's table has the
    shape for it, the byte loop otherwise) == the table applied byte by byte, for the stock alphabets, a table of no
    particular shape and one whose symbols share low nibbles; batches that start inside the buffer, end off a 32-symbol
    border, and hold every byte value."""
    from genedex_amd import _lib, alphabet

    lib = _lib.load()
    rng = np.random.default_rng(hash(name) % 1000)
    if name == "random_table":
        table = rng.integers(0, 7, 256).astype(np.uint8)
    elif name == "low_nibble_clash":  # 'A' (0x41) and 'Q' (0x51) are different symbols: no nibble plan
        table = np.zeros(256, dtype=np.uint8)
        table[[0x41, 0x51, 0x43, 0x47]] = [1, 2, 3, 4]
    elif name == "u8_until_4":
        table = np.zeros(256, dtype=np.uint8)
        table[:4] = [1, 2, 3, 4]
    else:
        table = np.ascontiguousarray(getattr(alphabet, name)().io_to_dense_table, dtype=np.uint8)
    for trial in range(6):
        n = int(rng.integers(0, 3000)) if trial else 100_003
        pool = np.frombuffer(b"ACGTacgtACGTNnRYKMUu\x00\xff", dtype=np.uint8)
        qbuf = pool[rng.integers(0, pool.size if trial % 2 else 8, n)].copy()
        if trial == 3:
            qbuf = rng.integers(0, 256, n).astype(np.uint8)  # every byte value
        if name == "u8_until_4":
            qbuf = (qbuf & 7).astype(np.uint8)
        cuts = np.sort(rng.integers(0, n + 1, int(rng.integers(1, 40))))
        qoff = np.unique(np.concatenate([cuts, [n]])).astype(np.uint64) if trial % 3 else np.array([0, n], dtype=np.uint64)
        nq = qoff.size - 1
        packed = np.full(int(lib.gdx_packed_bytes(n)), 0xEE, dtype=np.uint8)
        exc = np.zeros(nq + 1, dtype=np.uint64)
        n_exc = C.c_uint64(0)
        rc = lib.gdx_pack_queries_table(table.ctypes.data_as(_lib.u8p), qbuf.ctypes.data_as(_lib.u8p) if n else None,
                                        qoff.ctypes.data_as(_lib.u64p), nq, packed.ctypes.data_as(_lib.u8p),
                                        exc.ctypes.data_as(_lib.u64p), exc.size, C.byref(n_exc))
        assert rc == 0, lib.gdx_last_error()
        want, want_exc = _pack_reference(table, qbuf, qoff.astype(np.int64))
        assert np.array_equal(packed[: want.size], want), (name, trial)
        assert bool((packed[want.size:] == 0xEE).all())  # nothing beyond the packed symbols is written
        assert exc[: n_exc.value].tolist() == want_exc.tolist(), (name, trial)


def _random_fastx_file(rng, kind, n_records):
    """a file that uses what the format allows: wrapped sequences and qualities, CRLF, blank lines between records, quality
    lines that start with '@' / '+' / '>', empty sequences, no newline at the end"""
    parts, want = [], []
    for i in range(n_records):
        ln = int(rng.integers(0, 140)) if i % 11 else 0
        seq = "".join(rng.choice(list("ACGTN"), ln))
        eol = "\r\n" if i % 4 == 0 else "\n"
        width = int(rng.integers(1, 80)) if i % 3 == 0 else 10 ** 6
        s_lines = [seq[j:j + width] for j in range(0, len(seq), width)]
        if kind == "fasta":
            parts.append(f">s{i} d{eol}" + "".join(x + eol for x in s_lines) + ("\n" if i % 9 == 0 else ""))
        else:
            qual = "".join(rng.choice(list("@+>IJ#!"), ln))
            q_width = int(rng.integers(1, 80)) if i % 3 == 0 else 10 ** 6
            q_lines = [qual[j:j + q_width] for j in range(0, len(qual), q_width)]
            parts.append(f"@r{i}{eol}" + "".join(x + eol for x in s_lines) + f"+{eol}" + "".join(x + eol for x in q_lines)
                         + ("\n" if i % 9 == 0 and ln else ""))
        want.append(seq.encode())
    return "".join(parts).rstrip("\n").encode() if kind == "fasta" else "".join(parts).encode(), want


@pytest.mark.parametrize("kind", ["fasta", "fastq"])
@pytest.mark.parametrize("threads,block", [(0, 0), (1, 64), (3, 300), (8, 64), (8, 5000), (5, 40000)])
def test_the_mapped_reader_delivers_what_the_streaming_reader_delivers(tmp_path, monkeypatch, kind, threads, block):
    """gdx_fastx_next_batch on a regular file: the file is mapped and a batch parsed by several threads, blocks cut at guessed
    record starts and checked to meet (fastx.hpp FastxMappedReader); GDX_FASTX_THREADS=0 is the streaming reader of rounds 1-5.
    Same records, same order, same batch limits -- with blocks of a few dozen bytes, so that guesses land inside wrapped
    records and quality lines that look like headers."""
    rng = np.random.default_rng(900 + threads + block)
    data, want = _random_fastx_file(rng, kind, 700)
    path = tmp_path / f"x.{kind}"
    path.write_bytes(data)
    monkeypatch.setenv("GDX_FASTX_THREADS", str(threads))
    if block:
        monkeypatch.setenv("GDX_FASTX_BLOCK_BYTES", str(block))
    assert fastx.read_sequences(str(path)) == want
    for max_records, buffer_bytes in ((5, 400), (64, 100000), (10 ** 6, 700)):
        got = []
        for qbuf, qoff, ulen in fastx.read_batches(str(path), max_records=max_records, buffer_bytes=buffer_bytes, with_uniform_len=True):
            raw = qbuf.tobytes()
            n = qoff.size - 1
            assert qoff[0] == 0 and 0 < n <= max_records and int(qoff[-1]) <= buffer_bytes
            lens = np.diff(qoff.astype(np.int64))
            assert ulen == (int(lens[0]) if bool((lens == lens[0]).all()) else 0)
            got += [raw[int(qoff[i]):int(qoff[i + 1])] for i in range(n)]
        assert got == want, (max_records, buffer_bytes)


@pytest.mark.parametrize("index", ["1", "0"])
def test_the_mapped_reader_on_plain_records_with_a_few_odd_ones(tmp_path, monkeypatch, index):
    """A FASTQ file as sequencers write it -- four lines a record -- with a wrapped, a CRLF and a blank-line-led record every few
    thousand, several tiles of the default size and of 50 KB on six threads: tiles are handed out in file order, decided in that
    order by whichever thread finishes a parse, and copied by their own threads (FastxMappedReader::next_batch); plain records
    are found in the tile's newline index (parse_block), everything else by the line loops, which the index then follows."""
    rng = np.random.default_rng(77)
    n = 60_000
    lens = rng.integers(1, 90, n)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(k))) for k in lens]
    parts = []
    for i, sq in enumerate(seqs):
        q = b"@" * len(sq) if i % 7 == 0 else b"I" * len(sq)  # (quality lines that start with '@')
        if i % 3001 == 5:
            parts.append(b"\n@w%d\r\n%s\r\n+\r\n%s\r\n" % (i, sq, q))
        elif i % 2503 == 7 and len(sq) > 4:
            parts.append(b"@f%d\n%s\n%s\n+\n%s\n%s\n" % (i, sq[:3], sq[3:], q[:2], q[2:]))
        else:
            parts.append(b"@r%d some text\n%s\n+\n%s\n" % (i, sq, q))
    path = tmp_path / "plain.fq"
    path.write_bytes(b"".join(parts)[:-1])  # (no newline at the end of the file)
    if index == "0":
        monkeypatch.setenv("GDX_FASTX_NEWLINE_INDEX", "0")
    for threads, block in ((0, 0), (6, 0), (6, 50_000), (2, 1 << 20)):
        monkeypatch.setenv("GDX_FASTX_THREADS", str(threads))
        if block:
            monkeypatch.setenv("GDX_FASTX_BLOCK_BYTES", str(block))
        else:
            monkeypatch.delenv("GDX_FASTX_BLOCK_BYTES", raising=False)
        for max_records, buffer_bytes in ((25_000, 1 << 22), (10 ** 6, 400_000)):
            got = []
            for qbuf, qoff, ulen in fastx.read_batches(str(path), max_records=max_records, buffer_bytes=buffer_bytes, with_uniform_len=True):
                raw, k = qbuf.tobytes(), qoff.size - 1
                assert 0 < k <= max_records and int(qoff[-1]) <= buffer_bytes and ulen == 0
                got += [raw[int(qoff[i]):int(qoff[i + 1])] for i in range(k)]
            assert got == seqs, (threads, block, max_records)


def test_random_files_limits_threads_and_tile_sizes(tmp_path, monkeypatch):
    """60 random files x 3 random (threads, tile bytes, max_records, buffer bytes): the mapped reader delivers the streaming
    reader's batches' records in its order, fails where it fails (a record larger than the buffer) and names the same batch
    property (uniform length).  1100 files x 4 ran clean when the tiles were written (round 6)."""
    from genedex_amd import _lib

    rng = np.random.default_rng(606)
    for trial in range(60):
        kind = "fastq" if trial % 3 else "fasta"
        n_rec = int(rng.integers(1, 300))
        if trial % 5 == 0:  # what sequencers write, a blank line or an '@' quality line here and there
            want = [bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(1, 200)))) for _ in range(n_rec)]
            data = b"".join((b"\n" if rng.random() < 0.02 else b"") + b"@r%d\n%s\n+\n%s\n" % (i, sq, (b"@" if rng.random() < 0.2 else b"I") * len(sq))
                            for i, sq in enumerate(want))
            data, kind = (data[:-1] if rng.random() < 0.5 else data), "fastq"
        else:
            data, want = _random_fastx_file(rng, kind, n_rec)
        path = tmp_path / f"f{trial}.{kind}"
        path.write_bytes(data)

        def run(threads, block, max_records, cap):
            monkeypatch.setenv("GDX_FASTX_THREADS", str(threads))
            if block:
                monkeypatch.setenv("GDX_FASTX_BLOCK_BYTES", str(block))
            else:
                monkeypatch.delenv("GDX_FASTX_BLOCK_BYTES", raising=False)
            out = []
            try:
                for qb, qo, ul in fastx.read_batches(str(path), max_records=max_records, buffer_bytes=cap, with_uniform_len=True):
                    raw, k = qb.tobytes(), qo.size - 1
                    lens = np.diff(qo.astype(np.int64))
                    assert 0 < k <= max_records and int(qo[-1]) <= cap
                    assert ul == (int(lens[0]) if bool((lens == lens[0]).all()) else 0)
                    out += [raw[int(qo[i]):int(qo[i + 1])] for i in range(k)]
            except _lib.GdxError as e:
                return out, e.status
            return out, None

        longest = max([len(w) for w in want] + [1])
        for _ in range(3):
            threads, block = int(rng.integers(1, 12)), int(rng.choice([16, 64, 300, 2000, 20000, 0]))
            max_records = int(rng.choice([1, 2, 7, 100, 10 ** 6]))
            cap = int(rng.choice([max(1, longest - 1), longest, longest + 3, 5 * longest, 10 ** 7]))
            a, b = run(0, 0, max_records, cap), run(threads, block, max_records, cap)
            assert a == b, (trial, kind, threads, block, max_records, cap)
            assert a[1] == _lib.GDX_ERR_CAPACITY if cap < longest else (a[1] is None and a[0] == want)


@pytest.mark.parametrize("threads", [0, 3])
def test_a_record_larger_than_the_buffer_is_read_after_the_buffer_grew(tmp_path, monkeypatch, threads):
    """gdx_fastx_next_batch[_ex]: GDX_ERR_CAPACITY for a record that cannot fit, the reader stays at it (both readers);
    fastx.read_sequences doubles its buffer -- a chromosome in a genome's FASTA file is such a record."""
    import ctypes as C

    from genedex_amd import _lib

    monkeypatch.setenv("GDX_FASTX_THREADS", str(threads))
    monkeypatch.setenv("GDX_FASTX_BLOCK_BYTES", "4096")
    rng = np.random.default_rng(5)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), k)) for k in (300, 70_000, 20, 150_000, 5)]
    path = tmp_path / "long.fa"
    path.write_bytes(b"".join(b">s%d\n" % i + b"\n".join(sq[j:j + 70] for j in range(0, len(sq), 70)) + b"\n" for i, sq in enumerate(seqs)))
    assert fastx.read_sequences(str(path), buffer_bytes=1 << 10) == seqs
    lib = _lib.load()
    handle = C.c_void_p()
    _lib.check(lib.gdx_fastx_open(str(path).encode(), C.byref(handle)))
    try:
        qbuf, qoff, n = np.empty(1 << 18, dtype=np.uint8), np.empty(9, dtype=np.uint64), C.c_uint64(0)
        call = lambda cap: lib.gdx_fastx_next_batch_ex(handle, qbuf.ctypes.data_as(C.c_void_p), cap, qoff.ctypes.data_as(C.c_void_p), 8,
                                                       C.byref(n), None)
        assert call(1000) == 0 and n.value == 1                  # the first record fits, the second does not: one record
        assert call(1000) == _lib.GDX_ERR_CAPACITY               # now the long one is first
        assert b"70000 symbols" in lib.gdx_last_error()
        assert call(1000) == _lib.GDX_ERR_CAPACITY               # ... and stays first
        rest = b""  # (a batch may hold fewer records than would fit: the mapped reader sizes its window from the records so far)
        while call(1 << 18) == 0 and n.value != 0:
            rest += qbuf[: int(qoff[n.value])].tobytes()
        assert rest == b"".join(seqs[1:]) and n.value == 0
    finally:
        lib.gdx_fastx_close(handle)


@pytest.mark.parametrize("threads", [0, 4])
def test_malformed_records_are_reported_by_either_reader(tmp_path, monkeypatch, threads):
    monkeypatch.setenv("GDX_FASTX_THREADS", str(threads))
    monkeypatch.setenv("GDX_FASTX_BLOCK_BYTES", "32")
    good = b"".join(b"@r%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(200))
    for tail in (b"@last\nACGT\n+\nII\n", b"@last\nACGT\n", b"garbage\n"):
        p = tmp_path / "m.fq"
        p.write_bytes(good + tail)
        with pytest.raises(GdxError):
            fastx.read_sequences(str(p))
    trunc = tmp_path / "t.fq"
    trunc.write_bytes(good[:-7])  # the last quality line cut short
    with pytest.raises(GdxError):
        fastx.read_sequences(str(trunc))
