"""FASTA / FASTQ ingestion (include/gdx.h gdx_fastx_*): host-only code of libgdx.so, runs without a GPU."""
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
