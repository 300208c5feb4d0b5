"""The host-side code of libgdx.so that parses untrusted bytes (FASTA / FASTQ reader, index-file header), compiled for
the CPU with AddressSanitizer + UndefinedBehaviorSanitizer and fed malformed inputs: every one must end in a clean
error, none in a sanitizer report.  (GPU sanitizers are not available on the pool; the kernels are covered by the
parity tests.)  The same sources are compiled into libgdx.so."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_checks", "host_checks.cpp")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("host_checks") / "host_checks")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-pthread", SRC, "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]

    def run(*args):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=120, env=env)
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        assert r.returncode in (0, 3), (r.returncode, r.stdout, r.stderr[-2000:])
        return r.returncode, r.stdout.strip()

    return run


def test_fastx_reader_on_wellformed_and_malformed_files(checker, tmp_path):
    rng = np.random.default_rng(11)
    seqs = ["".join(rng.choice(list("ACGTN"), int(rng.integers(0, 300)))) for _ in range(200)]
    fa = tmp_path / "ok.fa"
    fa.write_text("".join(f">s{i}\n{s[:70]}\n{s[70:]}\n" for i, s in enumerate(seqs)))
    rc, out = checker("fastx", fa, 7, 700)
    assert rc == 0 and out.split()[:3] == ["ok", "200", str(sum(map(len, seqs)))]
    fq = tmp_path / "ok.fq"
    fq.write_text("".join(f"@r{i}\n{s}\n+\n{'@' * len(s)}\n" for i, s in enumerate(seqs) if s))
    rc, out = checker("fastx", fq, 1000, 1 << 20)
    assert rc == 0 and out.split()[1] == str(sum(1 for s in seqs if s))
    cases = {
        "truncated_quality.fq": "@r\nACGTACGT\n+\n@@@",                     # fewer quality characters than symbols
        "no_plus.fq": "@r\nACGTACGT\nACGT",                                 # the file ends inside the sequence
        "wrong_first_byte.fa": "ACGT\n>x\nAC\n",                           # a record must start with '>' or '@'
        "quality_overrun.fq": "@r\nAC\n+\n@@@@@@@@\n@s\nAC\n+\n@@\n",       # more quality than symbols
        "huge_record.fa": ">x\n" + "A" * 5000 + "\n",                       # a record larger than the whole buffer
        "only_header.fq": "@r",
        "binary_junk.fa": ">" + "".join(map(chr, rng.integers(1, 255, 4000))),
        "nul_bytes.fq": "@r\n\0\0\0\n+\n\0\0\0\n",
    }
    # (round 6) the memory-mapped reader that parses blocks of the file in parallel: the same line for the same file, with blocks
    # of a few dozen bytes so that block borders fall inside records
    os.environ["GDX_FASTX_BLOCK_BYTES"] = "48"
    try:
        for path, mr, cap in ((fa, 7, 700), (fq, 1000, 1 << 20), (fq, 3, 400)):
            for threads in (1, 5):
                assert checker("fastxmap", path, mr, cap, threads) == checker("fastx", path, mr, cap), (path, mr, cap, threads)
        crlf = tmp_path / "crlf.fq"
        crlf.write_bytes("".join(f"@r{i}\r\n{s[:40]}\r\n{s[40:]}\r\n+\r\n{'+' * min(len(s), 40)}\r\n{'@' * max(len(s) - 40, 0)}\r\n\r\n"
                                 for i, s in enumerate(seqs)).encode())
        assert checker("fastxmap", crlf, 50, 5000, 6) == checker("fastx", crlf, 50, 5000)
        assert checker("fastx", crlf, 50, 5000)[1].split()[1] == "200"
    finally:
        pass
    for name, text in cases.items():
        p = tmp_path / name
        p.write_bytes(text.encode("latin-1"))
        rc, out = checker("fastx", p, 3, 64)
        assert rc in (0, 3), (name, out)
        rc_m, out_m = checker("fastxmap", p, 3, 64, 4)
        assert (rc_m, out_m.split(":")[0]) == (rc, out.split(":")[0]), (name, out, out_m)
        if rc == 0:
            assert out_m == out, (name, out, out_m)
        if name in ("truncated_quality.fq", "no_plus.fq", "wrong_first_byte.fa", "huge_record.fa", "only_header.fq"):
            assert rc == 3 and out.startswith("error:"), (name, out)
    os.environ.pop("GDX_FASTX_BLOCK_BYTES", None)
    (tmp_path / "empty.fa").write_bytes(b"")
    rc, out = checker("fastx", tmp_path / "empty.fa", 3, 64)
    assert (rc, out.split()[:3]) == (0, ["ok", "0", "0"])
    rc, out = checker("fastx", tmp_path / "missing.fa", 3, 64)
    assert rc == 3 and "cannot open" in out


def header_bytes(n=1000, n_texts=2, sa_rate=4, sigma=6, n_searchable=4, depth=0, width=32, plane_words=None,
                 n_samples=None, magic=b"GDXIDX01", table=bytes(256)):
    bits = max(1, (sigma - 1).bit_length()) if sigma > 1 else 0
    if plane_words is None:
        plane_words = -(-(n + 1) // 64) * bits
    if n_samples is None:
        n_samples = -(-n // sa_rate) if sa_rate else 0
    return (magic + struct.pack("<5Q4i", n, n_texts, sa_rate, plane_words, n_samples, sigma, n_searchable, depth, width)
            + table), plane_words, n_samples


def test_index_file_header_validation(checker, tmp_path):
    def write(name, header, payload_bytes):
        p = tmp_path / name
        p.write_bytes(header + bytes(payload_bytes))
        return p

    h, pw, ns = header_bytes()
    payload = (6 + 1 + 3 * 2 + pw) * 8 + ns * 4
    rc, out = checker("header", write("good.gdx", h, payload))
    assert rc == 0 and out == "ok n=1000 texts=2 sigma=6"
    bad = {
        "short_payload": (h, payload - 1),
        "long_payload": (h, payload + 8),
        "no_payload": (h, 0),
        "half_header": (h[:100], 0),
        "wrong_magic": (header_bytes(magic=b"NOTANIDX")[0], payload),
        "plane_words_small": (header_bytes(plane_words=1)[0], (6 + 1 + 6 + 1) * 8 + ns * 4),   # the ADVICE case
        "plane_words_huge": (header_bytes(plane_words=1 << 60)[0], payload),
        "n_texts_overflow": (header_bytes(n_texts=(1 << 64) - 1)[0], payload),
        "n_texts_gt_n": (header_bytes(n=5, n_texts=6)[0], payload),
        "n_too_large": (header_bytes(n=1 << 40)[0], payload),
        "zero_rate": (header_bytes(sa_rate=0)[0], payload),
        "samples_mismatch": (header_bytes(n_samples=3)[0], payload),
        "sigma_1": (header_bytes(sigma=1)[0], payload),
        "sigma_1000": (header_bytes(sigma=1000)[0], payload),
        "searchable_ge_sigma": (header_bytes(n_searchable=6)[0], payload),
        "negative_depth": (header_bytes(depth=-1)[0], payload),
        "odd_width": (header_bytes(width=48)[0], payload),
        "empty_file": (b"", 0),
        # a byte that maps to a dense code >= sigma would index count[] / the superblock offsets out of bounds at query time
        "dense_code_ge_sigma": (header_bytes(table=bytes(65) + bytes([6]) + bytes(190))[0], payload),
        "dense_code_255": (header_bytes(table=bytes(255) + bytes([255]))[0], payload),
    }
    ok_table = bytearray(256)
    ok_table[65], ok_table[67] = 1, 5  # the largest dense code, sigma - 1, is fine
    rc, out = checker("header", write("good_table.gdx", header_bytes(table=bytes(ok_table))[0], payload))
    assert rc == 0, out
    for name, (hdr, pay) in bad.items():
        rc, out = checker("header", write(name + ".gdx", hdr, pay))
        assert rc == 3 and out.startswith("error:"), (name, out)


def test_host_packer_stays_inside_its_buffers(checker, tmp_path):
    """pack_host.hpp (the AVX2 path of gdx_pack_queries / gdx_pack_queries_table and its byte loop) on exactly sized heap
    blocks under ASan + UBSan: pieces that start and end everywhere relative to the 32- and 128-symbol steps, a table with
    the nibble shape and one without; the packed bytes and the exception positions are those of the byte loop."""
    rng = np.random.default_rng(3)
    for name, n in (("reads.txt", 5000), ("short.txt", 37), ("empty.txt", 0), ("junk.bin", 3000)):
        p = tmp_path / name
        if name == "junk.bin":
            p.write_bytes(bytes(rng.integers(0, 256, n).astype(np.uint8)))
        else:
            p.write_bytes(bytes(rng.choice(list(b"ACGTacgtACGTACGTNQ"), n).astype(np.uint8)))
        rc, out = checker("pack", p)
        assert rc == 0 and out.startswith("ok "), (name, out)
        if n >= 3000:
            assert int(out.split()[1]) == 40 and int(out.split()[2]) > 0


def test_host_wire_expansion_equals_the_plain_construction(checker):
    """wire_host.hpp (what the drainer's workers of gdx_locate_many_alloc_layout32 run on a chunk's "found bitmap" wire) under
    ASan + UBSan on exactly sized heap blocks: offsets and hits are those of the plain construction, for one text and
    many (the coarse text table), mostly found and mostly not, chunk sizes around the 64-read words and 2048-read tiles,
    whatever the number of workers the tiles are shared among."""
    for seed, (nq, texts, found, exc_one_in) in enumerate([
            (5000, 1, 9, 3), (5000, 3, 9, 3), (4096, 200, 5, 3), (2049, 2, 1, 3), (63, 1, 9, 3), (64, 5, 0, 3), (1, 1, 9, 3),
            (20000, 40, 10, 3), (0, 1, 9, 3),
            # few exceptions: most tiles take the run-of-positions path, some do not
            (30000, 1, 9, 700), (30000, 30, 9, 700), (30000, 256, 5, 2000), (2048, 1, 9, 100000), (4097, 2, 9, 100000)]):
        rc, out = checker("wire", seed + 1, nq, texts, found, exc_one_in)
        assert rc == 0 and out.startswith("ok "), (nq, texts, found, rc, out)
        if nq >= 4096 and found < 10 and exc_one_in == 3:
            assert int(out.split()[2]) > 0
