"""A GPU host of libgdx.so without Python or torch in the process: tests/host_gpu/smoke.c (plain C, `-lgdx`) builds
the headline index and two others through gdx_index_build_ex and runs the reference's batch calls -- count_many,
cursors_for_many_queries, locate_many (/root/reference/src/lib.rs:147-246) -- on host pointers; this wrapper writes the
inputs and the oracle's answers to a file, starts the program as a fresh child process (which brings up the system's HIP
runtime on its own: the situation of a Rust host) and requires PASS."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_gpu", "smoke.c")


def _compile(out_dir):
    exe = os.path.join(str(out_dir), "gdx_host_smoke")
    libdir = os.path.join(ROOT, "genedex_amd")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", libdir, "-lgdx", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def _write_vectors(path, texts, alpha, qbuf, qoff, sa_rate, lookup_depth):
    from genedex_amd.index import pack_queries as pack_texts
    from oracle.oracle import OracleIndex

    sigma, n_searchable = alpha.num_dense_symbols(), alpha.num_searchable_dense_symbols()
    cpu = OracleIndex.build(texts, alpha.io_to_dense_table, sigma, n_searchable, sa_rate=sa_rate,
                            lookup_depth=lookup_depth, width=32)
    s, e = cpu.cursors_for_many(qbuf, qoff)
    off, t, p = cpu.locate_intervals(s, e)
    tbuf, toff = pack_texts(texts)
    nq = qoff.size - 1

    def pad(a):
        b = np.ascontiguousarray(a).tobytes()
        return b + b"\0" * (-len(b) % 8)

    with open(path, "wb") as f:
        f.write(b"GDXVEC01")
        f.write(np.array([len(texts), int(toff[-1]), nq, int(qoff[-1]), int(off[-1]), sigma, n_searchable, sa_rate,
                          lookup_depth], dtype=np.uint64).tobytes())
        f.write(np.ascontiguousarray(alpha.io_to_dense_table, dtype=np.uint8).tobytes())
        f.write(pad(toff.astype(np.uint64)))
        f.write(pad(tbuf[: int(toff[-1])]))
        f.write(pad(qoff.astype(np.uint64)))
        f.write(pad(qbuf[: int(qoff[-1])]))
        for a in (s, e, off, t, p):
            f.write(pad(a.astype(np.uint64)))
    return int(off[-1])


def test_the_c_host_compiles_and_links_against_the_c_abi(tmp_path):
    """(no GPU needed) include/gdx.h is C, and every call the program makes resolves in libgdx.so"""
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = _compile(tmp_path)
    assert os.path.exists(exe)
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libgdx.so" in needed and "torch" not in needed and "python" not in needed


@pytest.mark.gpu
@pytest.mark.parametrize("lookup_depth,narrow", [(0, "wire"), (3, "wire"), (0, "dma")])
def test_c_host_without_python_or_torch(tmp_path, lookup_depth, narrow):
    from genedex_amd import alphabet, synth

    a = alphabet.ascii_dna_with_n()
    texts = synth.host_texts(total=200_000, n_texts=3, seed=42)
    qbuf, qoff = synth.host_queries(texts, nq=30_000, len_min=18, len_max=70, sampled_fraction=0.8, seed=47)
    # reads across an N of the text (valid, not searchable: LF steps take it at depth 0, the seed kernel hands them on),
    # one-symbol reads and an empty read
    t0 = np.frombuffer(texts[0], dtype=np.uint8)
    extra, at = [], np.flatnonzero(t0 == ord("N"))
    for p in at[:40]:
        if lookup_depth == 0 and 30 <= p < t0.size - 30:
            extra.append(t0[p - 25:p + 25].tobytes())
    extra += [b"A", b"C", b"G", b"T", b""]
    qbuf = np.concatenate([qbuf[: int(qoff[-1])], np.frombuffer(b"".join(extra), dtype=np.uint8), np.zeros(8, dtype=np.uint8)])
    qoff = np.concatenate([qoff, qoff[-1] + np.cumsum([len(x) for x in extra]).astype(np.uint64)])
    vec = str(tmp_path / "vectors.bin")
    total = _write_vectors(vec, texts, a, qbuf, qoff, sa_rate=4, lookup_depth=lookup_depth)
    assert total > 20_000
    exe = _compile(tmp_path)
    # a process of its own, with nothing of this one's torch in its environment
    # (GDX_HOST_RESULTS: how the narrow locate call's results cross PCIe -- the found-bitmap wire expanded by host threads, the
    # default on hosts with four workers or more, or offsets and hits written by the device)
    env = {k: v for k, v in os.environ.items() if not k.startswith(("TORCH", "PYTORCH", "PYTHON"))}
    env["GDX_HOST_RESULTS"] = narrow
    res = subprocess.run([exe, vec], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and res.stdout.strip().endswith("PASS"), res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.count("\nok ") + res.stdout.startswith("ok ") == 3, res.stdout
