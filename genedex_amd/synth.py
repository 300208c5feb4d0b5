"""Host-side synthetic workloads (numpy) for tests and the CPU-only configuration of BASELINE.md
section 3: i.i.d. DNA with P(N) = 0.01 from a splitmix64 stream, queries that are substrings of the
text without N ("sampled reads") or uniform random ACGT.  The large GPU workloads are generated in
HBM by libgdx.so (include/gdx_bench.h) with the same text function."""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def splitmix_at(seed: int, i: np.ndarray) -> np.ndarray:
    """output number i (0-based) of the splitmix64 stream seeded with `seed`"""
    with np.errstate(over="ignore"):
        return _mix64(np.uint64(seed) + (i.astype(np.uint64) + np.uint64(1)) * _GOLDEN)


def host_text(n: int, seed: int = 42, n_per_million: int = 10_000, start: int = 0) -> np.ndarray:
    """IO symbols start..start+n of the synthetic text; identical to gdx_synth_text_dev."""
    r = splitmix_at(seed, np.arange(start, start + n, dtype=np.uint64))
    threshold = np.uint64((n_per_million << 32) // 1_000_000)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = acgt[((r >> np.uint64(8)) & np.uint64(3)).astype(np.int64)]
    out = np.where((r >> np.uint64(32)) < threshold, np.uint8(ord("N")), out)
    return out.astype(np.uint8)


def split_lengths(total: int, n_texts: int):
    """n_texts lengths proportional to hg38 chr1..22,X,Y (Mbp), summing to `total`."""
    mbp = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51,
           156, 57][:n_texts]
    w = np.array(mbp, dtype=np.float64)
    lens = np.floor(w / w.sum() * total).astype(np.int64)
    lens[0] += total - lens.sum()
    return lens.tolist()


def host_texts(total: int, n_texts: int = 1, seed: int = 42, n_per_million: int = 10_000):
    buf = host_text(total, seed, n_per_million).tobytes()
    out, at = [], 0
    for ln in split_lengths(total, n_texts):
        out.append(buf[at:at + ln])
        at += ln
    return out


def host_queries(texts, nq: int, len_min: int, len_max: int, sampled_fraction: float, seed: int = 43):
    """-> (qbuf u8, qoff u64[nq+1]).  Sampled reads are windows of one text that contain no N."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(len_min, len_max + 1, nq).astype(np.uint64)
    qoff = np.zeros(nq + 1, dtype=np.uint64)
    np.cumsum(lens, out=qoff[1:])
    qbuf = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(qoff[-1]))].copy()
    if qbuf.size == 0:
        qbuf = np.zeros(1, dtype=np.uint8)
    arrs = [np.frombuffer(t, dtype=np.uint8) for t in texts]
    sampled = rng.random(nq) < sampled_fraction
    tid = rng.integers(0, len(texts), nq)
    u = rng.random((nq, 8))
    for q in np.flatnonzero(sampled):
        t = arrs[tid[q]]
        ln = int(lens[q])
        if t.size < ln:
            continue
        for a in range(8):
            pos = int(u[q, a] * (t.size - ln + 1))
            w = t[pos:pos + ln]
            if not (w == ord("N")).any():
                qbuf[int(qoff[q]):int(qoff[q]) + ln] = w
                break
    return qbuf, qoff
