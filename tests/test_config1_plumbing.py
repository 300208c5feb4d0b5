"""BASELINE.json configs[0] / BASELINE.md workload 1: 1 MB random ascii_dna_with_n text, 10 k length-20 queries
(50 % sampled reads without N, 50 % random ACGT), count() on the CPU reference path, checked against naive search.
No GPU involved: this pins the plumbing (synthetic generators + oracle) the GPU workloads build on."""
import numpy as np

from genedex_amd import alphabet, synth
from oracle.oracle import OracleIndex


def test_one_megabyte_text_ten_thousand_queries_against_naive_search():
    a = alphabet.ascii_dna_with_n()
    texts = synth.host_texts(total=1_000_000, n_texts=1, seed=42)
    assert len(texts[0]) == 1_000_000 and 0.005 < texts[0].count(b"N") / 1e6 < 0.015
    qbuf, qoff = synth.host_queries(texts, nq=10_000, len_min=20, len_max=20, sampled_fraction=0.5, seed=43)
    ix = OracleIndex.build(texts, a.io_to_dense_table, 6, 4, sa_rate=4, lookup_depth=0, width=-32)
    assert ix.n == 1_000_001
    starts, ends = ix.cursors_for_many(qbuf, qoff, n_threads=4)
    counts = (ends - starts).astype(np.int64)
    text = texts[0]
    found = 0
    for q in range(10_000):
        query = qbuf[int(qoff[q]):int(qoff[q + 1])].tobytes()
        naive, at = 0, text.find(query)
        while at >= 0:
            naive += 1
            at = text.find(query, at + 1)
        assert counts[q] == naive, (q, query)
        found += naive > 0
    assert 4_800 < found < 5_300  # the sampled half is found, the random half (4^20 possibilities) is not
    # batched path == single-query path, and a few hits located
    s1, e1, status = ix.cursors_single(qbuf, qoff, n_threads=4)
    assert not status.any() and np.array_equal(s1, starts) and np.array_equal(e1, ends)
    off, tid, pos = ix.locate_intervals(starts[:200], ends[:200])
    for q in range(200):
        query = qbuf[int(qoff[q]):int(qoff[q + 1])].tobytes()
        for h in range(int(off[q]), int(off[q + 1])):
            assert tid[h] == 0 and text[int(pos[h]):int(pos[h]) + 20] == query
