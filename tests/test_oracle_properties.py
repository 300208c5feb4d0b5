"""Property tests of the CPU oracle against the reference's naive executable definitions.

Mirrors tests/fmindex.rs:264-315 (locate/locate_many == naive search),
tests/text_with_rank_support.rs:121-135 (rank == naive columns),
src/text_with_rank_support/mod.rs:194-246 (batched rank == scalar rank) and
src/sampled_suffix_array.rs:146-195 (recover_range == full suffix array).
"""
import numpy as np
import pytest

from genedex_amd import alphabet as alph
from helpers import naive_occurrence_columns, naive_search, random_texts
from oracle.oracle import OracleIndex, naive_suffix_array, pack_queries


def sample_queries(rng, texts, n_existing=20, n_random=100, max_extent=200, max_len=20):
    existing = []
    for _ in range(n_existing):  # tests/fmindex.rs:156-186
        tid = int(rng.integers(0, len(texts)))
        t = texts[tid]
        if len(t) == 0:
            break
        pos = int(rng.integers(0, len(t)))
        extent = int(rng.integers(0, min(max_extent, len(t) - pos + 1)))
        existing.append(((tid, pos), t[pos:pos + extent]))
    randoms = []
    for _ in range(n_random):  # tests/fmindex.rs:188-205
        ln = int(rng.integers(0, max_len))
        randoms.append(bytes(b"ACGT"[i] for i in rng.integers(0, 4, ln)))
    return existing, randoms


@pytest.mark.parametrize("seed", range(24))
def test_locate_equals_naive_search(seed):
    rng = np.random.default_rng(1000 + seed)
    texts = random_texts(rng)
    sa_rate = int(rng.integers(1, 65))
    depth = int(rng.integers(0, 6))
    existing, randoms = sample_queries(rng, texts)
    naive = [naive_search(texts, q) for q in randoms]
    configs = [(alph.ascii_dna(), -32), (alph.ascii_dna_with_n(), 32), (alph.ascii_dna_iupac_as_dna_with_n(), 64)]
    for a, width in configs:
        ix = OracleIndex.build(texts, a.io_to_dense_table, a.num_dense_symbols(), a.num_searchable_dense_symbols(),
                               sa_rate=sa_rate, lookup_depth=depth, width=width)
        assert ix.n == sum(len(t) for t in texts) + len(texts)
        off, t, p = ix.locate_many([q for _, q in existing])
        for k, (hit, q) in enumerate(existing):
            many = set(zip(t[off[k]:off[k + 1]].tolist(), p[off[k]:off[k + 1]].tolist()))
            assert hit in set(ix.locate(q)) and hit in many
        off, t, p = ix.locate_many(randoms)
        for k, q in enumerate(randoms):
            many = set(zip(t[off[k]:off[k + 1]].tolist(), p[off[k]:off[k + 1]].tolist()))
            single = set(ix.locate(q))
            assert single == naive[k], (q, sa_rate, depth)
            assert many == naive[k]
            assert int(off[k + 1] - off[k]) == len(naive[k])


@pytest.mark.parametrize("seed", range(12))
def test_batched_path_equals_single_path_intervals(seed):
    rng = np.random.default_rng(2000 + seed)
    texts = random_texts(rng, len_max=3000, symbols=b"ACGTN" if seed % 2 else b"ACGT")
    a = alph.ascii_dna_with_n()
    depth = int(rng.integers(0, 5))
    ix = OracleIndex.build(texts, a.io_to_dense_table, 6, 4, sa_rate=4, lookup_depth=depth)
    existing, randoms = sample_queries(rng, texts, n_existing=150, n_random=150, max_len=40)
    queries = [q for _, q in existing if b"N" not in q] + randoms
    order = rng.permutation(len(queries))
    queries = [queries[i] for i in order]
    qbuf, qoff = pack_queries(queries)
    s1, e1, st = ix.cursors_single(qbuf, qoff)
    assert not st.any()
    for threads in (1, 3):
        s2, e2 = ix.cursors_for_many(qbuf, qoff, n_threads=threads)
        assert s1.tolist() == s2.tolist() and e1.tolist() == e2.tolist()


@pytest.mark.parametrize("seed", range(10))
def test_rank_equals_naive_columns(seed):
    rng = np.random.default_rng(3000 + seed)
    sigma = int(rng.integers(2, 257))
    n = int(rng.integers(0, 1000))
    text = rng.integers(0, sigma, n).astype(np.uint8)
    ix = OracleIndex.table_only(text, sigma)
    cols = naive_occurrence_columns(text, sigma)
    syms = rng.integers(0, sigma, 600)
    idxs = rng.integers(0, n + 1, 600)
    for c, i in zip(syms, idxs):
        assert ix.rank(int(c), int(i)) == int(cols[c, i])
    for c in range(sigma):
        assert ix.rank(c, n) == int(cols[c, n])
    for i in range(n):
        assert ix.symbol_at(i) == int(text[i])


@pytest.mark.parametrize("seed", range(10))
def test_batched_rank_equals_scalar_rank(seed):
    rng = np.random.default_rng(4000 + seed)
    sigma = int(rng.integers(3, 33))
    n = int(rng.integers(0, 1000))
    text = rng.integers(0, sigma, n).astype(np.uint8)
    ix = OracleIndex.table_only(text, sigma)
    for _ in range(20):
        m = int(rng.integers(1, 65))
        starts = rng.integers(0, n + 1, m)  # start may exceed end here (mod.rs:203)
        ends = rng.integers(0, n + 1, m)
        syms = rng.integers(0, sigma, m)
        s, e = ix.replace_many(starts, ends, syms)
        assert s.tolist() == ix.rank_many_scalar(syms, starts).tolist()
        assert e.tolist() == ix.rank_many_scalar(syms, ends).tolist()


@pytest.mark.parametrize("seed", range(16))
def test_recover_range_equals_full_suffix_array(seed):
    rng = np.random.default_rng(5000 + seed)
    texts = random_texts(rng, symbols=b"ACGTN")
    a = alph.ascii_dna_with_n()
    rate = int(rng.integers(1, 9))
    ix = OracleIndex.build(texts, a.io_to_dense_table, 6, 4, sa_rate=rate, lookup_depth=4, width=-32)
    want = naive_suffix_array(ix.dense_text)
    assert ix.full_sa.tolist() == want.tolist()
    assert ix.recover_range(0, ix.n).tolist() == want.tolist()
    # BWT and border map: bwt.rs:96-116
    text = ix.dense_text
    bwt = text[(want.astype(np.int64) - 1) % ix.n]
    assert ix.bwt.tolist() == bwt.tolist()
    keys = np.flatnonzero(bwt == 0)
    assert ix.border_keys.tolist() == keys.tolist()
    assert ix.border_vals.tolist() == want[keys].tolist()
    # border values are exactly the text start positions
    starts = [0] + (ix.sentinel_indices[:-1] + 1).tolist()
    assert sorted(ix.border_vals.tolist()) == starts


def test_repetitive_texts_suffix_array():
    for text in (np.zeros(300, dtype=np.uint8), np.tile(np.array([1, 2, 3, 4], dtype=np.uint8), 80),
                 np.array([2, 0], dtype=np.uint8), np.array([1, 1, 0, 0, 0], dtype=np.uint8)):
        a = alph.u8_until(8)
        ix = OracleIndex.build([bytes(text)], a.io_to_dense_table, a.num_dense_symbols(),
                               a.num_searchable_dense_symbols(), sa_rate=3, lookup_depth=2)
        assert ix.full_sa.tolist() == naive_suffix_array(ix.dense_text).tolist()
        assert ix.recover_range(0, ix.n).tolist() == ix.full_sa.tolist()


def test_lookup_tables_hold_plain_backward_search_intervals():
    rng = np.random.default_rng(7)
    texts = random_texts(rng, len_max=400)
    a = alph.ascii_dna_with_n()
    ix0 = OracleIndex.build(texts, a.io_to_dense_table, 6, 4, lookup_depth=0)
    ix4 = OracleIndex.build(texts, a.io_to_dense_table, 6, 4, lookup_depth=4)
    for depth in range(5):
        tab = ix4.lookup_table(depth)
        assert tab.shape[0] == 4 ** depth
        for idx in range(tab.shape[0]):
            q = bytes(b"ACGT"[(idx // 4 ** j) % 4] for j in range(depth))  # digit j = j-th symbol (lookup_table.rs:99-113)
            s, e, st = ix0.cursor_for_query(q)
            assert st == 0 and (s, e) == (int(tab[idx, 0]), int(tab[idx, 1]))
    # every query gives identical intervals (also the empty ones) with and without the table
    qs = [bytes(b"ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(0, 12)))) for _ in range(500)]
    for q in qs:
        assert ix0.cursor_for_query(q) == ix4.cursor_for_query(q)


def test_status_codes_for_invalid_and_unsearchable_symbols():
    a = alph.ascii_dna_with_n()
    texts = [b"ACGTNACGTTTGACA", b"NNACGT"]
    ix0 = OracleIndex.build(texts, a.io_to_dense_table, 6, 4, lookup_depth=0)
    ix2 = OracleIndex.build(texts, a.io_to_dense_table, 6, 4, lookup_depth=2)
    assert ix0.cursor_for_query(b"TNA")[2] == 0 and ix0.count(b"TNA") == 1  # N walks through LF steps
    assert ix2.cursor_for_query(b"TNA")[2] == 2  # N inside the lookup suffix (lookup_table.rs:154-158)
    assert ix2.cursor_for_query(b"NAC")[2] == 0 and ix2.count(b"NAC") == 2
    assert ix0.cursor_for_query(b"AXG")[2] == 1  # alphabet.rs:195-198
    # lazy validation: symbols left of the point where the interval became empty are never looked at
    assert ix0.cursor_for_query(b"XGGGGGGG")[2] == 0
