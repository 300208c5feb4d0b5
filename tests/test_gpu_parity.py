"""GPU parity tests: the HIP path (through the C ABI of include/gdx.h) against the CPU oracle, the
reference's known-answer vectors and naive search.  Bit-exact everywhere (integer work)."""
import numpy as np
import pytest

from genedex_amd import alphabet as alph
from helpers import alphabet_by_name, as_bytes, naive_occurrence_columns, naive_search, random_texts
from oracle.oracle import OracleIndex, pack_queries

pytestmark = pytest.mark.gpu

WIDTH = {"i32": -32, "u32": 32, "i64": 64}


# build / query options (gdx_build_options_t, gdx_query_options_t) applied to every index gpu_index() makes;
# the search_variant fixture sweeps them
_BUILD_OPTIONS = {}
_QUERY_OPTIONS = {}


def gpu_index(texts, a, sa_rate=4, depth=0, storage="u32", **build_options):
    from genedex_amd import FmIndexConfig

    opts = dict(_BUILD_OPTIONS)
    opts.update(build_options)
    ix = (FmIndexConfig(storage).suffix_array_sampling_rate(sa_rate).lookup_table_depth(depth)
          .acceleration_structures(**opts).construct_index(texts, a))
    if _QUERY_OPTIONS:
        ix.set_query_options(**_QUERY_OPTIONS)
    return ix


def cpu_index(texts, a, sa_rate=4, depth=0, storage="u32"):
    return OracleIndex.build(texts, a.io_to_dense_table, a.num_dense_symbols(), a.num_searchable_dense_symbols(),
                             sa_rate=sa_rate, lookup_depth=depth, width=WIDTH[storage])


def both(texts, a, **kw):
    cpu_kw = {k: v for k, v in kw.items() if k in ("sa_rate", "depth", "storage")}
    return gpu_index(texts, a, **kw), cpu_index(texts, a, **cpu_kw)


# ------------------------------------------------------------------------------------------------
# the reference's own known-answer tests, through the drop-in API

def test_locate_kats(kat):
    for case in kat["locate"]:
        a = alphabet_by_name(case["alphabet"])
        for w in case["widths"]:
            storage = {32: "u32", -32: "i32"}[w]
            ix = gpu_index([as_bytes(t) for t in case["texts"]], a, sa_rate=case["sa_rate"],
                           depth=case["lookup_depth"], storage=storage)
            q = as_bytes(case["query"])
            want = {tuple(h) for h in case["hits"]}
            assert {tuple(h) for h in ix.locate(q)} == want, case["source"]
            assert {tuple(h) for h in ix.locate_many([q])[0]} == want, case["source"]
            assert ix.count(q) == len(want)


def test_count_cursor_kats(kat):
    for case in kat["count"]:
        a = alphabet_by_name(case["alphabet"])
        ix = gpu_index([as_bytes(t) for t in case["texts"]], a, sa_rate=case["sa_rate"], depth=case["lookup_depth"],
                       storage="i32")
        assert ix.count(as_bytes(case["query"])) == case["count"], case["source"]
    for case in kat["cursor"]:
        a = alphabet_by_name(case["alphabet"])
        ix = gpu_index([as_bytes(t) for t in case["texts"]], a, storage="i32")
        cur = ix.cursor_for_query(as_bytes(case["query"]))
        assert cur.count() == case["count"]
        cur.extend_query_front(as_bytes(case["extend_front"]))
        assert cur.count() == case["count_after"]
        assert len(cur.locate()) == case["count_after"]
    for case in kat["locate_many_runs"]:
        a = alphabet_by_name(case["alphabet"])
        ix = gpu_index([as_bytes(t) for t in case["texts"]], a, sa_rate=case["sa_rate"], storage="i32")
        res = ix.locate_many([as_bytes(q) for q in case["queries"]])
        assert [len(r) for r in res][2:] == [2, 1]


def test_rank_kats(kat):
    from genedex_amd import FmIndex

    for case in kat["rank"] + kat["rank_vs_naive"]:
        if "dense_text" in case:
            text = np.array(case["dense_text"], dtype=np.uint8)
        else:
            text = np.full(case["dense_text_repeat"]["times"], case["dense_text_repeat"]["symbol"], dtype=np.uint8)
        sigma = case["sigma"]
        # operator-level construct(text, sigma): import the oracle-built planes as a bare table
        o = OracleIndex.table_only(text, sigma)
        a = alph.Alphabet.from_io_symbols(bytes(range(1, sigma)), 0)  # any alphabet with sigma dense symbols
        count = np.zeros(sigma + 1, dtype=np.uint64)
        count[1:] = np.cumsum(np.bincount(text, minlength=sigma))
        if text.size == 0 or count[1] == 0:
            continue  # from_parts needs >= 1 sentinel to describe texts; covered by test_rank_everywhere below
        n_texts = int(count[1])
        sent = np.flatnonzero(text == 0)
        if sent[-1] != text.size - 1:
            continue
        samples = np.zeros(-(-text.size // 4), dtype=np.uint32)
        # the planes ARE the BWT here, so the rows that hold the sentinel are the positions of symbol 0
        ix = FmIndex.from_parts(count, o.blocks, text.size, samples, 4, sent, np.zeros(n_texts), sent, a)
        cols = naive_occurrence_columns(text, sigma)
        step = 1 if text.size < 3000 else 53
        idx = np.array(sorted(set(range(0, text.size + 1, step)) | {text.size}), dtype=np.uint64)
        for c in range(sigma):
            got = ix.rank_many(np.full(idx.size, c, dtype=np.uint8), idx)
            assert got.tolist() == cols[c, idx.astype(np.int64)].tolist(), (case["source"], c)
        pos = np.arange(0, text.size, step, dtype=np.uint64)
        assert ix.symbol_at_many(pos).tolist() == text[pos.astype(np.int64)].tolist()


# ------------------------------------------------------------------------------------------------
# the built index itself equals the oracle's, array by array

@pytest.mark.parametrize("seed", range(8))
def test_index_arrays_equal_oracle(seed):
    rng = np.random.default_rng(100 + seed)
    a = [alph.ascii_dna(), alph.ascii_dna_with_n(), alph.ascii_dna_iupac(), alph.u8_until(40)][seed % 4]
    symbols = [b"ACGT", b"ACGTN", b"ACGTNRYKMSWBDHV", bytes(range(41))][seed % 4]
    texts = random_texts(rng, len_max=[1500, 70000, 3000, 3000][seed % 4], symbols=symbols)
    rate = int(rng.integers(1, 9))
    depth = int(rng.integers(0, 4)) if seed % 4 < 3 else 1
    g, c = both(texts, a, sa_rate=rate, depth=depth)
    assert g.total_text_len() == c.n and g.num_texts() == c.num_texts
    assert g.export_count().tolist() == c.count_array.tolist()
    assert g.export_sentinel_indices().tolist() == c.sentinel_indices.tolist()
    assert g.export_bwt().tolist() == c.bwt.tolist()
    assert g.export_sa_samples().tolist() == c.sa_samples.tolist()
    gk, gv = g.export_borders()
    assert gk.tolist() == c.border_keys.tolist() and gv.tolist() == c.border_vals.tolist()
    blocks, bo, sbo = g.export_condensed_table()
    assert blocks.tolist() == c.blocks.tolist()
    assert bo.tolist() == c.block_offsets.tolist()
    assert sbo.tolist() == c.superblock_offsets.tolist()
    for d in range(depth + 1):
        assert g.export_lookup_table(d).tolist() == c.lookup_table(d).tolist(), d


def test_repetitive_and_degenerate_texts():
    a = alph.ascii_dna_with_n()
    cases = [[b""], [b"", b""], [b"A"], [b"A" * 70000], [b"ACGT" * 5000, b"ACGT" * 5000], [b"N" * 300, b"", b"N"],
             [b"AC" * 129, b"CA" * 64, b""]]
    for texts in cases:
        for rate, depth in ((1, 2), (3, 0), (4, 2)):
            g, c = both(texts, a, sa_rate=rate, depth=depth)
            assert g.export_bwt().tolist() == c.bwt.tolist(), texts[0][:8]
            assert g.export_sa_samples().tolist() == c.sa_samples.tolist()
            qs = [b"", b"A", b"AC", b"CA", b"ACGT", b"ACGTACGTA", b"TTTT", b"NAC"]
            if depth == 0:  # a non-searchable symbol may only be consumed by LF steps
                qs += [b"N", b"NN", b"ANN"]
            off, t, p, _ = g.locate_raw(*pack_queries(qs))
            co, ct, cp = c.locate_many(qs)
            assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist()


# ------------------------------------------------------------------------------------------------
# operator level

@pytest.mark.parametrize("seed", range(4))
def test_rank_everywhere(seed):
    rng = np.random.default_rng(200 + seed)
    a = [alph.ascii_dna_with_n(), alph.u8_until(26)][seed % 2]
    texts = random_texts(rng, len_max=[70000, 2000][seed % 2], symbols=[b"ACGTN", bytes(range(27))][seed % 2])
    g, c = both(texts, a)
    n, sigma = c.n, c.sigma
    cols = naive_occurrence_columns(c.bwt, sigma)
    idx = np.arange(n + 1, dtype=np.uint64)
    for s in range(sigma):
        got = g.rank_many(np.full(n + 1, s, dtype=np.uint8), idx)
        assert np.array_equal(got, cols[s])
    assert np.array_equal(g.symbol_at_many(np.arange(n, dtype=np.uint64)), c.bwt)
    from genedex_amd import GdxError

    with pytest.raises(GdxError):  # mod.rs:107-108
        g.rank_many([sigma], [0])
    with pytest.raises(GdxError):
        g.rank_many([0], [n + 1])
    with pytest.raises(GdxError):  # condensed.rs:344
        g.symbol_at_many([n])


# ------------------------------------------------------------------------------------------------
# search: intervals bit-identical to the reference's batched path, incl. empty results

def mixed_queries(rng, texts, n_sampled, n_random, max_len, allow_n=False):
    qs = []
    nonempty = [t for t in texts if len(t) > 0]
    for _ in range(n_sampled):
        if not nonempty:
            break
        t = nonempty[int(rng.integers(0, len(nonempty)))]
        pos = int(rng.integers(0, len(t)))
        ln = int(rng.integers(0, min(max_len, len(t) - pos) + 1))
        q = t[pos:pos + ln]
        if allow_n or b"N" not in q:
            qs.append(q)
    for _ in range(n_random):
        ln = int(rng.integers(0, max_len))
        qs.append(bytes(b"ACGT"[i] for i in rng.integers(0, 4, ln)))
    order = rng.permutation(len(qs))
    return [qs[i] for i in order]


_VARIANTS = {
    # name: (query options, build options)
    # the library's defaults (round 6): seed table + text units + full and inverse suffix array + pair lines + top table, no jump
    # table -- the shape bench.py's headline runs on; and the same with the seed table switched off at query time
    "default": ({}, {}),
    "default-no-seed": (dict(search_seed=False), {}),
    # the tables of rounds 1-3: 32-byte jump entries, top table sized from the text
    "pair": (dict(search_kernel="pair"), dict(jump_entry_bytes=32)),
    "pair-8lanes": (dict(search_kernel="pair", search_lanes=8, search_defer_after=1), dict(jump_entry_bytes=32)),
    "pair-park-all": (dict(search_kernel="pair", search_defer_after=1), dict(jump_entry_bytes=0, top_table_depth=6)),
    "pair-jump16": (dict(search_kernel="pair", search_defer_after=0), dict(jump_entry_bytes=16)),
    "pair-narrow": (dict(search_kernel="pair"), dict(jump_entry_bytes=8, top_table_depth=0)),
    # forced top depths: on these small texts most deep entries are empty, so the fall-back to the ordinary
    # path runs constantly
    "pair-top4": (dict(search_kernel="pair", locate_jump_walk=False), dict(top_table_depth=4)),
    "pair-top9": (dict(search_kernel="pair", load_policy=1), dict(top_table_depth=9)),
    "pair-no-fast": (dict(search_kernel="pair", search_fast=False), dict(jump_entry_bytes=32)),
    # the general kernel alone (no slim kernel in front of it) for exact intervals and cursors too
    "pair-general-only": (dict(search_kernel="pair", search_fast=False, search_exact=False), dict(jump_entry_bytes=32)),
    "pair-fast": (dict(search_kernel="pair", search_fast=1), dict(jump_entry_bytes=32)),  # (the default asks the index: wide_permille)
    "pair-fast-wide": (dict(search_kernel="pair", search_fast=2), dict(jump_entry_bytes=32)),
    "pair-lines-only": (dict(search_kernel="pair", length_schedule=0), dict(jump_entry_bytes=0, top_table_depth=0)),
    "quad": (dict(search_kernel="quad"), dict(pair_lines=False)),
    # text units instead of a jump table: count / locate searches compare the rest of the query with the text at SA[row]
    "verify-sa": (dict(search_kernel="pair"), dict(pair_lines=False, jump_entry_bytes=0, top_table_depth=7,
                                                   full_suffix_array=True, text_units=True)),
    "verify-walk": (dict(search_kernel="pair"), dict(pair_lines=False, jump_entry_bytes=0, top_table_depth=5, text_units=True)),
    "verify-pairs": (dict(search_kernel="pair"), dict(jump_entry_bytes=0, top_table_depth=8, full_suffix_array=True,
                                                      text_units=True)),
    # seed table in front (count / locate searches): a bucket fetch answers the last k symbols, and reads whose k-mer
    # occurs once are answered by its entry alone
    "seed-sa": (dict(search_kernel="pair"), dict(pair_lines=False, jump_entry_bytes=0, top_table_depth=0,
                                                 full_suffix_array=True, seed_symbols=12)),
    "seed-walk": (dict(search_kernel="quad"), dict(pair_lines=False, jump_entry_bytes=0, top_table_depth=0, seed_symbols=True)),
    # with everything else present too; a full table (every bucket overflows into its neighbours)
    "seed-all": (dict(search_kernel="pair"), dict(jump_entry_bytes=16, seed_symbols=9, seed_load_percent=100)),
    # + inverse suffix array: exact intervals of reads that occur once come from the seed entry and one ISA fetch
    "seed-isa": (dict(search_kernel="pair"), dict(seed_symbols=10, inverse_suffix_array=True)),
    "seed-isa-nojump": (dict(search_kernel="pair"), dict(jump_entry_bytes=0, top_table_depth=0, seed_symbols=True,
                                                         inverse_suffix_array=True)),
    # the reference's own occurrence table in its four variants, queried as it is (one lane per query)
    # (all four, bit for bit and at the block boundaries: tests/test_gpu_seed.py::test_reference_table_layouts_...)
    "ref-condensed512": (dict(search_kernel="lane"), dict(reference_table_layout="condensed512")),
    "ref-flat64": (dict(search_kernel="lane", locate_kernel="lane"), dict(reference_table_layout="flat64")),
    "lane": (dict(search_kernel="lane", locate_kernel="lane"), dict(jump_entry_bytes=32)),
}


@pytest.fixture(params=list(_VARIANTS))
def search_variant(request):
    """Every search kernel variant and index acceleration structure must give the same answers."""
    query, build = _VARIANTS[request.param]
    _QUERY_OPTIONS.clear()
    _QUERY_OPTIONS.update(query)
    _BUILD_OPTIONS.clear()
    _BUILD_OPTIONS.update(build)
    yield request.param
    _QUERY_OPTIONS.clear()
    _BUILD_OPTIONS.clear()


@pytest.mark.parametrize("seed", range(10))
def test_cursors_equal_oracle_batched_path(seed, search_variant):
    rng = np.random.default_rng(300 + seed)
    a = alph.ascii_dna_with_n()
    texts = random_texts(rng, len_max=20000, symbols=b"ACGTN" if seed % 2 else b"ACGT")
    depth = [0, 1, 3, 5, 8][seed % 5]
    g, c = both(texts, a, depth=depth)
    qs = mixed_queries(rng, texts, 700, 700, 60)
    qbuf, qoff = pack_queries(qs)
    s, e, st = g.cursors_raw(qbuf, qoff)
    cs, ce = c.cursors_for_many(qbuf, qoff)
    assert not st.any()
    assert s.tolist() == cs.tolist() and e.tolist() == ce.tolist()
    counts, _ = g.count_raw(qbuf, qoff)
    assert counts.tolist() == (ce - cs).tolist()
    # and the counts are right in the first place
    fold = a.io_to_dense_table
    for q, cnt in list(zip(qs, counts))[:200]:
        assert int(cnt) == len(naive_search(texts, q, fold=fold))


@pytest.mark.parametrize("seed", range(6))
def test_locate_equals_oracle_and_naive(seed):
    rng = np.random.default_rng(400 + seed)
    a = [alph.ascii_dna(), alph.ascii_dna_with_n(), alph.ascii_dna_iupac_as_dna_with_n()][seed % 3]
    texts = random_texts(rng, len_max=1500)
    rate = int(rng.integers(1, 65))
    depth = int(rng.integers(0, 6))
    storage = ["i32", "u32", "i64"][seed % 3]
    g, c = both(texts, a, sa_rate=rate, depth=depth, storage=storage)
    qs = mixed_queries(rng, texts, 40, 160, 20) + [b""]
    off, t, p, st = g.locate_raw(*pack_queries(qs))
    co, ct, cp = c.locate_many(qs)
    # same hits in the same (suffix array) order
    assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist()
    for k, q in enumerate(qs):
        got = set(zip(t[off[k]:off[k + 1]].tolist(), p[off[k]:off[k + 1]].tolist()))
        assert got == naive_search(texts, q), q
    # Python mirror of the reference API
    hits = g.locate_many(qs[:20])
    for k in range(20):
        assert {tuple(h) for h in hits[k]} == naive_search(texts, qs[k])
        assert {tuple(h) for h in g.locate(qs[k])} == naive_search(texts, qs[k])


def test_status_codes_match_the_reference_panics(search_variant):
    a = alph.ascii_dna_with_n()
    texts = [b"ACGTNACGTTTGACA", b"NNACGT"]
    qs = [b"TNA", b"NAC", b"AXG", b"XGGGGGGG", b"ACGT", b"", b"Z"]
    for depth in (0, 2):
        g, c = both(texts, a, depth=depth)
        qbuf, qoff = pack_queries(qs)
        s, e, st = g.cursors_raw(qbuf, qoff, strict=False)
        cs, ce, cst = c.cursors_single(qbuf, qoff)
        assert st.tolist() == cst.tolist()
        ok = st == 0
        assert s[ok].tolist() == cs[ok].tolist() and e[ok].tolist() == ce[ok].tolist()
        from genedex_amd import GdxError

        with pytest.raises(GdxError):
            g.cursors_raw(qbuf, qoff)  # strict: a reference panic is an error of the call


@pytest.mark.parametrize("fast", [0, 1, 2])
def test_lookup_deeper_than_top_table_with_unsearchable_symbol(fast):
    """The fast path must not take a query whose N sits between the top table's depth and the configured lookup
    depth: the reference rejects it (lookup_table.rs:154-158 -> GDX_Q_UNSEARCHABLE_IN_LOOKUP), whatever
    search_fast says, for the count, the record (locate) and the interval entry points alike."""
    rng = np.random.default_rng(1234)
    a = alph.ascii_dna_with_n()
    texts = random_texts(rng, len_max=20000, symbols=b"ACGT")
    g, c = both(texts, a, depth=6, top_table_depth=4)
    g.set_query_options(search_kernel="pair", search_fast=fast)
    qs = []
    for q in mixed_queries(rng, texts, 300, 50, 40):
        if len(q) >= 24:
            k = int(rng.integers(0, 3))
            if k == 0:  # an N right above the top table's four symbols, inside the six of the lookup table
                q = q[:-5] + b"N" + q[-4:]
            elif k == 1:  # an N outside both: consumed by an LF step, count 0, status OK
                q = q[:-9] + b"N" + q[-8:]
        qs.append(q)
    qbuf, qoff = pack_queries(qs)
    cs, ce, cst = c.cursors_single(qbuf, qoff)
    assert (cst == 2).sum() > 20 and (cst == 0).sum() > 100
    s, e, st = g.cursors_raw(qbuf, qoff, strict=False)
    counts, st_count = g.count_raw(qbuf, qoff, strict=False)
    off, t, p, st_loc = g.locate_raw(qbuf, qoff, strict=False)
    assert st.tolist() == cst.tolist() and st_count.tolist() == cst.tolist() and st_loc.tolist() == cst.tolist()
    ok = cst == 0
    assert s[ok].tolist() == cs[ok].tolist() and e[ok].tolist() == ce[ok].tolist()
    assert counts[ok].tolist() == (ce - cs)[ok].tolist() and not counts[~ok].any()
    co, ct, cp = c.locate_intervals(np.where(ok, cs, 0), np.where(ok, ce, 0))
    assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist()


def test_cursor_api():
    rng = np.random.default_rng(9)
    a = alph.ascii_dna_with_n()
    texts = random_texts(rng, len_max=5000, symbols=b"ACGTN")
    g, c = both(texts, a, depth=3)
    cur = g.cursor_empty()
    assert cur.interval() == (0, c.n)
    # batched extension equals repeated scalar extension on the oracle
    m = 500
    starts = np.zeros(m, dtype=np.uint64)
    ends = np.full(m, c.n, dtype=np.uint64)
    want = [(0, c.n)] * m
    for step in range(14):
        syms = np.frombuffer(bytes(b"ACGTNacgtn"[i] for i in rng.integers(0, 10, m)), dtype=np.uint8)
        starts, ends, st = g.extend_front_raw(starts, ends, syms)
        want = [c.extend_front(s, e, int(sym))[:2] for (s, e), sym in zip(want, syms)]
        assert not st.any()
        assert list(zip(starts.tolist(), ends.tolist())) == want
    off, t, p = g.locate_intervals_raw(starts, ends)
    co, ct, cp = c.locate_intervals(starts, ends)
    assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist()
    _, _, st = g.extend_front_raw([0], [c.n], [ord("X")], strict=False)
    assert st.tolist() == [1]


def test_from_parts_round_trip():
    from genedex_amd import FmIndex

    rng = np.random.default_rng(11)
    a = alph.ascii_dna_with_n()
    texts = random_texts(rng, len_max=30000, symbols=b"ACGTN")
    c = cpu_index(texts, a, sa_rate=5, depth=2)
    g = FmIndex.from_parts(c.count_array, c.blocks, c.n, c.sa_samples, 5, c.border_keys, c.border_vals,
                           c.sentinel_indices, a, lookup_depth=2)
    qs = mixed_queries(rng, texts, 300, 300, 40)
    off, t, p, _ = g.locate_raw(*pack_queries(qs))
    co, ct, cp = c.locate_many(qs)
    assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist()
    blocks, bo, sbo = g.export_condensed_table()
    assert blocks.tolist() == c.blocks.tolist() and bo.tolist() == c.block_offsets.tolist()
    assert sbo.tolist() == c.superblock_offsets.tolist()


def test_invalid_construction_arguments():
    from genedex_amd import FmIndexConfig, GdxError

    with pytest.raises(GdxError):  # alphabet.rs:195-198 while encoding the text
        FmIndexConfig("i32").construct_index([b"ACGX"], alph.ascii_dna())
    with pytest.raises(GdxError):  # construction/mod.rs:303
        FmIndexConfig("i32").construct_index([], alph.ascii_dna())


def test_medium_text_against_oracle(search_variant):
    """4 Mi symbols, 200k queries: full equality of intervals and hits with the CPU restatement."""
    from genedex_amd import synth

    a = alph.ascii_dna_with_n()
    texts = synth.host_texts(total=1 << 22, n_texts=3, seed=42)
    g, c = both(texts, a, depth=6)
    assert g.export_bwt().tobytes() == c.bwt.tobytes()
    qbuf, qoff = synth.host_queries(texts, nq=200_000, len_min=50, len_max=50, sampled_fraction=0.9, seed=43)
    s, e, st = g.cursors_raw(qbuf, qoff)
    cs, ce = c.cursors_for_many(qbuf, qoff, n_threads=4)
    assert np.array_equal(s, cs) and np.array_equal(e, ce)
    off, t, p, _ = g.locate_raw(qbuf, qoff)
    co, ct, cp = c.locate_intervals(cs, ce, n_threads=4)
    assert np.array_equal(off, co) and np.array_equal(t, ct) and np.array_equal(p, cp)
    assert (e - s)[:1000].sum() > 0


_ALPHABETS = {
    # name: (constructor, symbols texts are drawn from, symbols random queries are drawn from)
    "ascii_dna": (alph.ascii_dna, b"ACGTacgt", b"ACGT"),
    "ascii_dna_with_n": (alph.ascii_dna_with_n, b"ACGTNacgtn", b"ACGTN"),
    "ascii_dna_iupac": (alph.ascii_dna_iupac, b"ACGTNRYKMSWBDHVacgtnry", b"ACGTNRYKMSWBDHV"),
    "ascii_dna_iupac_as_dna_with_n": (alph.ascii_dna_iupac_as_dna_with_n, b"ACGTNRYKMSWBDHVacgt", b"ACGTN"),
    "ascii_amino_acid": (alph.ascii_amino_acid, b"ACDEFGHIKLMNPQRSTVWYacdef", b"ACDEFGHIKLMNPQRSTVWY"),
    "ascii_amino_acid_iupac": (alph.ascii_amino_acid_iupac, b"ACDEFGHIKLMNPQRSTVWYBZXacd", b"ACDEFGHIKLMNPQRSTVWYBZX"),
    "u8_until": (lambda: alph.u8_until(200), bytes(range(201)), bytes(range(201))),
    "ascii_printable": (alph.ascii_printable, bytes(range(0x20, 0x7f)), bytes(range(0x20, 0x7f))),
}


@pytest.mark.parametrize("name", list(_ALPHABETS))
def test_every_reference_alphabet(name):
    """alphabet.rs:251-345: all eight stock alphabets (case folding, ambiguity groups, non-searchable tails,
    rank-line layout for sigma <= 8 and the generic planes above) through build, search and locate."""
    make, text_symbols, query_symbols = _ALPHABETS[name]
    a = make()
    rng = np.random.default_rng(sum(name.encode()))
    texts = [bytes(text_symbols[i] for i in rng.integers(0, len(text_symbols), int(rng.integers(0, 4000))))
             for _ in range(5)]
    for depth in (0, 2):
        g, c = both(texts, a, depth=depth, sa_rate=3)
        assert g.export_bwt().tolist() == c.bwt.tolist()
        qs = []
        for _ in range(400):
            t = texts[int(rng.integers(0, len(texts)))]
            if len(t) > 0:
                pos = int(rng.integers(0, len(t)))
                qs.append(t[pos:pos + int(rng.integers(0, 12))])
            qs.append(bytes(query_symbols[i] for i in rng.integers(0, len(query_symbols), int(rng.integers(0, 6)))))
        qbuf, qoff = pack_queries(qs)
        s, e, st = g.cursors_raw(qbuf, qoff, strict=False)
        cs, ce, cst = c.cursors_single(qbuf, qoff)
        assert st.tolist() == cst.tolist()
        ok = st == 0
        assert ok.sum() > len(qs) // 2
        assert s[ok].tolist() == cs[ok].tolist() and e[ok].tolist() == ce[ok].tolist()
        off, t_, p_, _ = g.locate_raw(qbuf, qoff, strict=False)
        co, ct, cp = c.locate_intervals(np.where(ok, cs, 0), np.where(ok, ce, 0))
        assert off.tolist() == co.tolist() and t_.tolist() == ct.tolist() and p_.tolist() == cp.tolist()
        # a naive count on the case-folded / group-folded dense texts pins a few of them independently of the oracle
        dense_texts = [a.encode(t) for t in texts]
        for q in [q for q, good in zip(qs, ok) if good and len(q) > 0][:40]:
            dq = a.encode(q).tobytes()
            want = sum(sum(1 for i in range(len(dt) - len(dq) + 1) if dt[i:i + len(dq)].tobytes() == dq)
                       for dt in dense_texts)
            assert g.count(q) == want, (name, q)


def test_many_short_texts_and_sentinel_crossings(search_variant):
    """Thousands of tiny texts: pair steps and 8-symbol jumps constantly run into text borders, N and sentinels."""
    rng = np.random.default_rng(77)
    a = alph.ascii_dna_with_n()
    texts = [bytes(b"ACGTN"[i] for i in rng.choice(5, int(rng.integers(0, 60)), p=[.24, .24, .24, .24, .04]))
             for _ in range(3000)]
    for depth, rate in ((0, 4), (3, 3)):
        g, c = both(texts, a, depth=depth, sa_rate=rate)
        qs = mixed_queries(rng, texts, 3000, 500, 45, allow_n=(depth == 0))
        qbuf, qoff = pack_queries(qs)
        s, e, st = g.cursors_raw(qbuf, qoff, strict=False)
        cs, ce, cst = c.cursors_single(qbuf, qoff)
        assert st.tolist() == cst.tolist()
        ok = st == 0
        assert s[ok].tolist() == cs[ok].tolist() and e[ok].tolist() == ce[ok].tolist()
        off, t, p, _ = g.locate_raw(qbuf, qoff, strict=False)
        co, ct, cp = c.locate_intervals(np.where(ok, cs, 0), np.where(ok, ce, 0))
        assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist()


def test_long_repeats_and_long_queries(search_variant):
    """Highly repetitive text (many doubling rounds in the suffix sorter, wide intervals deep into the search)
    and queries much longer than the jump width."""
    rng = np.random.default_rng(78)
    a = alph.ascii_dna_with_n()
    unit = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 997))
    t1 = bytearray(unit * 300)
    for pos in rng.integers(0, len(t1), 200):
        t1[pos] = b"ACGT"[int(rng.integers(0, 4))]
    texts = [bytes(t1), unit * 3, b"A" * 5000]
    g, c = both(texts, a, depth=4, sa_rate=8)
    assert g.export_bwt().tobytes() == c.bwt.tobytes()
    qs = mixed_queries(rng, texts, 1500, 100, 400)
    qbuf, qoff = pack_queries(qs)
    s, e, st = g.cursors_raw(qbuf, qoff)
    cs, ce = c.cursors_for_many(qbuf, qoff, n_threads=4)
    assert s.tolist() == cs.tolist() and e.tolist() == ce.tolist()
    few = (ce - cs) < 2000
    off, t, p = g.locate_intervals_raw(cs[few], ce[few])
    co, ct, cp = c.locate_intervals(cs[few], ce[few], n_threads=4)
    assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist()


@pytest.mark.parametrize("budget,want", [(1, (0, 0)), (300_000, None), (None, (32, None))])
def test_acceleration_structures_shrink_to_the_memory_budget(budget, want):
    """The jump and top tables are optional: they shrink to gdx_build_options_t.aux_budget_bytes (or the free HBM) and
    the answers do not change; gdx_index_aux reports what was wanted and what was built."""
    from genedex_amd.device import DeviceEngine

    rng = np.random.default_rng(41)
    a = alph.ascii_dna_with_n()
    texts = random_texts(rng, len_max=20000, symbols=b"ACGT")
    g, c = both(texts, a, aux_budget_bytes=budget)
    full = g.aux()
    aux = DeviceEngine(g).aux_info()
    n = g.total_text_len()
    if budget is None:
        # every option at its default and room for it: the default shape (fm_index.hip build_aux) -- seed table, text units,
        # full and inverse suffix array, pair lines, a top table, no jump table
        assert aux["default_shape"] and aux["seed"]["k"] >= 8 and aux["full_suffix_array"] and aux["inverse_suffix_array"] \
            and aux["text_units"] and aux["pair_lines"] and aux["jump_entry_bytes"] == 0 and 0 < aux["top_table_depth"] <= 14
        g, c = both(texts, a, aux_budget_bytes=budget, jump_entry_bytes=32)  # the tables of rounds 1-3, asked for
        full = g.aux()
        aux = DeviceEngine(g).aux_info()
    # (a budget the default shape does not fit: the options mean the tables, which shrink)
    assert not aux["default_shape"] and aux["seed"]["k"] == 0
    assert full["wanted_jump_entry_bytes"] == 32
    assert (budget is None) == ((full["wanted_jump_entry_bytes"], full["wanted_top_table_depth"])
                                == (full["jump_entry_bytes"], full["top_table_depth"]))
    if want is not None:
        if want[0] is not None:
            assert aux["jump_entry_bytes"] == want[0]
        if want[1] is not None:
            assert aux["top_table_depth"] == want[1]
    else:  # 300 kB: whatever was kept fits
        used = aux["jump_entry_bytes"] * n + (8 * 4 ** aux["top_table_depth"] if aux["top_table_depth"] else 0)
        assert used <= 300_000 and full["aux_bytes"] <= 300_016  # allocation padded to 16 bytes
    qs = mixed_queries(rng, texts, 600, 300, 70)
    qbuf, qoff = pack_queries(qs)
    s_, e_, st = g.cursors_raw(qbuf, qoff)
    cs, ce = c.cursors_for_many(qbuf, qoff)
    assert s_.tolist() == cs.tolist() and e_.tolist() == ce.tolist()
    off, t, p, _ = g.locate_raw(qbuf, qoff)
    co, ct, cp = c.locate_many(qs)
    assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist()


def test_genome_like_text_properties():
    """A 32 M-symbol text with segmental duplications, tandem repeats, poly-A and long runs of N (tools/genome_like.py):
    many doubling rounds in the suffix sorter, intervals of millions of rows, wide intervals deep into the search.
    Checked without the oracle: every sampled read is found, located hits spell their query, counts equal a scan."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "genome_like.py")
    spec = importlib.util.spec_from_file_location("genome_like", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run(1 << 25, 200_000)
    assert res["build_stats"]["sa_rounds"] >= 10
    assert res["queries_with_status"] == 0
    assert res["queries_found"] >= 180_000  # the 90 % drawn from the text
    assert res["max_count"] > 10_000
    assert res["hits_checked"] > 0 and res["hits_checked"] == res["hits_spelling_their_query"]
    assert res["counts_checked_by_scan"] == res["counts_equal_scan"] == 9


def test_full_size_properties_workload2():
    """BASELINE workload 2 at full size (256 MB text, 10 M len-50 reads) through size-independent properties:
    every sampled read is found, every reported hit spells its query in the text, count == number of hits,
    and a 200 k prefix equals the CPU oracle bit for bit."""
    import torch

    from genedex_amd.device import DeviceEngine, DeviceQueries, build_index_from_device_text, synth_text

    dev = torch.device("cuda", 0)
    total, nq = 1 << 28, 10_000_000
    a = alph.ascii_dna_with_n()
    io_text = synth_text(total, device=dev)
    g = build_index_from_device_text(io_text, [total], a, index_storage="i32")
    q = DeviceQueries.synth(io_text, [total], nq, 50, 50, 900_000)
    eng = DeviceEngine(g)
    out = eng.alloc_outputs(nq, hint=True)
    eng.search(q, out)
    eng.hit_offsets(out, nq)
    torch.cuda.synchronize()
    total_hits = int(out["hit_offsets"][nq].item())
    assert int((out["status"] != 0).sum().item()) == 0
    hinted = int(((out["hint"] & 0xffffffff) != 0xffffffff).sum().item())
    if eng.aux_info()["jump_entry_bytes"] == 32:
        # 32-byte jump entries carry SA[row]: locate resolves any row with one fetch, so the exact-interval search
        # leaves no sampled-row hints (they would save nothing)
        assert hinted == 0
    elif eng.aux_info()["jump_entry_bytes"] != 0:
        assert hinted > nq // 3  # most one-occurrence reads pass through a sampled row while they jump
    # (the default shape has no jump table: SA[row] is one fetch of the full suffix array, hints are not needed)
    counts = (out["end"] - out["start"]).to(torch.int64)
    assert int(counts.sum().item()) == total_hits
    found = int((counts > 0).sum().item())
    assert 0.895 * nq < found < 0.905 * nq  # 90 % sampled reads (all must be found) + a few random ones
    hits = torch.empty((total_hits, 2), dtype=torch.int32, device=dev)
    ws = torch.empty(eng.locate_workspace_bytes(total_hits), dtype=torch.uint8, device=dev)
    eng.locate(out, nq, total_hits, hits, ws)
    torch.cuda.synchronize()
    # every hit spells its query (checked for all hits, on the device)
    hq = torch.repeat_interleave(torch.arange(nq, device=dev), counts)
    assert hq.numel() == total_hits
    pos = hits[:, 1].to(torch.int64)
    assert bool((hits[:, 0] == 0).all()) and bool(((pos >= 0) & (pos + 50 <= total)).all())
    for j in range(0, 50, 10):
        col = torch.arange(j, j + 10, device=dev)
        same = io_text[pos[:, None] + col[None, :]] == q.qbuf[(q.qoff[hq])[:, None] + col[None, :]]
        assert bool(same.all())
    # the walk without the search's hints gives the same hits
    plain = {k: v for k, v in out.items() if k != "hint"}
    hits2 = torch.empty_like(hits)
    eng.locate(plain, nq, total_hits, hits2, ws)
    torch.cuda.synchronize()
    assert torch.equal(hits, hits2)
    # a prefix against the oracle, on the same index
    m = 200_000
    cpu = OracleIndex.from_bwt(g.export_bwt(), g.export_sa_samples(), 4, *g.export_borders(),
                               g.export_sentinel_indices(), a.io_to_dense_table, 6, 4, width=-32, n_threads=8)
    qbuf, qoff = q.host_slice(0, m)
    cs, ce = cpu.cursors_for_many(qbuf, qoff, n_threads=8)
    assert np.array_equal(out["start"][:m].cpu().numpy().astype(np.uint64), cs)
    assert np.array_equal(out["end"][:m].cpu().numpy().astype(np.uint64), ce)
    co, ct, cp = cpu.locate_intervals(cs, ce, n_threads=8)
    gh = hits[: int(co[-1])].cpu().numpy()
    assert np.array_equal(gh[:, 0].astype(np.uint64), ct) and np.array_equal(gh[:, 1].astype(np.uint64), cp)


@pytest.mark.parametrize("kind,bits", [("condensed", 64), ("condensed", 512), ("flat", 64), ("flat", 512)])
def test_import_of_all_four_table_variants(kind, bits):
    """gdx_index_from_parts_ex: an index handed over in any of the reference's four occurrence-table variants
    (FmIndexCondensed64/512, FmIndexFlat64/512, lib.rs:102-113) answers rank / symbol_at like the naive columns
    (tests/text_with_rank_support.rs:46-75) and searches / locates like the oracle."""
    from genedex_amd import FmIndex
    from oracle.oracle import OracleTable

    rng = np.random.default_rng(900 + bits + (kind == "flat"))
    a = alph.ascii_dna_with_n()
    texts = random_texts(rng, len_max=40000, symbols=b"ACGTN")
    c = cpu_index(texts, a, sa_rate=3, depth=2)
    table = OracleTable(c.bwt, c.sigma, kind, bits)
    g = FmIndex.from_parts(c.count_array, table.blocks, c.n, c.sa_samples, 3, c.border_keys, c.border_vals,
                           c.sentinel_indices, a, lookup_depth=2, table_kind=kind, block_bits=bits)
    cols = naive_occurrence_columns(c.bwt, c.sigma)
    idx = np.arange(c.n + 1, dtype=np.uint64)
    for s in range(c.sigma):
        assert np.array_equal(g.rank_many(np.full(c.n + 1, s, dtype=np.uint8), idx), cols[s])
    assert np.array_equal(g.symbol_at_many(np.arange(c.n, dtype=np.uint64)), c.bwt)
    qs = mixed_queries(rng, texts, 400, 200, 60)
    off, t, p, _ = g.locate_raw(*pack_queries(qs))
    co, ct, cp = c.locate_many(qs)
    assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist()
    blocks, bo, sbo = g.export_condensed_table()  # always exported as Condensed / Block64
    assert np.array_equal(blocks, c.blocks) and np.array_equal(bo, c.block_offsets)
    assert np.array_equal(sbo, c.superblock_offsets)


def test_import_of_a_wide_alphabet_flat512_table():
    from genedex_amd import FmIndex
    from oracle.oracle import OracleTable

    rng = np.random.default_rng(931)
    a = alph.u8_until(26)
    texts = random_texts(rng, len_max=3000, symbols=bytes(range(27)))
    c = cpu_index(texts, a, sa_rate=2, depth=1)
    table = OracleTable(c.bwt, c.sigma, "flat", 512)
    g = FmIndex.from_parts(c.count_array, table.blocks, c.n, c.sa_samples, 2, c.border_keys, c.border_vals,
                           c.sentinel_indices, a, lookup_depth=1, table_kind="flat", block_bits=512)
    cols = naive_occurrence_columns(c.bwt, c.sigma)
    idx = np.arange(c.n + 1, dtype=np.uint64)
    for s in range(c.sigma):
        assert np.array_equal(g.rank_many(np.full(c.n + 1, s, dtype=np.uint8), idx), cols[s])
    qs = [bytes(q) for q in mixed_queries(rng, texts, 200, 0, 12)]
    off, t, p, _ = g.locate_raw(*pack_queries(qs))
    co, ct, cp = c.locate_many(qs)
    assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist()


def test_randomised_configurations_sample():
    """A short run of tests/parity_sweep.py (random alphabets, text shapes, sampling rates, table sizes, hinted and
    un-hinted device path against the oracle); the committed profiles hold the long runs."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parity_sweep.py")
    spec = importlib.util.spec_from_file_location("parity_sweep", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(2024)
    stats = {"queries": 0, "hits": 0, "status_nonzero": 0, "hinted_queries": 0}
    for _ in range(25):
        mod.one_round(rng, stats)
    assert stats["queries"] > 10_000 and stats["hits"] > 0


def _device_queries(qs):
    from genedex_amd.device import DeviceQueries

    qbuf, qoff = pack_queries(qs)
    return DeviceQueries.from_host(qbuf, qoff), qbuf, qoff


@pytest.mark.parametrize("seed", range(6))
def test_fused_record_path_equals_oracle(seed, search_variant):
    """gdx_locate_many_{search,offsets,hits}_dev: the search may finish a query from the jump entry it holds (lazy
    tail) and reports counts + locate hints in 16-byte records; counts, hit offsets and hits (order included) are the
    oracle's for every query length modulo 8, every sampling rate and every index configuration."""
    import torch

    from genedex_amd.device import DeviceEngine

    rng = np.random.default_rng(7000 + seed)
    a = alph.ascii_dna_with_n()
    texts = random_texts(rng, len_max=60000, symbols=b"ACGTN" if seed % 3 == 0 else b"ACGT")
    rate = [1, 2, 3, 4, 7, 16][seed]
    g, c = both(texts, a, sa_rate=rate, depth=[0, 2][seed % 2])
    qs = mixed_queries(rng, texts, 1500, 300, 75) + [b"", b"A", b"ACGTACGTAC"]
    dq, qbuf, qoff = _device_queries(qs)
    eng = DeviceEngine(g)
    rec = eng.alloc_records(dq.nq)
    off = torch.empty(dq.nq + 1, dtype=torch.int64, device="cuda")
    eng.locate_search(dq, rec)
    eng.locate_offsets(rec, dq.nq, off)
    torch.cuda.synchronize()
    total = int(off[dq.nq].item())
    hits = torch.empty((max(total, 1), 2), dtype=torch.int32, device="cuda")
    ws = torch.empty(max(eng.locate_workspace_bytes(total), 16), dtype=torch.uint8, device="cuda")
    eng.locate_hits(rec, dq.nq, off, total, hits, ws)
    counts = torch.empty(dq.nq, dtype=torch.int32, device="cuda")
    status = torch.empty(dq.nq, dtype=torch.uint8, device="cuda")
    eng.unpack_records(rec, dq.nq, counts, status)
    torch.cuda.synchronize()
    co, ct, cp = c.locate_many(qs)
    assert off.cpu().numpy().astype(np.uint64).tolist() == co.tolist()
    h = hits[:total].cpu().numpy().astype(np.uint32)
    assert h[:, 0].tolist() == ct.tolist() and h[:, 1].tolist() == cp.tolist()
    assert counts.cpu().numpy().astype(np.uint32).tolist() == np.diff(co).tolist()
    assert not status.any().item()
    # gdx_count_many_dev takes the same shortcut
    counts2 = torch.empty(dq.nq, dtype=torch.int32, device="cuda")
    eng.count(dq, counts2, status)
    torch.cuda.synchronize()
    assert torch.equal(counts, counts2)


@pytest.mark.parametrize("seed", range(4))
def test_fused_records_on_repeat_families(seed, search_variant):
    """Reads from repeat families end on several rows: the fast-path search takes intervals of up to sixteen rows
    through the jump table and decides the last few symbols of every row from the row's own entry (masked records);
    wider families and reads next to N or a text border go on in the general kernel from where the fast path
    stopped.  Counts, hit offsets and hits in the reference's order for family sizes 1..40, every read length
    16..90 and several sampling rates."""
    import torch

    from genedex_amd.device import DeviceEngine

    rng = np.random.default_rng(9100 + seed)
    a = alph.ascii_dna_with_n()
    texts = []
    for _ in range(6):
        parts = []
        for _ in range(int(rng.integers(3, 9))):
            unit = bytes(b"ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(40, 400))))
            for _ in range(int(rng.choice([1, 2, 3, 5, 8, 15, 17, 40]))):
                copy = bytearray(unit)
                if rng.random() < 0.7:  # a diverged copy
                    copy[int(rng.integers(0, len(copy)))] = b"ACGT"[int(rng.integers(0, 4))]
                parts.append(bytes(copy))
                if rng.random() < 0.3:
                    parts.append(b"N" * int(rng.integers(1, 4)) if seed % 2 else b"")
        order = rng.permutation(len(parts))
        texts.append(b"".join(parts[i] for i in order))
    g, c = both(texts, a, sa_rate=[4, 1, 3, 16][seed])
    if "search_fast" not in _QUERY_OPTIONS:  # (off by default on a text this repetitive: wide_permille > 500)
        g.set_query_options(**{**_QUERY_OPTIONS, "search_fast": 2})
    qs = []
    for _ in range(4000):
        t = texts[int(rng.integers(0, len(texts)))]
        ln = int(rng.integers(16, 91))
        pos = int(rng.integers(0, max(1, len(t) - ln)))
        q = t[pos:pos + ln]
        if b"N" not in q:
            qs.append(q)
    dq, qbuf, qoff = _device_queries(qs)
    eng = DeviceEngine(g)
    rec = eng.alloc_records(dq.nq)
    off = torch.empty(dq.nq + 1, dtype=torch.int64, device="cuda")
    eng.locate_search(dq, rec)
    eng.locate_offsets(rec, dq.nq, off)
    torch.cuda.synchronize()
    total = int(off[dq.nq].item())
    hits = torch.empty((max(total, 1), 2), dtype=torch.int32, device="cuda")
    ws = torch.empty(max(eng.locate_workspace_bytes(total), 16), dtype=torch.uint8, device="cuda")
    eng.locate_hits(rec, dq.nq, off, total, hits, ws)
    torch.cuda.synchronize()
    co, ct, cp = c.locate_many(qs)
    assert off.cpu().numpy().astype(np.uint64).tolist() == co.tolist()
    h = hits[:total].cpu().numpy().astype(np.uint32)
    assert h[:, 0].tolist() == ct.tolist() and h[:, 1].tolist() == cp.tolist()
    assert int((np.diff(co) > 1).sum()) > len(qs) // 4  # many multi-hit reads
    counts, st = g.count_raw(qbuf, qoff)
    assert counts.tolist() == np.diff(co).tolist() and not st.any()


@pytest.mark.parametrize("fast", [None, 0, 1, 2])
def test_genome_like_text_against_oracle(fast):
    """4 Mi symbols with the repeat structure of bench.py's genome-like text (segmental duplications with 0.5 %
    divergence, tandem repeats, poly-A, N gaps), 60 k reads of length 50 and of lengths 20..150: counts, hit offsets
    and hits of the record path equal the oracle's whichever way the search kernels split the work (general kernel
    only, fast path with 4-row or 16-row jumps, the index's own choice)."""
    import torch

    from genedex_amd import synth
    from genedex_amd.device import DeviceEngine, genome_like_text

    a = alph.ascii_dna_with_n()
    total = 1 << 22
    buf = genome_like_text(total, torch.device("cuda"), seed=11).cpu().numpy().tobytes()
    texts, at = [], 0
    for ln in synth.split_lengths(total, 3):
        texts.append(buf[at:at + ln])
        at += ln
    g, c = both(texts, a, sa_rate=4)
    if fast is not None:
        g.set_query_options(search_fast=fast)
    eng = DeviceEngine(g)
    masked = 0
    for len_min, len_max, seed in ((50, 50, 5), (20, 150, 6)):
        qbuf, qoff = synth.host_queries(texts, nq=30_000, len_min=len_min, len_max=len_max, sampled_fraction=0.9, seed=seed)
        from genedex_amd.device import DeviceQueries

        dq = DeviceQueries.from_host(qbuf, qoff)
        rec = eng.alloc_records(dq.nq)
        off = torch.empty(dq.nq + 1, dtype=torch.int64, device="cuda")
        eng.locate_search(dq, rec)
        eng.locate_offsets(rec, dq.nq, off)
        torch.cuda.synchronize()
        tot = int(off[dq.nq].item())
        hits = torch.empty((max(tot, 1), 2), dtype=torch.int32, device="cuda")
        ws = torch.empty(max(eng.locate_workspace_bytes(tot), 16), dtype=torch.uint8, device="cuda")
        eng.locate_hits(rec, dq.nq, off, tot, hits, ws)
        torch.cuda.synchronize()
        cs, ce = c.cursors_for_many(qbuf, qoff, n_threads=4)
        co, ct, cp = c.locate_intervals(cs, ce, n_threads=4)
        assert off.cpu().numpy().astype(np.uint64).tolist() == co.tolist()
        h = hits[:tot].cpu().numpy().astype(np.uint32)
        assert np.array_equal(h[:, 0], ct.astype(np.uint32)) and np.array_equal(h[:, 1], cp.astype(np.uint32))
        masked += int((((rec[:dq.nq, 3] >> 23) & 1) == 1).sum().item())
        # (... or, two rows with SA[row] in their jump entries, as a resolved record of two: kernels.hpp)
        masked += int(((((rec[:dq.nq, 3] >> 22) & 1) == 1) & ((rec[:dq.nq, 1] - rec[:dq.nq, 0]) == 2)).sum().item())
        assert int((np.diff(co.astype(np.int64)) > 1).sum()) > 1000  # reads from repeats
    if fast in (1, 2):
        assert masked > 1000  # reads that end on several rows were finished by the fast path


_TRANSLATIONS = {
    # how the fast-path kernel turns query bytes into 2-bit codes (IndexView::perm_*): name -> (alphabet, symbols of
    # the texts, symbols of random queries -- some of them outside the alphabet or not searchable)
    "case_insensitive": (alph.ascii_dna_with_n, b"ACGTacgtNn", b"ACGTacgtN!"),       # v_perm tables, mask 0xdf
    "case_sensitive": (lambda: alph.Alphabet.from_io_symbols(b"ACGTN", 1), b"ACGTN", b"ACGTacgtN"),  # mask 0xff
    # A, I, Q, Y share their low three bits: no v_perm table, the kernel translates through the table in LDS
    "clashing_low_bits": (lambda: alph.Alphabet.from_io_symbols(b"AIQYN", 1), b"AIQYN", b"AIQYNC"),
    "bytes_0_to_4": (lambda: alph.u8_until(4), bytes(range(5)), bytes(range(7))),
}


@pytest.mark.parametrize("name", list(_TRANSLATIONS))
def test_fast_path_query_translation(name):
    """The fast-path search kernel translates query bytes with v_perm_b32 tables when the alphabet allows it and
    through the 256-byte table otherwise; counts, statuses and hits are the oracle's either way, also for queries
    with symbols that are not searchable or not in the alphabet (alphabet.rs:195-198: status, no result)."""
    make, text_symbols, query_symbols = _TRANSLATIONS[name]
    a = make()
    rng = np.random.default_rng(sum(name.encode()))
    texts = [bytes(text_symbols[i] for i in rng.integers(0, len(text_symbols), int(rng.integers(1000, 30000))))
             for _ in range(4)]
    g, c = both(texts, a, sa_rate=3, jump_entry_bytes=32)
    g.set_query_options(search_fast=1 + sum(name.encode()) % 2)
    assert g.aux()["jump_entry_bytes"] == 32 and g.aux()["top_table_depth"] >= 1
    qs = []
    for _ in range(3000):
        t = texts[int(rng.integers(0, len(texts)))]
        pos = int(rng.integers(0, len(t)))
        qs.append(t[pos:pos + int(rng.integers(0, 90))])
    for _ in range(600):
        qs.append(bytes(query_symbols[i] for i in rng.integers(0, len(query_symbols), int(rng.integers(0, 40)))))
    qbuf, qoff = pack_queries(qs)
    cs, ce, cst = c.cursors_single(qbuf, qoff)
    ok = cst == 0
    assert ok.sum() > len(qs) // 2 and (~ok).sum() > 10
    counts, st = g.count_raw(qbuf, qoff, strict=False)
    assert st.tolist() == cst.tolist()
    assert counts[ok].tolist() == (ce - cs)[ok].tolist()
    off, t_, p_, st2 = g.locate_raw(qbuf, qoff, strict=False)
    assert st2.tolist() == cst.tolist()
    co, ct, cp = c.locate_intervals(np.where(ok, cs, 0), np.where(ok, ce, 0))
    assert off.tolist() == co.tolist() and t_.tolist() == ct.tolist() and p_.tolist() == cp.tolist()


@pytest.mark.parametrize("chunk", [1, 7, 8, 16, 32, 40, 1000])
def test_cursor_strings_in_chunks_equal_fused_search(chunk, search_variant):
    """gdx_cursor_extend_front_strings_dev: queries fed to cursor_empty cursors `chunk` symbols at a time from the
    right, with device-side active lists, end in exactly the intervals of cursors_for_many_queries (and hence of
    cursor.rs:34-51 applied symbol by symbol), frozen empty intervals included."""
    import torch

    from genedex_amd.device import DeviceEngine

    rng = np.random.default_rng(7100 + chunk)
    a = alph.ascii_dna_with_n()
    texts = random_texts(rng, len_max=50000, symbols=b"ACGTN" if chunk % 2 else b"ACGT")
    g, c = both(texts, a)
    qs = mixed_queries(rng, texts, 1200, 500, 160) + [b""]
    dq, qbuf, qoff = _device_queries(qs)
    cs, ce = c.cursors_for_many(qbuf, qoff)
    eng = DeviceEngine(g)
    m = dq.nq
    n = g.total_text_len()
    beg, end = dq.qoff[:-1].clone(), dq.qoff[1:].clone()
    cur_s = torch.zeros(m, dtype=torch.int32, device="cuda")
    cur_e = torch.full((m,), n, dtype=torch.int32, device="cuda")
    cur_st = torch.zeros(m, dtype=torch.uint8, device="cuda")
    act = [torch.arange(m, dtype=torch.int32, device="cuda"), torch.empty(m, dtype=torch.int32, device="cuda")]
    n_act = [torch.tensor([m], dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")]
    edge = end.clone()
    seen_counts = []
    while True:
        lo = torch.maximum(beg, edge - chunk)
        eng.cursor_extend_strings(cur_s, cur_e, dq.qbuf, lo, edge, m, cur_st, act[0], n_act[0], act[1], n_act[1])
        edge = lo
        act.reverse()
        n_act.reverse()
        live = int(n_act[0].item())
        seen_counts.append(live)
        # the active list holds exactly the cursors that are non-empty
        listed = sorted(act[0][:live].cpu().tolist())
        assert listed == torch.nonzero(cur_s != cur_e).flatten().cpu().tolist()
        if live == 0 or bool((edge <= beg).all().item()):
            break
    assert cur_s.cpu().numpy().astype(np.uint32).tolist() == cs.astype(np.uint32).tolist()
    assert cur_e.cpu().numpy().astype(np.uint32).tolist() == ce.astype(np.uint32).tolist()
    assert not cur_st.any().item()
    assert seen_counts == sorted(seen_counts, reverse=True)  # cursors only ever leave the list
    # gdx_cursor_extend_front_chunk_dev: the same feeding without edge arrays; its live list holds the cursors that
    # are non-empty AND have symbols left of the chunk just taken
    ck_s = torch.zeros(m, dtype=torch.int32, device="cuda")
    ck_e = torch.full((m,), n, dtype=torch.int32, device="cuda")
    ck_st = torch.zeros(m, dtype=torch.uint8, device="cuda")
    lens = (end - beg)
    a_in, na_in = None, None
    k = 0
    while True:
        eng.cursor_extend_chunk(ck_s, ck_e, dq.qbuf, dq.qoff, m, chunk, k, ck_st, a_in, na_in, act[1], n_act[1])
        live = int(n_act[1].item())
        want = torch.nonzero((ck_s != ck_e) & (lens > (k + 1) * chunk)).flatten().cpu().tolist()
        got = act[1][:live].cpu().tolist()
        assert sorted(got) == want
        if live == 0:
            break
        a_in, na_in = act[1].clone(), n_act[1].clone()
        k += 1
    assert torch.equal(ck_s, cur_s) and torch.equal(ck_e, cur_e) and not ck_st.any().item()
    # host form, whole strings at once, and the Python mirror of Cursor
    s0 = np.zeros(m, dtype=np.uint64)
    e0 = np.full(m, n, dtype=np.uint64)
    hs, he, hst = g.extend_front_strings_raw(s0, e0, qbuf, qoff)
    assert hs.tolist() == cs.tolist() and he.tolist() == ce.tolist() and not hst.any()
    cur = g.cursor_empty()
    cur.extend_query_front_by(qs[0])
    assert cur.interval() == (int(cs[0]), int(ce[0]))


def test_cursor_strings_stop_at_an_invalid_symbol():
    """alphabet.rs:195-198 panics on a symbol outside the alphabet; here the cursor stops where it stands, its
    status is set, it leaves the active list and ignores later calls."""
    a = alph.ascii_dna()
    texts = [b"ACGTACGTTTGACA"]
    g, c = both(texts, a)
    qs = [b"ACGT", b"AC!T", b"TTGA"]
    qbuf, qoff = pack_queries(qs)
    n = g.total_text_len()
    s, e, st = g.extend_front_strings_raw(np.zeros(3, np.uint64), np.full(3, n, np.uint64), qbuf, qoff, strict=False)
    want = g.cursors_for_many_queries([b"ACGT", b"T", b"TTGA"])  # query 1 stops after its last symbol
    assert [(int(a_), int(b_)) for a_, b_ in zip(s, e)] == [w.interval() for w in want]
    assert st.tolist() == [0, 1, 0]
    s2, e2, st2 = g.extend_front_strings_raw(s, e, *pack_queries([b"T", b"A", b""]), status=st, strict=False)
    assert (int(s2[1]), int(e2[1])) == (int(s[1]), int(e[1])) and st2.tolist() == [0, 1, 0]
    assert (int(s2[0]), int(e2[0])) == g.cursor_for_query(b"TACGT").interval()


@pytest.mark.parametrize("seed", range(4))
def test_packed_queries_equal_ascii_queries(seed, search_variant):
    """Packed queries (2 bits per symbol, include/gdx.h): host and device packing agree, the exception list names
    exactly the queries with a symbol outside A, C, G, T, and every other query gets the oracle's intervals, counts
    and hits from the packed host calls (through many small chunks) and the packed device calls."""
    import ctypes as C

    import torch

    from genedex_amd import _lib
    from genedex_amd.device import DeviceEngine, _ptr, _stream

    if search_variant.startswith("ref-"):
        pytest.skip("packed queries go with the rank-line layout (the reference's own tables are queried as ASCII)")
    lib = _lib.load()
    rng = np.random.default_rng(7300 + seed)
    a = alph.ascii_dna_with_n()
    texts = random_texts(rng, len_max=40000, symbols=b"ACGTN" if seed % 2 else b"ACGT")
    g, c = both(texts, a, sa_rate=[4, 1, 5, 16][seed], depth=[0, 3][seed % 2])
    qs = mixed_queries(rng, texts, 1200, 400, 90) + [b"", b"ACGT", b"NNNN", b"ACGNT" * 5, b"acgtACGT"]
    qbuf, qoff = pack_queries(qs)
    nq = qoff.size - 1
    n_sym = int(qoff[-1])
    packed = np.zeros(int(lib.gdx_packed_bytes(n_sym)), dtype=np.uint8)
    exc = np.zeros(nq, dtype=np.uint64)
    n_exc = C.c_uint64(0)
    _lib.check(lib.gdx_pack_queries(g._h, qbuf.ctypes.data_as(_lib.u8p), qoff.ctypes.data_as(_lib.u64p), nq,
                                    packed.ctypes.data_as(_lib.u8p), exc.ctypes.data_as(_lib.u64p), nq, C.byref(n_exc)))
    dense = a.io_to_dense_table
    want_exc = [i for i, q in enumerate(qs) if any(not 1 <= dense[b] <= 4 for b in q)]
    assert exc[: n_exc.value].tolist() == want_exc
    # too small an exception buffer: the needed size is reported
    if want_exc:
        rc = lib.gdx_pack_queries(g._h, qbuf.ctypes.data_as(_lib.u8p), qoff.ctypes.data_as(_lib.u64p), nq,
                                  packed.ctypes.data_as(_lib.u8p), exc.ctypes.data_as(_lib.u64p), 0, C.byref(n_exc))
        assert rc == _lib.GDX_ERR_CAPACITY and n_exc.value == len(want_exc)
    # device packing gives the same bytes and counts the bad symbols
    d_q = torch.from_numpy(np.concatenate([qbuf[:n_sym], np.zeros(8, np.uint8)])).cuda()
    d_packed = torch.zeros(packed.size, dtype=torch.uint8, device="cuda")
    d_bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    _lib.check(lib.gdx_pack_queries_dev(g._h, _ptr(d_q), n_sym, _ptr(d_packed), None, _ptr(d_bad), _stream()))
    torch.cuda.synchronize()
    assert d_packed.cpu().numpy()[: (n_sym + 3) // 4].tolist() == packed[: (n_sym + 3) // 4].tolist()
    assert int(d_bad.item()) == sum(1 for q in qs for b in q if not 1 <= dense[b] <= 4)
    ok = np.ones(nq, dtype=bool)
    ok[want_exc] = False
    cs, ce = c.cursors_for_many(*pack_queries([q if k else b"" for q, k in zip(qs, ok)]))  # exceptions -> empty query
    # host calls on the packed buffer, through many chunks
    lib.gdx_debug_set_host_chunking(97, 0)
    try:
        s = np.zeros(nq, dtype=np.uint64)
        e = np.zeros(nq, dtype=np.uint64)
        st = np.zeros(nq, dtype=np.uint8)
        _lib.check(lib.gdx_cursors_for_many_queries_packed(g._h, packed.ctypes.data_as(_lib.u8p), qoff.ctypes.data_as(_lib.u64p),
                                                           nq, s.ctypes.data_as(_lib.u64p), e.ctypes.data_as(_lib.u64p),
                                                           st.ctypes.data_as(_lib.u8p)))
        assert s[ok].tolist() == cs[ok].tolist() and e[ok].tolist() == ce[ok].tolist() and not st.any()
        cnt = np.zeros(nq, dtype=np.uint64)
        _lib.check(lib.gdx_count_many_packed(g._h, packed.ctypes.data_as(_lib.u8p), qoff.ctypes.data_as(_lib.u64p), nq,
                                             cnt.ctypes.data_as(_lib.u64p), None))
        assert cnt[ok].tolist() == (ce - cs)[ok].tolist()
    finally:
        lib.gdx_debug_set_host_chunking(0, 0)
    # device calls: intervals, and the fused count + locate over records
    eng = DeviceEngine(g)
    d_off = torch.from_numpy(qoff.astype(np.int64)).cuda()
    d_s = torch.empty(nq, dtype=torch.int32, device="cuda")
    d_e = torch.empty(nq, dtype=torch.int32, device="cuda")
    d_st = torch.empty(nq, dtype=torch.uint8, device="cuda")
    _lib.check(lib.gdx_cursors_for_many_queries_packed_dev(g._h, _ptr(d_packed), _ptr(d_off), nq, _ptr(d_s), _ptr(d_e),
                                                           _ptr(d_st), _stream()))
    torch.cuda.synchronize()
    assert d_s.cpu().numpy().astype(np.uint32)[ok].tolist() == cs.astype(np.uint32)[ok].tolist()
    assert d_e.cpu().numpy().astype(np.uint32)[ok].tolist() == ce.astype(np.uint32)[ok].tolist()
    rec = eng.alloc_records(nq)
    _lib.check(lib.gdx_locate_many_search_packed_dev(g._h, _ptr(d_packed), _ptr(d_off), nq, _ptr(rec), _stream()))
    rec[torch.from_numpy(~ok).cuda(), 1] = rec[torch.from_numpy(~ok).cuda(), 0]  # the exceptions count as absent here
    off = torch.empty(nq + 1, dtype=torch.int64, device="cuda")
    eng.locate_offsets(rec, nq, off)
    torch.cuda.synchronize()
    total = int(off[nq].item())
    hits = torch.empty((max(total, 1), 2), dtype=torch.int32, device="cuda")
    ws = torch.empty(max(eng.locate_workspace_bytes(total), 16), dtype=torch.uint8, device="cuda")
    eng.locate_hits(rec, nq, off, total, hits, ws)
    torch.cuda.synchronize()
    co, ct, cp = c.locate_intervals(np.where(ok, cs, 0), np.where(ok, ce, 0))
    assert off.cpu().numpy().astype(np.uint64).tolist() == co.tolist()
    h = hits[:total].cpu().numpy().astype(np.uint32)
    assert h[:, 0].tolist() == ct.tolist() and h[:, 1].tolist() == cp.tolist()


@pytest.mark.parametrize("seed", range(4))
def test_query_layouts_equal_plain_queries(seed, search_variant):
    """gdx_query_layout_t: a batch as IO symbols with offsets (the plain form), uniform (no offsets), packed (2 bits per
    symbol), packed + uniform -- intervals, counts and located hits of every form are the oracle's, on every kernel variant
    and index structure (the seed-table chain, the text-unit kernels and the rank-line kernel read all forms natively; the
    pair-line kernels get a uniform batch's offsets written into scratch).  Uniform batches of reads shorter than the seed,
    of exactly the seed's length, of the entry's reach (k + 32) and longer (text units) are included."""
    import ctypes as C

    import torch

    from genedex_amd import _lib
    from genedex_amd.device import DeviceEngine, DeviceQueries

    if search_variant.startswith("ref-"):
        pytest.skip("layouts go with the rank-line layout")
    rng = np.random.default_rng(8100 + seed)
    a = alph.ascii_dna_with_n()
    texts = random_texts(rng, len_max=30000, symbols=b"ACGTN" if seed % 2 else b"ACGT")
    g, c = both(texts, a, sa_rate=[4, 1, 3, 8][seed], depth=[0, 2][seed % 2])
    eng = DeviceEngine(g)
    k = int(g.seed_info()["k"])
    lengths = sorted({1, 7, 20, 50, 101, max(k - 1, 1), max(k, 1), k + 32, k + 33, k + 70})
    for length in lengths:
        # reads of this length: substrings (without N: expressible in 2 bits), random ones, and substrings with one change
        qs = []
        for _ in range(400):
            t = texts[int(rng.integers(0, len(texts)))]
            if len(t) >= length:
                pos = int(rng.integers(0, len(t) - length + 1))
                q = bytearray(t[pos:pos + length])
                if b"N" in q:
                    continue
                if rng.random() < 0.3:
                    q[int(rng.integers(0, length))] = b"ACGT"[int(rng.integers(0, 4))]
                qs.append(bytes(q))
        qs += [bytes(b"ACGT"[i] for i in rng.integers(0, 4, length)) for _ in range(100)]
        qbuf, qoff = pack_queries(qs)
        nq = len(qs)
        cs, ce = c.cursors_for_many(qbuf, qoff)
        co, ct, cp = c.locate_intervals(cs, ce)
        plain = DeviceQueries.from_host(qbuf, qoff)
        forms = {"uniform": plain.as_uniform(length), "packed": plain.as_packed(g),
                 "packed+uniform": plain.as_packed(g).as_uniform(length)}
        for name, dq in forms.items():
            what = f"{name}, length {length}"
            out = eng.alloc_outputs(nq)
            eng.search(dq, out)
            assert out["start"].cpu().numpy().astype(np.uint32).tolist() == cs.astype(np.uint32).tolist(), what
            assert out["end"].cpu().numpy().astype(np.uint32).tolist() == ce.astype(np.uint32).tolist(), what
            assert not out["status"].any().item(), what
            cnt = torch.empty(nq, dtype=torch.int32, device="cuda")
            st = torch.empty(nq, dtype=torch.uint8, device="cuda")
            eng.count(dq, cnt, st)
            assert cnt.cpu().numpy().astype(np.uint64).tolist() == (ce - cs).tolist() and not st.any().item(), what
            for compact in (True, False):
                rec = eng.alloc_records(nq)
                cw = eng.alloc_compact(nq) if compact else None
                eng.locate_search(dq, rec, compact=cw)
                off = torch.empty(nq + 1, dtype=torch.int64, device="cuda")
                eng.locate_offsets(rec, nq, off, compact=cw)
                torch.cuda.synchronize()
                total = int(off[nq].item())
                hits = torch.empty((max(total, 1), 2), dtype=torch.int32, device="cuda")
                ws = torch.empty(max(eng.locate_workspace_bytes(total), 16), dtype=torch.uint8, device="cuda")
                eng.locate_hits(rec, nq, off, total, hits, ws, compact=cw)
                torch.cuda.synchronize()
                assert off.cpu().numpy().astype(np.uint64).tolist() == co.tolist(), what
                h = hits[:total].cpu().numpy().astype(np.uint32)
                assert h[:, 0].tolist() == ct.tolist() and h[:, 1].tolist() == cp.tolist(), what
    # the host-pointer calls on the same forms, through many small chunks (gdx_*_layout)
    lib = _lib.load()
    length = 50
    qs = []
    for _ in range(3000):  # (texts over A C G T N have next to no N-free window of 50 symbols: then the random reads are the batch)
        t = texts[int(rng.integers(0, len(texts)))]
        if len(t) >= length:
            pos = int(rng.integers(0, len(t) - length + 1))
            if b"N" not in t[pos:pos + length]:
                qs.append(t[pos:pos + length])
    qs += [bytes(b"ACGT"[i] for i in rng.integers(0, 4, length)) for _ in range(500)]
    qbuf, qoff = pack_queries(qs)
    nq = len(qs)
    cs, ce = c.cursors_for_many(qbuf, qoff)
    co, ct, cp = c.locate_intervals(cs, ce)
    packed = np.zeros(int(lib.gdx_packed_bytes(int(qoff[-1]))), dtype=np.uint8)
    n_exc = C.c_uint64(0)
    _lib.check(lib.gdx_pack_queries(g._h, qbuf.ctypes.data_as(_lib.u8p), qoff.ctypes.data_as(_lib.u64p), nq,
                                    packed.ctypes.data_as(_lib.u8p), None, 0, C.byref(n_exc)))
    assert n_exc.value == 0
    lib.gdx_debug_set_host_chunking(203, 0)
    try:
        for buf, pk in ((qbuf, False), (packed, True)):
            for off_arr, ul in ((qoff, 0), (None, length)):
                what = f"host, packed={pk}, uniform_len={ul}"
                cnt, st = g.count_layout_raw(buf, off_arr, nq, packed=pk, uniform_len=ul)
                assert cnt.tolist() == (ce - cs).tolist() and not st.any(), what
                s_, e_, st = g.cursors_layout_raw(buf, off_arr, nq, packed=pk, uniform_len=ul)
                assert s_.tolist() == cs.tolist() and e_.tolist() == ce.tolist() and not st.any(), what
                off, t_, p_, st = g.locate_layout_raw(buf, off_arr, nq, packed=pk, uniform_len=ul)
                assert off.tolist() == co.tolist() and t_.tolist() == ct.tolist() and p_.tolist() == cp.tolist(), what
                # the narrow form of the same call: u32 offsets and 8-byte hits in pinned memory of the library's, every chunk one
                # fused step; from a pageable and from a pinned query buffer (no staging copy)
                off, t_, p_, st = g.locate_layout32_raw(buf, off_arr, nq, packed=pk, uniform_len=ul)
                assert off.dtype == np.uint32 and t_.dtype == np.uint32 and not st.any(), what
                assert off.tolist() == co.tolist() and t_.tolist() == ct.tolist() and p_.tolist() == cp.tolist(), what
                pinned = torch.from_numpy(buf.copy()).pin_memory()
                off, t_, p_, st = g.locate_layout32_raw(None, off_arr, nq, packed=pk, uniform_len=ul, qbuf_ptr=pinned.data_ptr())
                assert off.tolist() == co.tolist() and t_.tolist() == ct.tolist() and p_.tolist() == cp.tolist(), what + " (pinned)"
                cnt, st = g.count_layout_raw(pinned.numpy(), off_arr, nq, packed=pk, uniform_len=ul)
                assert cnt.tolist() == (ce - cs).tolist(), what + " (pinned)"
        # take(k) through the narrow call (max_hits_per_query), and the library's held-back arrays given up
        g.set_query_options(max_hits_per_query=1)
        off, t_, p_, st = g.locate_layout32_raw(qbuf, qoff, nq)
        want = np.minimum(np.diff(co), 1)
        assert np.diff(off.astype(np.int64)).tolist() == want.tolist()
        first = co[:-1][want > 0]
        assert t_.tolist() == ct[first].tolist() and p_.tolist() == cp[first].tolist()
        g.set_query_options(max_hits_per_query=0)
        # short reads with hundreds of hits each: a chunk's hits exceed what its device buffers were sized for (the second
        # half of the step runs again with room) and the pinned hit array grows while chunks are on their way
        short = [bytes(b"ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(3, 6)))) for _ in range(900)]
        sbuf, soff = pack_queries(short)
        ss_, se_ = c.cursors_for_many(sbuf, soff)
        so, st_, sp_ = c.locate_intervals(ss_, se_)
        if int(so[-1]) > 20_000:
            off, t_, p_, st = g.locate_layout32_raw(sbuf, soff, len(short))
            assert off.tolist() == so.tolist() and t_.tolist() == st_.tolist() and p_.tolist() == sp_.tolist()
        lib.gdx_release_cached_hits()
        off, t_, p_, st = g.locate_layout32_raw(qbuf, qoff, nq)
        assert off.tolist() == co.tolist() and t_.tolist() == ct.tolist() and p_.tolist() == cp.tolist()
        # a read with a symbol outside the alphabet among the others: its status byte comes back (the chunks without such a read
        # keep their status bytes on the device), it has no hits, everything else is as before
        bad_at = min(1500, len(qs) // 2)
        # (GDX_Q_INVALID_SYMBOL: "reached while the interval was still non-empty" -- a read of the text with its third last symbol
        # replaced: the search has matched two symbols when it meets it)
        bad = list(qs[:bad_at]) + [qs[0][:-3] + b"#" + qs[0][-2:]] + list(qs[bad_at:])
        bbuf, boff = pack_queries(bad)
        off, t_, p_, st = g.locate_layout32_raw(bbuf, boff, len(bad), strict=False)
        _, _, _, st_wide = g.locate_raw(bbuf, boff, strict=False)
        assert np.flatnonzero(st).tolist() == [bad_at] and st.tolist() == st_wide.tolist(), (np.flatnonzero(st), np.flatnonzero(st_wide))
        want_counts = np.insert(np.diff(co.astype(np.int64)), bad_at, 0)
        assert np.diff(off.astype(np.int64)).tolist() == want_counts.tolist()
        assert t_.tolist() == ct.tolist() and p_.tolist() == cp.tolist()
    finally:
        lib.gdx_debug_set_host_chunking(0, 0)
    # what a layout must refuse
    lay = _lib.QueryLayout()
    lib.gdx_query_layout_init(C.byref(lay))
    assert lay.struct_size == C.sizeof(_lib.QueryLayout) and lay.packed == 0 and lay.uniform_len == 0
    dq = DeviceQueries.from_host(*pack_queries([b"ACGT", b"ACGT"]))
    cnt = torch.empty(2, dtype=torch.int32, device="cuda")
    st = torch.empty(2, dtype=torch.uint8, device="cuda")
    from genedex_amd.device import _ptr, _stream
    assert lib.gdx_count_many_layout_dev(g._h, _ptr(dq.qbuf), None, 2, C.byref(lay), _ptr(cnt), _ptr(st), _stream()) == _lib.GDX_ERR_INVALID_ARGUMENT
    lay.packed = 2
    assert lib.gdx_count_many_layout_dev(g._h, _ptr(dq.qbuf), _ptr(dq.qoff), 2, C.byref(lay), _ptr(cnt), _ptr(st), _stream()) == _lib.GDX_ERR_INVALID_ARGUMENT
    lay.packed, lay.uniform_len = 0, 1 << 21
    assert lib.gdx_count_many_layout_dev(g._h, _ptr(dq.qbuf), None, 2, C.byref(lay), _ptr(cnt), _ptr(st), _stream()) == _lib.GDX_ERR_INVALID_ARGUMENT
    with pytest.raises(ValueError):
        DeviceQueries.from_host(*pack_queries([b"ACGT", b"ACG"])).as_uniform(4)
    with pytest.raises(ValueError):
        DeviceQueries.from_host(*pack_queries([b"ACGT", b"ACNT"])).as_packed(g)


def test_a_lookup_table_that_cannot_exist_is_refused():
    """k^depth table entries: 95 searchable symbols overflow 64 bits at depth 10, 20 at depth 15 -- an argument error
    (GDX_ERR_INVALID_ARGUMENT) at build time, never a wrapped size; a table that merely does not fit the device fails as an
    allocation (GDX_ERR_DEVICE) as before; the same alphabets at small depths still build."""
    from genedex_amd import GdxError, _lib

    for a, depths in ((alph.ascii_printable(), (7, 10, 24)), (alph.ascii_amino_acid(), (10, 15, 24))):
        texts = [bytes(a.dense_to_io_table[:12]) * 20]
        for depth in depths:
            with pytest.raises(GdxError) as e:
                gpu_index(texts, a, depth=depth)
            assert e.value.status == _lib.GDX_ERR_INVALID_ARGUMENT and "lookup" in str(e.value), (depth, str(e.value))
        g = gpu_index(texts, a, depth=2)
        assert g.count(texts[0][:12]) == 20


@pytest.mark.parametrize("depth", [16, 19])
def test_lookup_tables_deeper_than_fifteen(depth):
    """The reference indexes its lookup tables with const-curried code up to depth 15 and a dynamic loop beyond
    (lookup_table.rs:68-113); here every depth is the loop, with 64-bit table offsets.  A two-symbol alphabet keeps the
    deep tables small (2^depth entries); an unsearchable symbol inside the suffix is still reported."""
    rng = np.random.default_rng(1600 + depth)
    a = alph.Alphabet.from_io_symbols(b"ACN", 1)  # A, C searchable; N valid but not searchable
    texts = [bytes(b"ACN"[i] for i in rng.choice(3, int(rng.integers(3000, 9000)), p=[.49, .49, .02])) for _ in range(3)]
    g, c = both(texts, a, depth=depth)
    assert g.export_lookup_table(depth).tolist() == c.lookup_table(depth).tolist()
    qs = []
    for _ in range(1500):
        t = texts[int(rng.integers(0, 3))]
        pos = int(rng.integers(0, len(t)))
        qs.append(t[pos:pos + int(rng.integers(0, 60))])
        qs.append(bytes(b"AC"[i] for i in rng.integers(0, 2, int(rng.integers(0, 40)))))
    qbuf, qoff = pack_queries(qs)
    s, e, st = g.cursors_raw(qbuf, qoff, strict=False)
    cs, ce, cst = c.cursors_single(qbuf, qoff)
    assert st.tolist() == cst.tolist() and (cst == 2).sum() > 10
    ok = st == 0
    assert s[ok].tolist() == cs[ok].tolist() and e[ok].tolist() == ce[ok].tolist()
    off, t_, p_, _ = g.locate_raw(qbuf, qoff, strict=False)
    co, ct, cp = c.locate_intervals(np.where(ok, cs, 0), np.where(ok, ce, 0))
    assert off.tolist() == co.tolist() and t_.tolist() == ct.tolist() and p_.tolist() == cp.tolist()


@pytest.mark.parametrize("n_texts", [2, 200, 256, 257, 700])
def test_host_locate_on_collections_of_many_texts(n_texts):
    """The host-pointer locate calls send a chunk's results across PCIe as the found-bitmap wire -- text ids as BYTES beside
    positions in the text -- when the collection has at most 256 texts, and let the device write offsets and hits otherwise
    (host_api.hip): wide and narrow results, counts and statuses are the oracle's on either side of that border, through many
    small chunks, with reads that occur in several texts (every hit's text id matters) and reads that occur nowhere."""
    from genedex_amd import _lib

    rng = np.random.default_rng(7700 + n_texts)
    a = alph.ascii_dna()
    shared = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 40))  # a stretch that many texts hold
    texts = []
    for t in range(n_texts):
        body = bytes(b"ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(60, 400))))
        texts.append(body + shared if t % 5 == 0 else body)
    g, c = both(texts, a)
    qs = []
    for i in range(6000):
        t = texts[int(rng.integers(0, n_texts))]
        ln = int(rng.integers(12, 45))
        at = int(rng.integers(0, len(t) - ln + 1))
        qs.append(t[at:at + ln] if i % 7 else bytes(b"ACGT"[k] for k in rng.integers(0, 4, ln)))
    qs += [shared, shared[5:35], shared[:20]]
    qbuf, qoff = pack_queries(qs)
    co, ct, cp = c.locate_many(qs)
    assert int(np.diff(co).max()) > n_texts // 6  # (the shared stretch: a hit in every fifth text)
    lib = _lib.load()
    lib.gdx_debug_set_host_chunking(501, 0)
    try:
        off, t_, p_, st = g.locate_raw(qbuf, qoff)
        assert off.tolist() == co.tolist() and t_.tolist() == ct.tolist() and p_.tolist() == cp.tolist() and not st.any()
        off, t_, p_, st = g.locate_alloc_raw(qbuf, qoff)
        assert off.tolist() == co.tolist() and t_.tolist() == ct.tolist() and p_.tolist() == cp.tolist() and not st.any()
        off, t_, p_, st = g.locate_layout32_raw(qbuf, qoff, len(qs))
        assert off.tolist() == co.tolist() and t_.tolist() == ct.tolist() and p_.tolist() == cp.tolist() and not st.any()
        cnt, st = g.count_raw(qbuf, qoff)
        assert cnt.tolist() == np.diff(co).tolist() and not st.any()
    finally:
        lib.gdx_debug_set_host_chunking(0, 0)
