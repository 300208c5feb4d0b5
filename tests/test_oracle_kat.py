"""Pins the CPU oracle against the reference's own known-answer tests (tests/golden/genedex_kat.json)."""
import numpy as np
import pytest

from helpers import alphabet_by_name, as_bytes, naive_occurrence_columns
from oracle.oracle import OracleIndex, naive_suffix_array


def build(case, width=None, sa_rate=None):
    a = alphabet_by_name(case["alphabet"])
    texts = [as_bytes(t) for t in case["texts"]]
    w = width if width is not None else case.get("width", 32)
    return OracleIndex.build(texts, a.io_to_dense_table, a.num_dense_symbols(), a.num_searchable_dense_symbols(),
                             sa_rate=sa_rate or case["sa_rate"], lookup_depth=case["lookup_depth"], width=w)


def test_locate_kats(kat):
    for case in kat["locate"]:
        for w in case["widths"]:
            ix = build(case, width=w)
            q = as_bytes(case["query"])
            want = {tuple(h) for h in case["hits"]}
            assert set(ix.locate(q)) == want, case["source"]
            off, t, p = ix.locate_many([q])
            assert set(zip(t.tolist(), p.tolist())) == want, case["source"]
            assert ix.count(q) == len(want)


def test_count_kats(kat):
    for case in kat["count"]:
        ix = build(case)
        assert ix.count(as_bytes(case["query"])) == case["count"], case["source"]
        assert ix.count_many([as_bytes(case["query"])]).tolist() == [case["count"]]


def test_locate_many_with_unsearchable_symbol_runs(kat):
    for case in kat["locate_many_runs"]:
        ix = build(case)
        qs = [as_bytes(q) for q in case["queries"]]
        off, t, p = ix.locate_many(qs)
        assert off.size == len(qs) + 1
        # GT occurs twice, GTN once (text 1: acGtn)
        assert int(off[3] - off[2]) == 2 and int(off[4] - off[3]) == 1


def test_cursor_kat(kat):
    for case in kat["cursor"]:
        ix = build(case)
        s, e, st = ix.cursor_for_query(as_bytes(case["query"]))
        assert st == 0 and e - s == case["count"]
        s, e, st = ix.extend_front(s, e, as_bytes(case["extend_front"])[0])
        assert st == 0 and e - s == case["count_after"]


def test_rank_kat(kat):
    for case in kat["rank"]:
        ix = OracleIndex.table_only(np.array(case["dense_text"], dtype=np.uint8), case["sigma"])
        for idx, sym in case["symbol_at"]:
            assert ix.symbol_at(idx) == sym
        for sym, idx, want in case["rank"]:
            assert ix.rank(sym, idx) == want


def _dense(case):
    if "dense_text" in case:
        return np.array(case["dense_text"], dtype=np.uint8)
    r = case["dense_text_repeat"]
    return np.full(r["times"], r["symbol"], dtype=np.uint8)


def test_rank_against_naive_unit_cases(kat):
    for case in kat["rank_vs_naive"]:
        text = _dense(case)
        sigma = case["sigma"]
        ix = OracleIndex.table_only(text, sigma)
        assert ix.n == text.size
        cols = naive_occurrence_columns(text, sigma)
        step = 1 if text.size < 5000 else 97
        idxs = sorted(set(range(0, text.size + 1, step)) | {text.size, max(text.size - 1, 0)}
                      | {i for i in (63, 64, 65, 65535, 65536) if i <= text.size})
        for c in range(sigma):
            for i in idxs:
                assert ix.rank(c, i) == int(cols[c, i]), (case["source"], c, i)
        for i in range(0, text.size, step):
            assert ix.symbol_at(i) == int(text[i])
        with pytest.raises(AssertionError):
            ix.rank(sigma, 0)
        with pytest.raises(AssertionError):
            ix.rank(0, text.size + 1)
        with pytest.raises(AssertionError):
            ix.symbol_at(text.size)


def test_sigma_below_two_is_rejected():
    # proptest-regressions/text_with_rank_support/mod.txt:7 (text=[], alphabet_size=1); condensed.rs:64 assert
    with pytest.raises(ValueError):
        OracleIndex.table_only(np.zeros(0, dtype=np.uint8), 1)


def test_concat_text_kat(kat):
    for case in kat["concat_text"]:
        a = alphabet_by_name(case["alphabet"])
        ix = OracleIndex.build([as_bytes(t) for t in case["texts"]], a.io_to_dense_table, a.num_dense_symbols(),
                               a.num_searchable_dense_symbols())
        assert ix.dense_text.tolist() == case["dense_text"]
        assert ix.sentinel_indices.tolist() == case["sentinel_indices"]
        freq = np.diff(ix.count_array).tolist()
        assert freq == case["frequency"]
        assert ix.count_array[0] == 0


def test_text_id_tree_kat(kat):
    for case in kat["text_id_tree"]:
        # build an index whose sentinels sit exactly at the given positions
        sent = case["sentinel_indices"]
        lens = [sent[0]] + [sent[i] - sent[i - 1] - 1 for i in range(1, len(sent))]
        a = alphabet_by_name("ascii_dna")
        ix = OracleIndex.build([b"A" * ln for ln in lens], a.io_to_dense_table, 5, 4)
        assert ix.sentinel_indices.tolist() == sent
        for pos, want in case["lookups"]:
            assert ix.lookup_text_id(pos) == want
        # the tree is a lower_bound over the sentinel positions
        for pos in range(sent[-1] + 1):
            assert ix.lookup_text_id(pos) == int(np.searchsorted(np.array(sent), pos, side="left"))


def test_recover_range_equals_full_suffix_array(kat):
    for case in kat["recover_range_equals_full_sa"]:
        sampled = build(case)
        full = build(case, sa_rate=1)
        n = sampled.n
        got = sampled.recover_range(0, n)
        want = full.recover_range(0, n)
        assert got.tolist() == want.tolist(), case["source"]
        assert want.tolist() == full.full_sa.tolist()
        assert full.full_sa.tolist() == naive_suffix_array(full.dense_text).tolist()


def test_alphabet_sizes(kat):
    for case in kat["alphabet_sizes"]:
        a = alphabet_by_name(case["name"])
        assert a.num_dense_symbols() == case["dense"]
        assert a.num_searchable_dense_symbols() == case["searchable"]
    for max_symbol in range(1, 255):
        a = alphabet_by_name(f"u8_until({max_symbol})")
        assert a.num_dense_symbols() == max_symbol + 2
        assert a.num_searchable_dense_symbols() == max_symbol + 1
