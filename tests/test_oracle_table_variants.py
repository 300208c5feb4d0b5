"""The reference's four occurrence-table variants (Condensed/Flat x Block64/Block512) restated in the oracle,
against the naive rank columns -- mirrors tests/text_with_rank_support.rs:46-135."""
import numpy as np
import pytest

from helpers import naive_occurrence_columns
from oracle.oracle import OracleIndex, OracleTable

VARIANTS = [("condensed", 64), ("condensed", 512), ("flat", 64), ("flat", 512)]


def check(text, sigma, step=1):
    cols = naive_occurrence_columns(text, sigma)
    for kind, bits in VARIANTS:
        t = OracleTable(text, sigma, kind, bits)
        idxs = sorted(set(range(0, text.size + 1, step)) | {text.size}
                      | {i for i in (47, 48, 49, 495, 496, 497, 511, 512, 513, 65471, 65472, 65519, 65520, 65535, 65536)
                         if i <= text.size})
        for c in range(sigma):
            for i in idxs:
                assert t.rank(c, i) == int(cols[c, i]), (kind, bits, c, i)
        for i in range(0, text.size, step):
            assert t.symbol_at(i) == int(text[i]), (kind, bits, i)
        with pytest.raises(AssertionError):
            t.rank(sigma, 0)
        with pytest.raises(AssertionError):
            t.rank(0, text.size + 1)


def test_unit_cases(kat):
    for case in kat["rank_vs_naive"]:
        if "dense_text" in case:
            text = np.array(case["dense_text"], dtype=np.uint8)
        else:
            text = np.full(case["dense_text_repeat"]["times"], case["dense_text_repeat"]["symbol"], dtype=np.uint8)
        check(text, case["sigma"], step=1 if text.size < 3000 else 61)


@pytest.mark.parametrize("seed", range(8))
def test_random_texts(seed):
    rng = np.random.default_rng(6000 + seed)
    sigma = int(rng.integers(2, 257))
    n = int(rng.integers(0, 1000))
    check(rng.integers(0, sigma, n).astype(np.uint8), sigma)


def test_block_boundaries_of_every_variant():
    rng = np.random.default_rng(5)
    for n in (48, 96, 496, 512, 992, 1024, 65472, 65520, 65536, 65537, 70000):
        check(rng.integers(0, 6, n).astype(np.uint8), 6, step=97)


def test_condensed64_variant_equals_the_index_table():
    rng = np.random.default_rng(6)
    text = rng.integers(0, 6, 70000).astype(np.uint8)
    a = OracleIndex.table_only(text, 6)
    b = OracleTable(text, 6, "condensed", 64)
    assert np.array_equal(a.blocks, b.blocks) and np.array_equal(a.block_offsets, b.block_offsets)
    assert np.array_equal(a.superblock_offsets, b.superblock_offsets)
