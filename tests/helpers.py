"""Shared test helpers: naive executable definitions (the reference's own test oracles,
tests/fmindex.rs:207-227 and tests/text_with_rank_support.rs:8-44) and fixture decoding."""
from __future__ import annotations

import numpy as np

from genedex_amd import alphabet as alph


def alphabet_by_name(name: str):
    if name.startswith("u8_until("):
        return alph.u8_until(int(name[len("u8_until("):-1]))
    return getattr(alph, name)()


def as_bytes(x) -> bytes:
    return x.encode() if isinstance(x, str) else bytes(x)


def naive_search(texts, query, fold=None):
    """tests/fmindex.rs:207-227.  `fold` maps IO symbols to dense symbols for case-insensitive
    alphabets (the reference test only uses upper-case ACGT, where it is the identity)."""
    hits = set()
    q = bytes(query)
    if fold is not None:
        q = bytes(fold[b] for b in q)
    for text_id, text in enumerate(texts):
        t = bytes(text)
        if fold is not None:
            t = bytes(fold[b] for b in t)
        if len(q) == 0:
            for position in range(len(t) + 1):
                hits.add((text_id, position))
            continue
        start = 0
        while True:
            p = t.find(q, start)
            if p < 0:
                break
            hits.add((text_id, p))
            start = p + 1
    return hits


def naive_occurrence_columns(dense_text: np.ndarray, sigma: int) -> np.ndarray:
    """tests/text_with_rank_support.rs:8-44: column[c][i] = #c in text[0..i)."""
    t = np.asarray(dense_text, dtype=np.uint8)
    cols = np.zeros((sigma, t.size + 1), dtype=np.uint64)
    for c in range(sigma):
        cols[c, 1:] = np.cumsum(t == c)
    return cols


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def random_texts(rng, n_texts_max=4, len_max=1500, symbols=b"ACGT"):
    n_texts = int(rng.integers(1, n_texts_max + 1))
    out = []
    for _ in range(n_texts):
        ln = int(rng.integers(0, len_max))
        out.append(bytes(symbols[i] for i in rng.integers(0, len(symbols), ln)))
    return out
