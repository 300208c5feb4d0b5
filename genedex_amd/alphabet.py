"""Alphabets: IO symbol (usually ASCII) -> dense symbol tables.

Mirrors the reference's ``genedex::alphabet`` module (src/alphabet.rs:24-345): dense symbol 0
is the sentinel / text delimiter and has no IO representation; the last
``num_io_symbols_not_searchable`` dense symbols may appear in texts but not in lookup-table
suffixes of queries.  Only the 256-byte translation table travels to the device.
"""
from __future__ import annotations

import numpy as np


class Alphabet:
    """src/alphabet.rs:24-28."""

    def __init__(self, io_to_dense, dense_to_io, num_io_symbols_not_searchable):
        io_to_dense = np.ascontiguousarray(io_to_dense, dtype=np.uint8)
        if io_to_dense.shape != (256,):
            raise ValueError("io_to_dense table must have 256 entries")
        size = len(dense_to_io) + 1
        # src/alphabet.rs:151-192 (Alphabet::new asserts)
        if not 1 < size <= 256:
            raise ValueError("Alphabet size must be in 2..=256 (including sentinel)")
        if len(set(io_to_dense[io_to_dense != 0].tolist())) + 1 != size:
            raise ValueError("The alphabet translation tables are invalid.")
        if num_io_symbols_not_searchable + 2 > size:
            raise ValueError("Invalid alphabet. there must be at least one searchable symbol.")
        self.io_to_dense_table = io_to_dense
        self.dense_to_io_table = bytes(dense_to_io)
        self.num_io_symbols_not_searchable = int(num_io_symbols_not_searchable)

    @classmethod
    def from_io_symbols(cls, symbols, num_io_symbols_not_searchable=0):
        """src/alphabet.rs:43-75."""
        symbols = bytes(symbols)
        if len(set(symbols)) != len(symbols):
            raise ValueError("Symbols of the alphabet must be unique.")
        if len(symbols) > 255:
            raise ValueError("Alphabet size can be at most 255 (to leave space for the sentinel).")
        table = np.zeros(256, dtype=np.uint8)
        for i, s in enumerate(symbols):
            table[s] = i + 1
        return cls(table, symbols, num_io_symbols_not_searchable)

    @classmethod
    def from_ambiguous_io_symbols(cls, groups, num_io_symbols_not_searchable=0):
        """src/alphabet.rs:99-149: several IO symbols may map to one dense symbol."""
        groups = [bytes(g) for g in groups]
        if any(len(g) == 0 for g in groups):
            raise ValueError("Every group of symbols must contain at least one symbol")
        flat = b"".join(groups)
        if len(set(flat)) != len(flat):
            raise ValueError("Symbols of the alphabet must be unique.")
        if len(groups) > 255:
            raise ValueError("Alphabet size can be at most 255 (to leave space for the sentinel).")
        table = np.zeros(256, dtype=np.uint8)
        for i, g in enumerate(groups):
            for s in g:
                table[s] = i + 1
        return cls(table, bytes(g[0] for g in groups), num_io_symbols_not_searchable)

    def try_io_to_dense_representation(self, symbol):
        d = int(self.io_to_dense_table[symbol])
        return d if d != 0 else None

    def io_to_dense_representation(self, symbol):
        """src/alphabet.rs:195-198 (panics in the reference)."""
        d = self.try_io_to_dense_representation(symbol)
        if d is None:
            raise ValueError("symbol in io representation should be valid")
        return d

    def dense_to_io_representation(self, symbol):
        if symbol == 0 or symbol > len(self.dense_to_io_table):
            raise ValueError("symbol in dense representation should be valid")
        return self.dense_to_io_table[symbol - 1]

    def num_dense_symbols(self):
        return len(self.dense_to_io_table) + 1

    def num_searchable_dense_symbols(self):
        return self.num_dense_symbols() - self.num_io_symbols_not_searchable - 1

    def encode(self, text) -> np.ndarray:
        """Dense encoding of a text; raises on a symbol outside the alphabet."""
        a = np.frombuffer(bytes(text), dtype=np.uint8)
        d = self.io_to_dense_table[a]
        if d.size and int(d.min()) == 0:
            raise ValueError("symbol in io representation should be valid")
        return d


def _ci(letters):
    return [bytes([c, c + 32]) if 65 <= c <= 90 else bytes([c]) for c in letters]


def ascii_dna():
    """src/alphabet.rs:251-253."""
    return Alphabet.from_ambiguous_io_symbols([b"Aa", b"Cc", b"Gg", b"Tt"], 0)


def ascii_dna_with_n():
    """src/alphabet.rs:256-258."""
    return Alphabet.from_ambiguous_io_symbols([b"Aa", b"Cc", b"Gg", b"Tt", b"Nn"], 1)


def ascii_dna_iupac():
    """src/alphabet.rs:264-273."""
    return Alphabet.from_ambiguous_io_symbols(_ci(b"ACGTNRYKMSWBDHV"), 0)


def ascii_dna_iupac_as_dna_with_n():
    """src/alphabet.rs:277-288."""
    return Alphabet.from_ambiguous_io_symbols([b"Aa", b"Cc", b"Gg", b"Tt", b"NnRrYyKkMmSsWwBbDdHhVv"], 1)


def ascii_amino_acid():
    """src/alphabet.rs:291-299."""
    return Alphabet.from_ambiguous_io_symbols(_ci(b"ACDEFGHIKLMNOPQRSTUVWY"), 0)


def ascii_amino_acid_iupac():
    """src/alphabet.rs:303-336."""
    return Alphabet.from_ambiguous_io_symbols(_ci(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ") + [b"*"], 0)


def u8_until(max_symbol):
    """src/alphabet.rs:339-341."""
    return Alphabet.from_io_symbols(bytes(range(max_symbol + 1)), 0)


def ascii_printable():
    """src/alphabet.rs:344-346."""
    return Alphabet.from_io_symbols(bytes(range(32, 127)), 0)
