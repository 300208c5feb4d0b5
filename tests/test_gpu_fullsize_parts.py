"""A collection beyond 32-bit index storage: 2^32 + 2^20 symbols in 48 texts (two hg38-scale assemblies' worth) through
the partitioned index (gdx_parts_*), which stands in for the reference's `IndexStorage = i64`
(construction/mod.rs:225-252): build, count and locate, every checked hit spelled against the text."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_collection_beyond_two_to_the_32():
    import torch

    import bench
    from genedex_amd import alphabet
    from genedex_amd.device import DeviceQueries, build_parts_from_device_text, hg38_text_lengths, synth_text
    from genedex_amd.index import build_options

    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    n_texts = 48
    total = (1 << 32) + (1 << 20) - n_texts  # symbols without the sentinels: n = 2^32 + 2^20 with them
    a = alphabet.ascii_dna_with_n()
    io_text = synth_text(total, seed=77, n_per_million=10_000, device=dev)
    lengths = hg38_text_lengths(total // 2, n_texts // 2)
    lengths = lengths + hg38_text_lengths(total - sum(lengths), n_texts // 2)
    assert sum(lengths) == total and len(lengths) == n_texts
    # (top table + text units + full suffix array per part: ~45 GB each, no 100 GB jump tables for a test)
    opts = build_options(pair_lines=False, jump_entry_bytes=0, top_table_depth=16, full_suffix_array=True, text_units=True)
    g = build_parts_from_device_text(io_text, lengths, a, options=opts)
    assert g.num_parts == 2 and g.num_texts() == n_texts and g.total_text_len() == (1 << 32) + (1 << 20)
    nq = 10_000_000
    q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=78)
    qbuf, qoff = q.host_slice(0, nq)
    counts, st = g.count_raw(qbuf, qoff)
    assert not st.any()
    found = int((counts > 0).sum())
    assert 0.899 * nq < found < 0.9005 * nq  # the 90 % sampled reads are all found, a few random ones too
    off, t, p, st = g.locate_raw(qbuf, qoff)
    assert not st.any() and int(off[-1]) == int(counts.sum()) == t.size
    assert np.array_equal(np.diff(off.astype(np.int64)), counts.astype(np.int64))
    assert int(t.max()) == n_texts - 1 and int((t >= n_texts // 2).sum()) > 0.3 * t.size  # both parts answer
    hits = torch.from_numpy(np.stack([t.astype(np.int64), p.astype(np.int64)], axis=1)).to(dev).to(torch.int32)
    chk = bench.verify_hits(torch, io_text, lengths, q, {"hit_offsets": torch.from_numpy(off.astype(np.int64)).to(dev)},
                            hits, int(off[-1]), nq, 2_000_000)
    assert chk["hits_checked"] == chk["hits_matching_text"] == 2_000_000  # every checked hit spells its read


def test_single_index_beyond_two_to_the_32():
    """n = 2^32 + 2^20 in ONE index with 64-bit rows (wide.hip, index_width 64): build, count, intervals and locate on one
    MI355X; every checked hit spelled against the text; the same reads through the partitioned index of the test above
    would give the same counts (checked here against direct occurrence counts of a few reads in the text)."""
    import torch

    import bench
    from genedex_amd import alphabet
    from genedex_amd.device import DeviceQueries, build_index_from_device_text, hg38_text_lengths, synth_text

    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    n_texts = 3
    total = (1 << 32) + (1 << 20) - n_texts
    a = alphabet.ascii_dna_with_n()
    io_text = synth_text(total, seed=79, n_per_million=10_000, device=dev)
    lengths = [total - (1 << 21), (1 << 21) - 5, 5]  # one text alone is longer than 2^32 - 1 symbols
    g = build_index_from_device_text(io_text, lengths, a, index_storage="i64")
    assert g.info.index_width == 64 and g.total_text_len() == (1 << 32) + (1 << 20) and g.num_texts() == n_texts
    nq = 2_000_000
    q = DeviceQueries.synth(io_text, lengths, nq, 50, 50, 900_000, seed=80)
    qbuf, qoff = q.host_slice(0, nq)
    counts, st = g.count_raw(qbuf, qoff)
    assert not st.any()
    found = int((counts > 0).sum())
    assert 0.899 * nq < found < 0.9005 * nq
    s, e, st = g.cursors_raw(qbuf, qoff)
    assert np.array_equal(e - s, counts)
    # rows beyond 32 bits are in use: N is the largest symbol, so the suffixes that start with N are the last rows
    ns, ne, _ = g.cursors_raw(*__import__("genedex_amd").pack_queries([b"N", b"NA"]))
    assert int(ne[0]) == (1 << 32) + (1 << 20) and int(ns[0]) < (1 << 32) and int(ns[1]) > (1 << 32) - (1 << 26)
    n_count = int((io_text == ord("N")).sum().item())
    assert int(ne[0] - ns[0]) == n_count
    off, t, p, st = g.locate_alloc_raw(qbuf, qoff)
    assert not st.any() and int(off[-1]) == int(counts.sum()) == t.size
    assert int(p.max()) > (1 << 32) - (1 << 22)  # positions near the end of the 4 G text
    hits = torch.from_numpy(np.stack([t.astype(np.int64), p.astype(np.int64)], axis=1)).to(dev)
    # bench.verify_hits works on int32 hit tensors of the 32-bit engine: positions here need 64 bits
    g_ = torch.Generator(device=dev)
    g_.manual_seed(5)
    h = torch.randint(0, int(off[-1]), (500_000,), device=dev, generator=g_)
    off_t = torch.from_numpy(off.astype(np.int64)).to(dev)
    qi = torch.searchsorted(off_t, h, right=True) - 1
    toff = torch.zeros(n_texts + 1, dtype=torch.int64, device=dev)
    toff[1:] = torch.cumsum(torch.tensor(lengths, dtype=torch.int64, device=dev), 0)
    basepos = toff[hits[h, 0]] + hits[h, 1]
    j = torch.arange(50, device=dev)
    same = io_text[(basepos[:, None] + j[None, :]).clamp_(max=total - 1)] == q.qbuf[(q.qoff[qi][:, None] + j[None, :])]
    assert bool(same.all().item())  # every checked hit spells its read
