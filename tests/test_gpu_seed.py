"""Seed table (gdx_build_options_t.seed_symbols): a bucketed hash table over the distinct k-mers of the text in front of
the count / locate searches.  Whatever it holds, counts, hits and hit order are the reference's (the oracle's), and the
same index answers identically with the table switched off."""
import numpy as np
import pytest

from genedex_amd import alphabet as alph

from helpers import naive_search, random_texts
from test_gpu_parity import cpu_index, gpu_index, mixed_queries, pack_queries

pytestmark = pytest.mark.gpu

LEAN = dict(pair_lines=False, jump_entry_bytes=0, top_table_depth=0)


def repetitive_texts(rng, n_texts=3, unit_len=300, copies=5, tail=800, symbols=b"ACGT"):
    """texts that share segments (k-mers on several rows) next to unique stretches"""
    unit = bytes(symbols[i] for i in rng.integers(0, len(symbols), unit_len))
    out = []
    for _ in range(n_texts):
        parts = []
        for _ in range(copies):
            parts.append(unit[: int(rng.integers(40, unit_len))])
            parts.append(bytes(symbols[i] for i in rng.integers(0, len(symbols), int(rng.integers(0, tail)))))
        out.append(b"".join(parts))
    return out


def check_against_oracle(g, c, qs, texts=None, fold=None):
    qbuf, qoff = pack_queries(qs)
    off, t, p, st = g.locate_raw(qbuf, qoff)
    co, ct, cp = c.locate_many(qs)
    assert not st.any()
    assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist()
    counts, st = g.count_raw(qbuf, qoff)
    assert not st.any()
    assert counts.tolist() == np.diff(co).tolist()
    if texts is not None:
        for k, q in list(enumerate(qs))[:150]:
            got = set(zip(t[off[k]:off[k + 1]].tolist(), p[off[k]:off[k + 1]].tolist()))
            assert got == naive_search(texts, q, fold=fold), q
    return off, t, p


@pytest.mark.parametrize("seed", range(12))
def test_seed_index_equals_oracle(seed):
    rng = np.random.default_rng(9100 + seed)
    a = [alph.ascii_dna(), alph.ascii_dna_with_n(), alph.ascii_dna_iupac_as_dna_with_n()][seed % 3]
    symbols = b"ACGTN" if seed % 3 else b"ACGT"
    texts = (repetitive_texts(rng, symbols=symbols) if seed % 2 else random_texts(rng, len_max=6000, symbols=symbols))
    k = [8, 9, 12, 13, 16, True][seed % 6]
    load = [None, 100, 40, 85][seed % 4]
    extra = [LEAN, dict(LEAN, full_suffix_array=True), {}, dict(jump_entry_bytes=16, top_table_depth=5)][(seed // 2) % 4]
    depth = [0, 3, 0, 6][seed % 4]
    rate = int(rng.integers(1, 20))
    g = gpu_index(texts, a, sa_rate=rate, depth=depth, seed_symbols=k, seed_load_percent=load, **extra)
    c = cpu_index(texts, a, sa_rate=rate, depth=depth)
    info = g.seed_info()
    assert info["k"] == (k if k is not True else info["k"]) and info["k"] >= 8
    assert info["single_entries"] + info["interval_entries"] > 0 or sum(len(t) for t in texts) < 8
    qs = mixed_queries(rng, texts, 500, 150, 140) + [b"", b"A", b"ACGTACGT", b"ACGTACGTA"]
    # reads that agree with the text on the seed and disagree further front (within and beyond the entry's 32 symbols)
    for q in list(qs[:200]):
        if len(q) > 12:
            at = int(rng.integers(0, len(q) - 9))
            qs.append(q[:at] + bytes([b"ACGT"[(b"ACGT".find(q[at:at + 1]) + 1) % 4]]) + q[at + 1:])
    qs = [q for q in qs if b"N" not in q or depth == 0]
    want = check_against_oracle(g, c, qs, texts, fold=a.io_to_dense_table)
    g.set_query_options(search_seed=False)
    off, t, p, _ = g.locate_raw(*pack_queries(qs))
    assert off.tolist() == want[0].tolist() and t.tolist() == want[1].tolist() and p.tolist() == want[2].tolist()


@pytest.mark.parametrize("seed", range(8))
def test_exact_intervals_through_seed_table_and_inverse_suffix_array(seed):
    """cursors_for_many_queries: a read that occurs exactly once gets [ISA[position], + 1) from its seed entry; every
    other read -- absent (the reference's frozen empty interval), on several rows, with other symbols -- takes the usual
    kernels.  All intervals equal the oracle's, and the switch changes nothing."""
    rng = np.random.default_rng(9300 + seed)
    a = [alph.ascii_dna(), alph.ascii_dna_with_n()][seed % 2]
    symbols = b"ACGTN" if seed % 2 else b"ACGT"
    texts = repetitive_texts(rng, symbols=symbols) if seed % 4 < 2 else random_texts(rng, len_max=8000, symbols=symbols)
    k = [8, 11, 16, True][seed % 4]
    extra = [{}, dict(jump_entry_bytes=16, top_table_depth=6), dict(jump_entry_bytes=0, top_table_depth=0), {}][(seed // 2) % 4]
    depth = [0, 4][seed % 2]
    g = gpu_index(texts, a, depth=depth, seed_symbols=k, inverse_suffix_array=True, seed_load_percent=[None, 100][seed % 2], **extra)
    c = cpu_index(texts, a, depth=depth)
    qs = mixed_queries(rng, texts, 600, 200, 150) + [b"", b"A", b"ACGTACGT"]
    for q in list(qs[:200]):  # one symbol off
        if len(q) > 12:
            at = int(rng.integers(0, len(q)))
            qs.append(q[:at] + bytes([b"ACGT"[(b"ACGT".find(q[at:at + 1]) + 1) % 4]]) + q[at + 1:])
    qs = [q for q in qs if b"N" not in q or depth == 0]
    qbuf, qoff = pack_queries(qs)
    cs, ce = c.cursors_for_many(qbuf, qoff)
    s, e, st = g.cursors_raw(qbuf, qoff)
    assert not st.any()
    assert s.tolist() == cs.tolist() and e.tolist() == ce.tolist()
    g.set_query_options(search_seed=False)
    s2, e2, _ = g.cursors_raw(qbuf, qoff)
    assert s2.tolist() == cs.tolist() and e2.tolist() == ce.tolist()
    # the cursor API on the same index (cursor_for_query -> count / locate)
    for q in qs[:20]:
        cur = g.cursor_for_query(q)
        assert cur.count() == len(naive_search(texts, q, fold=a.io_to_dense_table))


def device_locate(g, qs, compact, max_hits=0, fused=False, packed=False):
    """search -> offsets -> hits on the device through the records calls, with or without the compact results;
    fused = "search": the hit totals come out of the search call itself (gdx_locate_many_search_totals_compact_layout_dev)"""
    import torch

    from genedex_amd.device import DeviceEngine, DeviceQueries

    eng = DeviceEngine(g)
    dq = DeviceQueries.from_host(*pack_queries(qs))
    if packed:
        dq = dq.as_packed(g)
    rec = eng.alloc_records(dq.nq)
    rec.fill_(0x5a5a5a5a)  # (stale bytes: the compact path leaves the records of answered reads untouched)
    cmp_ = eng.alloc_compact(dq.nq) if compact else None
    off = torch.empty(dq.nq + 1, dtype=torch.int64, device="cuda")
    if fused != "search":
        eng.locate_search(dq, rec, compact=cmp_)
    if fused:  # totals -> (host) -> offsets + the compactly answered hits in one pass, then the rest
        sws = torch.empty(max(eng.totals_workspace_bytes(dq.nq), 16), dtype=torch.uint8, device="cuda")
        totals = torch.zeros(2, dtype=torch.int64, device="cuda")
        if fused == "search":
            eng.locate_search_totals(dq, rec, cmp_, sws, totals, max_hits)
            check = torch.zeros(2, dtype=torch.int64, device="cuda")
            sws2 = torch.empty_like(sws)
            eng.locate_totals(rec, dq.nq, sws2, check, max_hits, compact=cmp_)
            assert totals.tolist() == check.tolist()  # the folded totals are those of the separate pass
        else:
            eng.locate_totals(rec, dq.nq, sws, totals, max_hits, compact=cmp_)
        tot, rest = (int(x) for x in totals.tolist())
        hits = torch.full((max(tot, 1), 2), -7, dtype=torch.int32, device="cuda")
        ws = torch.empty(max(eng.locate_workspace_bytes(tot), 16), dtype=torch.uint8, device="cuda")
        eng.locate_offsets_hits(rec, dq.nq, sws, off, tot, rest, hits, ws, max_hits, compact=cmp_)
        torch.cuda.synchronize()
        assert int(off[dq.nq].item()) == tot and rest <= tot
        # the narrow form of the same call: u32 offsets, the same hits
        off32 = torch.full((dq.nq + 1,), -1, dtype=torch.int32, device="cuda")
        hits32 = torch.full_like(hits, -9)
        eng.locate_offsets_hits(rec, dq.nq, sws, off32, tot, rest, hits32, ws, max_hits, compact=cmp_)
        torch.cuda.synchronize()
        assert torch.equal(off32.to(torch.int64), off) and torch.equal(hits32[:tot], hits[:tot])
        # the whole step as ONE call without the round trip (gdx_locate_many_step_compact_layout_dev) into an exact, a roomy
        # and too small a hit buffer: totals, offsets and the hits below the capacity are those of the calls above
        for cap in (tot, tot + 1000, tot // 2):
            for dt in (torch.int64, torch.int32):
                rec2 = torch.full_like(rec, 0x3c3c3c3c)
                cmp2 = torch.full_like(cmp_, -5) if cmp_ is not None else None
                off2 = torch.full((dq.nq + 1,), -1, dtype=dt, device="cuda")
                hits2 = torch.full((cap, 2), -11, dtype=torch.int32, device="cuda")
                tot2 = torch.full((2,), -1, dtype=torch.int64, device="cuda")
                sws3 = torch.full_like(sws, 0x77)
                ws2 = torch.full((max(eng.locate_workspace_bytes(cap), 16),), 0x55, dtype=torch.uint8, device="cuda")
                eng.locate_step(dq, rec2, cmp2, sws3, tot2, off2, hits2, ws2, max_hits)
                torch.cuda.synchronize()
                assert tot2.tolist() == [tot, rest] if cmp_ is not None else int(tot2[0].item()) == tot
                assert torch.equal(off2.to(torch.int64), off)
                assert torch.equal(hits2[: min(cap, tot)], hits[: min(cap, tot)])
                if cap > tot:
                    assert bool((hits2[tot:] == -11).all().item())  # nothing is written beyond the batch's hits
    else:
        eng.locate_offsets(rec, dq.nq, off, max_hits, compact=cmp_)
        torch.cuda.synchronize()
        tot = int(off[dq.nq].item())
        hits = torch.empty((max(tot, 1), 2), dtype=torch.int32, device="cuda")
        ws = torch.empty(max(eng.locate_workspace_bytes(tot), 16), dtype=torch.uint8, device="cuda")
        if tot:
            eng.locate_hits(rec, dq.nq, off, tot, hits, ws, compact=cmp_)
    counts = torch.empty(dq.nq, dtype=torch.int32, device="cuda")
    stat = torch.empty(dq.nq, dtype=torch.uint8, device="cuda")
    eng.unpack_records(rec, dq.nq, counts, stat, compact=cmp_)
    torch.cuda.synchronize()
    answered = int((cmp_[:dq.nq] != -2).sum().item()) if compact else 0
    return (off.cpu().numpy().astype(np.uint64), hits[:tot].cpu().numpy().astype(np.uint32), counts.cpu().numpy().astype(np.uint32),
            stat.cpu().numpy(), answered)


@pytest.mark.parametrize("seed", range(8))
def test_compact_results_equal_records_and_oracle(seed):
    """gdx_locate_many_*_compact_dev: 4 bytes per query wherever the seed kernel answers a read, "see the record" elsewhere
    (every query on an index without a seed table) -- offsets, hits, counts and statuses are those of the records path and
    of the oracle."""
    rng = np.random.default_rng(9500 + seed)
    a = [alph.ascii_dna(), alph.ascii_dna_with_n()][seed % 2]
    symbols = b"ACGTN" if seed % 2 else b"ACGT"
    texts = repetitive_texts(rng, symbols=symbols) if seed % 4 < 2 else random_texts(rng, len_max=8000, symbols=symbols)
    build = [dict(LEAN, seed_symbols=True, full_suffix_array=True), dict(seed_symbols=11), dict(jump_entry_bytes=32),
             dict(LEAN, seed_symbols=9, seed_load_percent=100)][(seed // 2) % 4]
    g = gpu_index(texts, a, **build)
    c = cpu_index(texts, a)
    qs = [q for q in mixed_queries(rng, texts, 600, 200, 120) + [b"", b"ACGTACGTACGT"] if b"N" not in q]
    co, ct, cp = c.locate_many(qs)
    got = {}
    for compact, fused, packed in ((False, False, False), (True, False, False), (True, True, False), (False, True, False),
                                   (True, "search", False), (True, "search", True), (True, True, True)):
        off, hits, counts, stat, answered = device_locate(g, qs, compact, fused=fused, packed=packed)
        assert off.tolist() == co.tolist()
        assert hits[:, 0].tolist() == ct.astype(np.uint32).tolist() and hits[:, 1].tolist() == cp.astype(np.uint32).tolist()
        assert counts.tolist() == np.diff(co).astype(np.uint32).tolist() and not stat.any()
        got[compact] = answered
    if build.get("seed_symbols"):
        assert got[True] > len(qs) // 4  # the seed kernel answered reads compactly
    else:
        assert got[True] == 0
    # a cap on the hits per query applies to the records behind "see the record" alike
    off_a, hits_a, _, _, _ = device_locate(g, qs, False, max_hits=2)
    off_b, hits_b, _, _, _ = device_locate(g, qs, True, max_hits=2)
    assert off_a.tolist() == off_b.tolist() and hits_a.tolist() == hits_b.tolist()
    off_c, hits_c, _, _, _ = device_locate(g, qs, True, max_hits=2, fused=True)
    assert off_a.tolist() == off_c.tolist() and hits_a.tolist() == hits_c.tolist()
    off_d, hits_d, _, _, _ = device_locate(g, qs, True, max_hits=2, fused="search")
    assert off_a.tolist() == off_d.tolist() and hits_a.tolist() == hits_d.tolist()


@pytest.mark.parametrize("blocks", [1, 3])
def test_lane_kernel_blocks_that_take_several_ranges(blocks, monkeypatch):
    """search_seed_lane_kernel with a capped grid (GDX_SEED_LANE_BLOCKS): a block then goes through several ranges, reads parked
    in one range are answered in a later one, and the hit totals the kernel counts per scan tile (the fused search + totals
    call) cross range borders -- offsets, hits and totals stay those of the oracle.  A full table (every bucket has turned
    entries away) makes the parked queue busy; 20 000 reads make ten ranges."""
    import torch

    from genedex_amd.device import DeviceEngine, DeviceQueries

    rng = np.random.default_rng(9900 + blocks)
    a = alph.ascii_dna()
    texts = random_texts(rng, len_max=60000, symbols=b"ACGT")
    g = gpu_index(texts, a, **dict(LEAN, seed_symbols=10, seed_load_percent=100, full_suffix_array=True))
    c = cpu_index(texts, a)
    length = 40
    qs = []
    for _ in range(20000):
        t = texts[int(rng.integers(0, len(texts)))]
        if len(t) >= length and rng.random() < 0.85:
            pos = int(rng.integers(0, len(t) - length + 1))
            qs.append(t[pos:pos + length])
        else:
            qs.append(bytes(b"ACGT"[i] for i in rng.integers(0, 4, length)))
    qbuf, qoff = pack_queries(qs)
    cs, ce = c.cursors_for_many(qbuf, qoff)
    co, ct, cp = c.locate_intervals(cs, ce)
    eng = DeviceEngine(g)
    plain = DeviceQueries.from_host(qbuf, qoff)
    monkeypatch.setenv("GDX_SEED_LANE_BLOCKS", str(blocks))
    for dq in (plain, plain.as_uniform(length), plain.as_packed(g).as_uniform(length)):
        nq = dq.nq
        rec = eng.alloc_records(nq)
        cmp_ = eng.alloc_compact(nq)
        sws = torch.empty(max(eng.totals_workspace_bytes(nq), 16), dtype=torch.uint8, device="cuda")
        totals = torch.zeros(2, dtype=torch.int64, device="cuda")
        eng.locate_search_totals(dq, rec, cmp_, sws, totals)
        tot, rest = (int(x) for x in totals.tolist())
        assert tot == int(co[-1])
        off = torch.full((nq + 1,), -1, dtype=torch.int32, device="cuda")
        hits = torch.full((max(tot, 1), 2), -7, dtype=torch.int32, device="cuda")
        ws = torch.empty(max(eng.locate_workspace_bytes(tot), 16), dtype=torch.uint8, device="cuda")
        eng.locate_offsets_hits(rec, nq, sws, off, tot, rest, hits, ws, compact=cmp_)
        torch.cuda.synchronize()
        assert off.cpu().numpy().astype(np.uint64).tolist() == co.tolist()
        h = hits[:tot].cpu().numpy().astype(np.uint32)
        assert h[:, 0].tolist() == ct.astype(np.uint32).tolist() and h[:, 1].tolist() == cp.astype(np.uint32).tolist()
        assert int((cmp_[:nq] != -2).sum().item()) > nq // 2  # (the lane kernel answered them: the seed chain ran)


@pytest.mark.parametrize("copies", [3, 40, 300])
@pytest.mark.parametrize("wide", [False, True])
def test_seed_intervals_hand_the_fast_kernel_their_state(copies, wide):
    """A k-mer on several rows: the seed kernel lists the read with the entry's interval and the 32 symbols in front of
    the seed packed into its state, the fast kernel finishes it from there (reads of up to k + 32 symbols without a
    look at their bytes; 256+ rows travel as a plain interval and end in the general kernel).  Records and the compact
    / fused results alike."""
    rng = np.random.default_rng(9700 + copies)
    unit = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 60))
    parts = []
    for _ in range(copies):
        parts.append(bytes(b"ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(20, 90)))))
        parts.append(unit)
    texts = [b"".join(parts), bytes(b"ACGT"[i] for i in rng.integers(0, 4, 3000)) + unit[20:]]
    a = alph.ascii_dna()
    g = gpu_index(texts, a, sa_rate=3, seed_symbols=12)
    g.set_query_options(search_fast=2 if wide else 1)
    c = cpu_index(texts, a, sa_rate=3)
    assert g.seed_info()["interval_entries"] > 0
    qs = [unit, unit[20:], unit[-12:], unit[-13:], unit[-44:], unit[-45:]]
    joined = texts[0]
    for _ in range(400):  # reads that end inside or at the end of a copy and start in the unique flank in front of it
        end = int(rng.integers(13, len(joined) + 1))
        qs.append(joined[max(0, end - int(rng.integers(12, 120))):end])
    for q in list(qs[6:206]):  # ... and with one symbol changed somewhere in front of the seed
        if len(q) > 13:
            at = int(rng.integers(0, len(q) - 12))
            qs.append(q[:at] + bytes([b"ACGT"[(b"ACGT".find(q[at:at + 1]) + 1) % 4]]) + q[at + 1:])
    want = check_against_oracle(g, c, qs, texts)
    for compact, fused in ((False, False), (True, False), (True, True)):
        off, hits, counts, stat, _ = device_locate(g, qs, compact, fused=fused)
        assert off.tolist() == want[0].tolist() and not stat.any()
        assert hits[:, 0].tolist() == want[1].astype(np.uint32).tolist() and hits[:, 1].tolist() == want[2].astype(np.uint32).tolist()
        assert counts.tolist() == np.diff(want[0]).astype(np.uint32).tolist()


@pytest.mark.parametrize("n_texts", [1, 5, 200])
def test_compact_results_split_into_text_id_and_position(n_texts):
    """gdx_compact_split_hits_dev (the receiving side of the multi-GPU gather): every compactly answered query's word becomes
    the text id and position of its only hit -- the oracle's -- "none" becomes -1, "see the record" -2."""
    import torch

    from genedex_amd.device import DeviceEngine, DeviceQueries

    rng = np.random.default_rng(9800 + n_texts)
    texts = [bytes(b"ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(30, 4000 // max(n_texts // 20, 1))))) for _ in range(n_texts)]
    texts[0] = texts[0] + texts[0][-40:]  # (a repeat: reads from it say "see the record")
    a = alph.ascii_dna()
    g = gpu_index(texts, a, seed_symbols=10)
    c = cpu_index(texts, a)
    qs = mixed_queries(rng, texts, 801, 200, 60) + [texts[0][-30:], b"ACGTTGCATTTAGGACCA"]
    co, ct, cp = c.locate_many(qs)
    eng = DeviceEngine(g)
    dq = DeviceQueries.from_host(*pack_queries(qs))
    rec, cmp_ = eng.alloc_records(dq.nq), eng.alloc_compact(dq.nq)
    eng.locate_search(dq, rec, compact=cmp_)
    ids = torch.full((dq.nq,), 77, dtype=torch.uint8, device="cuda")
    pos = torch.full((dq.nq,), 77, dtype=torch.int32, device="cuda")
    eng.compact_split_hits(cmp_, dq.nq, ids, pos)
    torch.cuda.synchronize()
    words, ids, pos = cmp_[:dq.nq].cpu().numpy(), ids.cpu().numpy(), pos.cpu().numpy()
    counts = np.diff(co)
    answered = 0
    for q in range(dq.nq):
        if words[q] == -2:
            assert pos[q] == -2 and ids[q] == 0
        elif words[q] == -1:
            assert counts[q] == 0 and pos[q] == -1 and ids[q] == 0
        else:
            assert counts[q] == 1 and (int(ids[q]), int(pos[q])) == (int(ct[co[q]]), int(cp[co[q]])), q
            answered += 1
    assert answered > dq.nq // 4 and (words == -2).any()
    # gdx_compact_exceptions_dev: the queries that say "see the record", in any order; a list too short still counts them all
    want = np.flatnonzero(words == -2)
    for cap in (len(want) + 3, max(len(want) // 2, 1)):
        listed = torch.full((cap,), -5, dtype=torch.int32, device="cuda")
        n = torch.zeros(1, dtype=torch.int64, device="cuda")
        eng.compact_exceptions(cmp_, dq.nq, listed, n)
        torch.cuda.synchronize()
        assert int(n.item()) == len(want)
        got = listed.cpu().numpy()[: min(cap, len(want))]
        assert len(set(got.tolist())) == len(got) and set(got.tolist()) <= set(want.tolist())
        if cap >= len(want):
            assert sorted(got.tolist()) == want.tolist() and (listed.cpu().numpy()[len(want):] == -5).all()
    if n_texts <= 5:
        return
    # more than 256 texts: text ids do not fit a byte
    many = [bytes(b"ACGT"[i] for i in rng.integers(0, 4, 20)) for _ in range(257)]
    g2 = gpu_index(many, a, seed_symbols=8)
    eng2 = DeviceEngine(g2)
    with pytest.raises(Exception, match="256 texts"):
        eng2.compact_split_hits(cmp_, 4, torch.zeros(4, dtype=torch.uint8, device="cuda"), torch.zeros(4, dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("n_texts,nq", [(1, 300), (5, 2048), (24, 9001), (200, 70_000)])
def test_found_bitmap_wire_equals_compact_split_and_reference(n_texts, nq):
    """gdx_wire_pack_dev / gdx_wire_split_dev (the "found bitmap" wire of the multi-GPU gather): a shard packed on the sender's
    side and split on the receiver's says about every read what gdx_compact_split_hits_dev says, its exceptions carry the
    oracle's counts and hits, and the packed bytes are those of the tensor restatement the gloo tests gather
    (dist.wire_pack_reference); capacities that are too small drop what does not fit and report the true numbers."""
    import torch

    from genedex_amd import dist as gdist
    from genedex_amd.device import DeviceEngine, DeviceQueries

    rng = np.random.default_rng(9900 + n_texts)
    texts = [bytes(b"ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(200, 6000)))) for _ in range(n_texts)]
    texts[0] = texts[0] + texts[0][-60:] + texts[0][-60:]  # (repeats: reads from them are exceptions with several hits)
    a = alph.ascii_dna()
    g = gpu_index(texts, a, seed_symbols=10)
    c = cpu_index(texts, a)
    qs = []
    for i in range(nq - 2):  # reads of 12..60 symbols from the texts, one in ten random, one in fifty short (an exception)
        t = texts[int(rng.integers(0, n_texts))]
        ln = int(rng.integers(7, 10)) if i % 50 == 7 else int(rng.integers(12, 61))
        at = int(rng.integers(0, max(len(t) - ln, 1)))
        qs.append(bytes(b"ACGT"[k] for k in rng.integers(0, 4, ln)) if i % 10 == 3 else t[at:at + ln])
    qs += [texts[0][-30:], b"ACGTTGCATTTAGGACCA"]
    co, ct, cp = c.locate_many(qs)
    eng = DeviceEngine(g)
    dq = DeviceQueries.from_host(*pack_queries(qs))
    for narrow in (False, True):
        rec, cmp_ = eng.alloc_records(dq.nq), eng.alloc_compact(dq.nq)
        sws = torch.empty(max(eng.totals_workspace_bytes(dq.nq), 16), dtype=torch.uint8, device="cuda")
        totals = torch.zeros(2, dtype=torch.int64, device="cuda")
        off = torch.empty(dq.nq + 1, dtype=torch.int32 if narrow else torch.int64, device="cuda")
        hits = torch.empty((int(co[-1]) + 5, 2), dtype=torch.int32, device="cuda")
        ws = torch.empty(max(eng.locate_workspace_bytes(hits.shape[0]), 16), dtype=torch.uint8, device="cuda")
        eng.locate_step(dq, rec, cmp_, sws, totals, off, hits, ws)
        torch.cuda.synchronize()
        assert int(totals[0].item()) == int(co[-1])
        words = cmp_[:dq.nq]
        n_exc, n_exc_hits = gdist.exception_sizes(words, off.to(torch.int64), dq.nq)
        n_found = int(((words >= 0) | (words < -2)).sum().item())
        assert n_exc > 0 and n_found > dq.nq // 4
        ids_c = torch.empty(dq.nq, dtype=torch.uint8, device="cuda")
        pos_c = torch.empty(dq.nq, dtype=torch.int32, device="cuda")
        eng.compact_split_hits(cmp_, dq.nq, ids_c, pos_c)
        wws = torch.empty(max(eng.wire_pack_workspace_bytes(dq.nq), 16), dtype=torch.uint8, device="cuda")
        for cap_f, cap_q, cap_h in ((n_found, n_exc, n_exc_hits), (n_found + 9, n_exc + 3, n_exc_hits + 7),
                                    (max(n_found // 2, 1), max(n_exc // 2, 1), max(n_exc_hits // 2, 1))):
            layout = gdist.WireLayout(dq.nq, cap_f, cap_q, cap_h)
            buf = torch.full((layout.nbytes,), 0xA5, dtype=torch.uint8, device="cuda")
            v = layout.views(buf)
            eng.wire_pack(cmp_, off, hits, dq.nq, v, wws)
            torch.cuda.synchronize()
            assert v["meta"].tolist() == [n_exc, n_exc_hits, n_found, 0]
            ref = torch.full_like(buf, 0xA5)
            rv = layout.views(ref)
            gdist.wire_pack_reference(words, off.to(torch.int64), hits, dq.nq, rv)
            tiles = (dq.nq + 2047) // 2048
            assert torch.equal(v["bitmap"][: (dq.nq + 7) // 8], rv["bitmap"][: (dq.nq + 7) // 8])
            assert torch.equal(v["tile_found"][: tiles + 1], rv["tile_found"][: tiles + 1])
            for k, n in (("found_pos", min(cap_f, n_found)), ("exc_q", min(cap_q, n_exc)), ("exc_cnt", min(cap_q, n_exc))):
                assert torch.equal(v[k][:n], rv[k][:n]), k
                assert bool((v[k][n:].view(torch.uint8) == 0xA5).all().item()), k  # nothing written beyond what there is
            if cap_h >= n_exc_hits:
                assert torch.equal(v["exc_ids"][:n_exc_hits], rv["exc_ids"][:n_exc_hits])
                assert torch.equal(v["exc_pos"][:n_exc_hits], rv["exc_pos"][:n_exc_hits])
            if cap_f < n_found or cap_q < n_exc:
                continue  # (a receiver sees the true numbers in meta and refuses: expand_split_results)
            ids = torch.full((dq.nq,), 77, dtype=torch.uint8, device="cuda")
            pos = torch.full((dq.nq,), 77, dtype=torch.int32, device="cuda")
            eng.wire_split(v, dq.nq, ids, pos)
            torch.cuda.synchronize()
            assert torch.equal(ids, ids_c) and torch.equal(pos, pos_c)
            cnt, hh = gdist.expand_split_results(ids, pos, v["exc_cnt"], v["exc_ids"], v["exc_pos"], v["meta"], dq.nq)
            assert cnt.cpu().numpy().tolist() == np.diff(co).tolist()
            assert hh[:, 0].cpu().numpy().tolist() == ct.astype(np.int64).tolist()
            assert hh[:, 1].cpu().numpy().tolist() == cp.astype(np.int64).tolist()
            # fewer bytes than the compact words where most reads are found
            if n_found > 0.5 * dq.nq and dq.nq > 4000:
                assert layout.payload_bytes(dq.nq, n_found, n_exc, n_exc_hits) < 4 * dq.nq + 4 * n_exc + 5 * n_exc_hits


def test_query_buffer_of_exactly_the_contracts_size():
    """gdx.h asks for a query buffer "padded to a multiple of 8 bytes", no more: a batch whose bytes ARE a multiple of 8 lies in
    a device buffer of exactly that size (the ASCII lane kernel's window of the last read once reached 8 bytes past it; it
    now loads its last dword only when the window needs it).  Counts and hits are the oracle's for every alignment of the
    last read's end."""
    import torch

    from genedex_amd.device import DeviceEngine, DeviceQueries

    rng = np.random.default_rng(4242)
    texts = [bytes(b"ACGT"[i] for i in rng.integers(0, 4, 20000)) for _ in range(3)]
    a = alph.ascii_dna()
    g = gpu_index(texts, a, seed_symbols=10, full_suffix_array=True, **LEAN)
    c = cpu_index(texts, a)
    eng = DeviceEngine(g)
    for tail in range(8):  # the last read ends at every residue mod 8; the buffer ends with it, rounded up to 8
        qs = [texts[int(rng.integers(0, 3))][p:p + 56] for p in rng.integers(0, 19000, 200)]
        qs.append(texts[0][100:100 + 56 + tail])
        qbuf, qoff = pack_queries(qs)
        total = int(qoff[-1])
        exact = np.zeros((total + 7) // 8 * 8, dtype=np.uint8)
        exact[:total] = qbuf[:total]
        dq = DeviceQueries(torch.from_numpy(exact).cuda(), torch.from_numpy(qoff.astype(np.int64)).cuda(), len(qs), total)
        assert dq.qbuf.numel() == (total + 7) // 8 * 8
        rec, cmp_ = eng.alloc_records(dq.nq), eng.alloc_compact(dq.nq)
        sws = torch.empty(max(eng.totals_workspace_bytes(dq.nq), 16), dtype=torch.uint8, device="cuda")
        totals = torch.zeros(2, dtype=torch.int64, device="cuda")
        co, ct, cp = c.locate_many(qs)
        off = torch.empty(dq.nq + 1, dtype=torch.int64, device="cuda")
        hits = torch.empty((int(co[-1]) + 1, 2), dtype=torch.int32, device="cuda")
        ws = torch.empty(max(eng.locate_workspace_bytes(hits.shape[0]), 16), dtype=torch.uint8, device="cuda")
        eng.locate_step(dq, rec, cmp_, sws, totals, off, hits, ws)
        torch.cuda.synchronize()
        assert off.cpu().numpy().astype(np.uint64).tolist() == co.tolist(), tail
        h = hits[: int(co[-1])].cpu().numpy().astype(np.uint32)
        assert h[:, 0].tolist() == ct.astype(np.uint32).tolist() and h[:, 1].tolist() == cp.astype(np.uint32).tolist(), tail


def test_seed_entries_are_the_distinct_kmers():
    rng = np.random.default_rng(77)
    a = alph.ascii_dna_with_n()
    texts = repetitive_texts(rng, symbols=b"ACGTN")
    twice = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 400))  # (a stretch that occurs exactly twice, once right at a text's start)
    thrice = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 300))
    gap = lambda: bytes(b"ACGTN"[i] for i in rng.integers(0, 5, 300))  # noqa: E731
    texts = texts + [twice + gap() + twice, thrice + gap() + thrice + gap() + thrice + gap() + thrice[:150]]
    k = 10
    g = gpu_index(texts, a, seed_symbols=k, **LEAN)
    kmers = {}
    for t in texts:
        for i in range(len(t) - k + 1):
            w = t[i:i + k]
            if b"N" not in w:
                kmers.setdefault(w, []).append((t, i))
    single = 0
    for w, occ in kmers.items():
        if len(occ) == 1:
            single += 1
    # k-mers on exactly two rows whose occurrences both have 32 symbols A C G T of their own text in front: a record of their own
    whole = lambda occ: all(i >= 32 and b"N" not in t[i - 32:i] for t, i in occ)  # noqa: E731
    pairs = sum(1 for occ in kmers.values() if len(occ) == 2 and whole(occ))
    quads = sum(1 for occ in kmers.values() if len(occ) in (3, 4) and whole(occ))  # (three or four copies: 64-byte records)
    info = g.seed_info()
    assert info["single_entries"] == single
    assert info["interval_entries"] == len(kmers) - single
    assert info["pair_records"] == pairs and pairs > 0
    assert info["quad_records"] == quads and quads > 0
    assert info["bytes"] == info["buckets"] * 128 + 32 * pairs + 64 * quads
    assert info["max_displacement"] <= 30


def test_reads_longer_than_the_entry_covers_and_text_starts():
    """k + 32 symbols are decided by the entry; the rest is compared with the text units.  Occurrences within 32
    symbols of a text start (or behind an N) have interval entries and take the general route."""
    rng = np.random.default_rng(5)
    a = alph.ascii_dna_with_n()
    texts = [bytes(b"ACGT"[i] for i in rng.integers(0, 4, 3000)) for _ in range(3)]
    texts[1] = texts[1][:500] + b"N" + texts[1][501:]
    g = gpu_index(texts, a, seed_symbols=12, full_suffix_array=True, **LEAN)
    c = cpu_index(texts, a)
    qs = []
    for t in texts:
        for start in (0, 1, 5, 20, 31, 32, 33, 100, 470, 489, 501, 502, 533, 534):
            for ln in (12, 13, 40, 43, 44, 45, 76, 77, 108, 109, 300):
                if start + ln <= len(t):
                    q = t[start:start + ln]
                    if b"N" not in q:
                        qs.append(q)
                        if ln > 50:  # one symbol off, far in front of the seed
                            qs.append(bytes([b"ACGT"[(b"ACGT".find(q[:1]) + 1) % 4]]) + q[1:])
    check_against_oracle(g, c, qs, texts)


def test_queries_with_other_symbols_and_short_ones_go_the_general_way():
    a = alph.ascii_dna_with_n()
    rng = np.random.default_rng(8)
    texts = [bytes(b"ACGTN"[i] for i in rng.integers(0, 5, 4000)), bytes(b"ACGT"[i] for i in rng.integers(0, 4, 2000))]
    g = gpu_index(texts, a, seed_symbols=9)
    c = cpu_index(texts, a)
    qs = mixed_queries(rng, texts, 400, 50, 60, allow_n=True)
    qs += [q[:3] for q in qs[:50]] + [b"N" * 12, b"ACGTNACGTACGT"]
    check_against_oracle(g, c, qs, texts)
    # status codes: a byte outside the alphabet is the reference's panic whatever the seed table says
    # (the search reaches the X: what follows it occurs in the text)
    qbuf, qoff = pack_queries([texts[1][100:105] + b"X" + texts[1][106:140], texts[1][100:140]])
    _, st = g.count_raw(qbuf, qoff, strict=False)
    _, _, cst = c.cursors_single(qbuf, qoff)
    assert st.tolist() == cst.tolist() and st[0] != 0 and st[1] == 0


def test_lookup_depth_beyond_the_seed_switches_it_off():
    a = alph.ascii_dna_with_n()
    rng = np.random.default_rng(9)
    texts = random_texts(rng, len_max=5000, symbols=b"ACGT")
    g = gpu_index(texts, a, depth=9, seed_symbols=8)
    c = cpu_index(texts, a, depth=9)
    qs = [q for q in mixed_queries(rng, texts, 300, 50, 50)]
    check_against_oracle(g, c, qs, texts)


def test_seed_k24_on_a_small_text():
    """the full-size k on a small text: 2^27 buckets (17 GB), nearly all empty"""
    import torch

    if torch.cuda.mem_get_info()[0] < 40e9:
        pytest.skip("needs 40 GB of free device memory")
    a = alph.ascii_dna()
    rng = np.random.default_rng(24)
    texts = repetitive_texts(rng, n_texts=4, unit_len=600, tail=3000)
    g = gpu_index(texts, a, seed_symbols=24, **LEAN)
    c = cpu_index(texts, a)
    info = g.seed_info()
    assert info["k"] == 24 and info["buckets"] == 1 << 27 and info["tag_bits"] == 21
    qs = mixed_queries(rng, texts, 800, 100, 160)
    check_against_oracle(g, c, qs, texts)


def test_bad_seed_options_are_rejected():
    from genedex_amd import GdxError

    a = alph.ascii_dna()
    for kw in (dict(seed_symbols=5), dict(seed_symbols=25), dict(seed_symbols=12, seed_load_percent=10)):
        with pytest.raises(GdxError):
            gpu_index([b"ACGTACGTACGTAAAC"], a, **kw)


def test_seed_table_on_a_loaded_index(tmp_path):
    """The seed table is derived from the index's own BWT and suffix array, so an index loaded from a file gets it like a
    built one (gdx_index_load_ex with build options)."""
    from genedex_amd import FmIndex
    from genedex_amd.index import build_options

    rng = np.random.default_rng(77)
    a = alph.ascii_dna_with_n()
    texts = repetitive_texts(rng, symbols=b"ACGTN")
    plain = gpu_index(texts, a, sa_rate=3, jump_entry_bytes=32)  # (the tables of rounds 1-3: no seed table)
    c = cpu_index(texts, a, sa_rate=3)
    path = tmp_path / "index.gdx"
    plain.save_to_file(path)
    opts = build_options(seed_symbols=11, inverse_suffix_array=True, full_suffix_array=True)
    loaded = FmIndex.load_from_file(path, a, options=opts)
    assert loaded.seed_info()["k"] == 11 and plain.seed_info()["k"] == 0
    qs = [q for q in mixed_queries(rng, texts, 500, 100, 90) if b"N" not in q]
    check_against_oracle(loaded, c, qs, texts, fold=a.io_to_dense_table)
    qbuf, qoff = pack_queries(qs)
    cs, ce = c.cursors_for_many(qbuf, qoff)
    s, e, st = loaded.cursors_raw(qbuf, qoff)
    assert not st.any() and s.tolist() == cs.tolist() and e.tolist() == ce.tolist()


# ---- the reference's own occurrence tables as selectable layouts (gdx_build_options_t.reference_table_layout) ----------

@pytest.mark.parametrize("layout", ["condensed64", "condensed512", "flat64", "flat512"])
def test_reference_table_layouts_bit_for_bit_and_queried_as_they_are(layout):
    """An index built with reference_table_layout holds the reference's table of that variant -- interleaved blocks (the flat
    ones with their block offsets inside) and superblock offsets equal the oracle's restatement of condensed.rs:59-124 /
    flat.rs:59-126 word for word -- and rank / symbol_at / search / locate run on it (GenericTable) with the oracle's results."""
    from helpers import naive_occurrence_columns
    from oracle.oracle import OracleTable

    rng = np.random.default_rng(sum(layout.encode()))
    a = alph.ascii_dna_with_n()
    for total in (0, 47, 48, 49, 495, 496, 497, 512, 65471, 65472, 65473, 65520, 65536, 70001, 200000):
        texts = [bytes(b"ACGTN"[i] for i in rng.integers(0, 5, max(total - 1, 0)))]
        g = gpu_index(texts, a, reference_table_layout=layout)
        c = cpu_index(texts, a)
        kind = layout.rstrip("0123456789")
        bits = int(layout[len(kind):])
        want = OracleTable(c.bwt, 6, kind, bits)
        blocks, sbo = g.export_reference_table()
        assert np.array_equal(blocks, want.blocks), (layout, total)
        assert np.array_equal(sbo, want.superblock_offsets), (layout, total)
        n = c.n
        step = 1 if n < 3000 else 37
        idx = np.array(sorted(set(range(0, n + 1, step)) | {n} | {i for i in (47, 48, 49, 495, 496, 497, 511, 512, 513, 65471, 65472,
                                                                             65519, 65520, 65535, 65536) if i <= n}), dtype=np.uint64)
        cols = naive_occurrence_columns(c.bwt, 6)
        for s in range(6):
            got = g.rank_many(np.full(idx.size, s, dtype=np.uint8), idx)
            assert np.array_equal(got, cols[s][idx.astype(np.int64)]), (layout, total, s)
        if n:
            at = np.arange(0, n, step, dtype=np.uint64)
            assert np.array_equal(g.symbol_at_many(at), c.bwt[at.astype(np.int64)])
        assert np.array_equal(g.export_bwt(), c.bwt)
        if total >= 495:
            qs = mixed_queries(rng, texts, 200, 60, 40, allow_n=True)
            check_against_oracle(g, c, qs, texts, fold=a.io_to_dense_table)
            qbuf, qoff = pack_queries(qs)
            cs, ce = c.cursors_for_many(qbuf, qoff)
            s_, e_, st = g.cursors_raw(qbuf, qoff)
            assert not st.any() and s_.tolist() == cs.tolist() and e_.tolist() == ce.tolist()
            # the Condensed / Block64 export (what gdx_index_save writes) of any layout is the same table
            b64 = OracleTable(c.bwt, 6, "condensed", 64)
            eb, ebo, esb = g.export_condensed_table()
            assert np.array_equal(eb, b64.blocks) and np.array_equal(ebo, b64.block_offsets) and np.array_equal(esb, b64.superblock_offsets)


@pytest.mark.parametrize("structures", ["seed+sa", "seed+jump32", "seed+sa, no tables", "seed+walk", "default shape", "default shape, no pair records"])
def test_reads_on_two_rows_carry_both_positions(structures, monkeypatch):
    """A read that ends on exactly two rows (every stretch of this text occurs twice) leaves search_fast_kernel4 as a resolved
    record OF TWO -- {second position, second + 2, first position, resolved}: end - start is the count, no row is named
    (kernels.hpp) -- where the search has both occurrences at hand: search_fast_kernel4 on 32-byte jump entries (they carry
    SA[row]), search_verify_kernel4 on the full suffix array.  The locate kernels write both hits from it, in the reference's
    order: locate_by_query_kernel (every slot open), locate_stream_kernel and the store pass's inline location (the one-call
    step's flagged chunks); with a limit of one hit such reads get no slots.  The records are looked at, so that the path
    cannot go unused; an index that walks to its suffix-array values makes no such records and gives the same hits."""
    import torch

    from genedex_amd.device import DeviceEngine, DeviceQueries

    rng = np.random.default_rng(660)
    unit = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 30000))
    other = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 9000))
    texts = [unit[:17000] + other[:4000], other[4000:] + unit]  # unit[:17000] twice; the rest of unit and `other` once
    a = alph.ascii_dna()
    opts = {"seed+sa": dict(seed_symbols=12, text_units=True, full_suffix_array=True),
            "seed+jump32": dict(seed_symbols=12, text_units=True, jump_entry_bytes=32),
            "seed+sa, no tables": dict(seed_symbols=12, text_units=True, full_suffix_array=True, pair_lines=False, jump_entry_bytes=0,
                                       top_table_depth=0),
            "seed+walk": dict(seed_symbols=12, text_units=True, jump_entry_bytes=0, top_table_depth=0, pair_lines=False),
            # (what gdx_index_build makes of nothing: k = 16 here; a read with up to 32 symbols in front of its seed is decided by
            # the 32-byte record of its two-copy k-mer -- IndexView::seed_pairs --, a longer one by SA line and text lines)
            "default shape": dict(), "default shape, no pair records": dict()}[structures]
    if structures == "default shape, no pair records":
        monkeypatch.setenv("GDX_SEARCH_SEED_PAIRS", "0")
    g = gpu_index(texts, a, **opts)
    if structures.startswith("default shape"):
        assert DeviceEngine(g).aux_info()["default_shape"] and g.seed_info()["pair_records"] > 10000
    g.set_query_options(search_fast=2)  # (the fast path is off by default on a text this repetitive: wide_permille > 500)
    c = cpu_index(texts, a)
    qs = []
    for _ in range(5000):
        ln = int(rng.integers(30, 70))
        at = int(rng.integers(0, len(unit) - ln))
        qs.append(unit[at:at + ln])
    co, ct, cp = c.locate_many(qs)
    counts = np.diff(co)
    assert int((counts == 2).sum()) > 2000 and int((counts == 1).sum()) > 1000
    eng = DeviceEngine(g)
    dq = DeviceQueries.from_host(*pack_queries(qs))
    for compact in (False, True):
        rec = eng.alloc_records(dq.nq)
        cw = eng.alloc_compact(dq.nq) if compact else None
        eng.locate_search(dq, rec, compact=cw)
        off = torch.empty(dq.nq + 1, dtype=torch.int64, device="cuda")
        eng.locate_offsets(rec, dq.nq, off, compact=cw)
        torch.cuda.synchronize()
        total = int(off[dq.nq].item())
        hits = torch.empty((total, 2), dtype=torch.int32, device="cuda")
        ws = torch.empty(max(eng.locate_workspace_bytes(total), 16), dtype=torch.uint8, device="cuda")
        eng.locate_hits(rec, dq.nq, off, total, hits, ws, compact=cw)
        torch.cuda.synchronize()
        assert off.cpu().numpy().astype(np.uint64).tolist() == co.tolist(), (structures, compact)
        h = hits.cpu().numpy().astype(np.uint32)
        assert h[:, 0].tolist() == ct.tolist() and h[:, 1].tolist() == cp.tolist(), (structures, compact)
        r = rec[:dq.nq].cpu().numpy().astype(np.uint32)
        two = counts == 2
        resolved_of_two = two & ((r[:, 3] >> 22) & 1).astype(bool) & ((r[:, 1] - r[:, 0]) == 2)
        if structures != "seed+walk":
            assert int(resolved_of_two.sum()) > 1500, (structures, compact, int(resolved_of_two.sum()))
    # the whole step in one call (flagged chunks: the stream kernel / the store pass's inline location); with a limit of one hit
    # per read the reads on two rows are counted but get no slots (the device calls' max_hits), the others stay as they are
    for max_hits in (0, 1):
        rec, cw = eng.alloc_records(dq.nq), eng.alloc_compact(dq.nq)
        sws = torch.empty(max(eng.totals_workspace_bytes(dq.nq), 16), dtype=torch.uint8, device="cuda")
        totals = torch.zeros(2, dtype=torch.int64, device="cuda")
        off32 = torch.empty(dq.nq + 1, dtype=torch.int32, device="cuda")
        hits = torch.full((int(co[-1]) + 3, 2), -1, dtype=torch.int32, device="cuda")
        ws = torch.empty(max(eng.locate_workspace_bytes(hits.shape[0]), 16), dtype=torch.uint8, device="cuda")
        eng.locate_step(dq, rec, cw, sws, totals, off32, hits, ws, max_hits=max_hits)
        torch.cuda.synchronize()
        keep = np.ones(len(qs), dtype=bool) if max_hits == 0 else counts <= max_hits
        want_off = np.concatenate([[0], np.cumsum(np.where(keep, counts, 0))]).astype(np.uint64)
        sel = np.repeat(keep, counts.astype(np.int64))
        assert off32.cpu().numpy().astype(np.uint64).tolist() == want_off.tolist(), (structures, max_hits)
        h = hits[: int(want_off[-1])].cpu().numpy().astype(np.uint32)
        assert h[:, 0].tolist() == ct[sel].tolist() and h[:, 1].tolist() == cp[sel].tolist(), (structures, max_hits)


@pytest.mark.parametrize("records", [True, False])
def test_reads_from_repeats_of_two_to_four_copies_are_decided_by_their_records(records, monkeypatch):
    """The default shape on a text whose stretches occur once, twice, three and four times, one copy of each family with a
    changed symbol every 40: a count / locate read with up to 32 symbols in front of its seed (k = 16 here) is decided by the
    32- or 64-byte record of its k-mer's rows (IndexView::seed_pairs / seed_quads; search_verify_kernel4) -- none, some or all
    of the copies, in the reference's order --, a longer one by suffix-array line and text lines; GDX_SEARCH_SEED_PAIRS=0 is
    the same search without the records.  Counts and hits against the oracle, compact results and the one-call step included."""
    from genedex_amd.device import DeviceEngine

    if not records:
        monkeypatch.setenv("GDX_SEARCH_SEED_PAIRS", "0")
    rng = np.random.default_rng(4242)
    acgt = lambda n_: bytes(b"ACGT"[i] for i in rng.integers(0, 4, n_))  # noqa: E731
    fams = [(acgt(int(rng.integers(150, 400))), copies) for copies in (2, 3, 4, 2, 3, 4, 3, 4) for _ in range(6)]
    parts, texts = [], []
    for fam, copies in fams:
        for c in range(copies):
            piece = bytearray(fam)
            if c == copies - 1:  # the last copy differs every 40 symbols: reads match some of the rows only
                for j in range(7, len(piece), 40):
                    piece[j] = b"ACGT"[(b"ACGT".index(piece[j]) + 1) % 4]
            parts.append(bytes(piece) + acgt(int(rng.integers(20, 120))))
    order = rng.permutation(len(parts))
    third = len(order) // 3
    for lo_, hi_ in ((0, third), (third, 2 * third), (2 * third, len(order))):
        texts.append(acgt(50) + b"".join(parts[i] for i in order[lo_:hi_]))
    a = alph.ascii_dna()
    g = gpu_index(texts, a)
    info = g.seed_info()
    assert DeviceEngine(g).aux_info()["default_shape"] and info["pair_records"] > 500 and info["quad_records"] > 1000
    c = cpu_index(texts, a)
    qs = []
    for _ in range(6000):
        t = texts[int(rng.integers(0, len(texts)))]
        ln = int(rng.integers(info["k"], info["k"] + 45))
        at = int(rng.integers(0, len(t) - ln))
        qs.append(t[at:at + ln])
    co, _, _ = c.locate_many(qs)
    counts = np.diff(co)
    assert all(int((counts == m).sum()) > 200 for m in (1, 2, 3, 4))
    check_against_oracle(g, c, qs, texts)


@pytest.mark.parametrize("structures", ["seed+sa", "seed+jump32"])
def test_hit_sparse_chunk_spanning_millions_of_reads(structures):
    """A chunk of 2048 hit slots of a hit-sparse batch spans millions of reads.  locate_stream_kernel (SA[row] one fetch away,
    compact results, flagged chunks) once kept a slot's query number relative to the chunk's first query in 21 bits beside the
    slot number: beyond 2^21 - 1 reads in a chunk the marks overflowed and the open slots got another query's hits (round-5
    advisor).  The reads of many hits at both ends (more than 2048 rows each: the store pass cannot place them inline) open
    chunks whose queries lie 2.3 M reads apart; through the one-call step and through the narrow host call, whose chunks hold
    more than 2^21 packed reads."""
    import ctypes as C

    import torch

    from genedex_amd import _lib
    from genedex_amd.device import DeviceEngine, DeviceQueries

    rng = np.random.default_rng(2121)
    body = bytes(b"ACGT"[i] for i in rng.integers(0, 4, 60000))
    texts = [body[:30000] + b"A" * 3000 + body[30000:], b"C" * 2600 + body[100:9000]]
    a = alph.ascii_dna()
    opts = {"seed+sa": dict(seed_symbols=12, text_units=True, full_suffix_array=True, **LEAN),
            "seed+jump32": dict(seed_symbols=12, text_units=True, jump_entry_bytes=32)}[structures]
    g = gpu_index(texts, a, **opts)
    c = cpu_index(texts, a)
    ln, n_absent = 32, 2_300_000
    absent = rng.integers(0, 4, (n_absent, ln), dtype=np.uint8)
    sym = np.frombuffer(b"ACGT", dtype=np.uint8)
    rows = [np.frombuffer(b"A" * ln, dtype=np.uint8)[None, :],                 # 2969 rows
            np.frombuffer(body[500:500 + ln], dtype=np.uint8)[None, :],
            sym[absent[: n_absent // 2]],
            np.frombuffer(body[40000:40000 + ln], dtype=np.uint8)[None, :],
            sym[absent[n_absent // 2:]],
            np.frombuffer(b"C" * ln, dtype=np.uint8)[None, :],                 # 2569 rows
            np.frombuffer(body[7000:7000 + ln], dtype=np.uint8)[None, :]]
    q2d = np.ascontiguousarray(np.concatenate(rows, axis=0))
    nq = q2d.shape[0]
    qbuf = np.zeros(nq * ln + 64, dtype=np.uint8)
    qbuf[: nq * ln] = q2d.reshape(-1)
    qoff = (np.arange(nq + 1, dtype=np.uint64) * np.uint64(ln))
    cs, ce = c.cursors_for_many(qbuf, qoff)
    co, ct, cp = c.locate_intervals(cs, ce)
    counts = np.diff(co)
    assert int(counts[0]) > 2048 and int(counts[nq - 2]) > 2048 and int(co[-1]) < 20000
    eng = DeviceEngine(g)
    dq = DeviceQueries.from_host(qbuf, qoff)
    rec, cw = eng.alloc_records(dq.nq), eng.alloc_compact(dq.nq)
    sws = torch.empty(max(eng.totals_workspace_bytes(dq.nq), 16), dtype=torch.uint8, device="cuda")
    totals = torch.zeros(2, dtype=torch.int64, device="cuda")
    off32 = torch.empty(dq.nq + 1, dtype=torch.int32, device="cuda")
    hits = torch.full((int(co[-1]) + 5, 2), -1, dtype=torch.int32, device="cuda")
    ws = torch.empty(max(eng.locate_workspace_bytes(hits.shape[0]), 16), dtype=torch.uint8, device="cuda")
    eng.locate_step(dq, rec, cw, sws, totals, off32, hits, ws)
    torch.cuda.synchronize()
    assert np.array_equal(off32.cpu().numpy().astype(np.uint64), co), structures
    h = hits[: int(co[-1])].cpu().numpy().astype(np.uint32)
    assert np.array_equal(h[:, 0], ct.astype(np.uint32)) and np.array_equal(h[:, 1], cp.astype(np.uint32)), structures
    # the host call on the batch as 2-bit codes of uniform length: one chunk holds all 2.3 M reads
    lib = _lib.load()
    packed = np.zeros(int(lib.gdx_packed_bytes(nq * ln)), dtype=np.uint8)
    n_exc = C.c_uint64(0)
    _lib.check(lib.gdx_pack_queries(g._h, qbuf.ctypes.data_as(_lib.u8p), qoff.ctypes.data_as(_lib.u64p), nq,
                                    packed.ctypes.data_as(_lib.u8p), None, 0, C.byref(n_exc)))
    assert n_exc.value == 0
    o4, t4, p4, st4 = g.locate_layout32_raw(packed, None, nq, packed=True, uniform_len=ln)
    assert not st4.any()
    assert np.array_equal(o4.astype(np.uint64), co) and np.array_equal(t4, ct.astype(np.uint32)) and \
        np.array_equal(p4, cp.astype(np.uint32)), structures
