#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the CPU oracle, wider than the pytest suite.

Every round draws a configuration (alphabet, number and size of texts, repeat structure, sampling rate, lookup
depth, jump entry size, forced top-table depth, lanes per query, load policy, index storage: gdx_build_options_t /
gdx_query_options_t), builds the index with both implementations and compares intervals, statuses, hits (host API;
device API with and without hints; the fused record path with its lazy tails; cursors extended chunk by chunk with
device-side active lists) bit for bit.  Test infrastructure: it uses oracle/ as the checker.

usage: python tests/parity_sweep.py [rounds, default 60] [seed, default 1]  -> one JSON line
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tests/)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genedex_amd import FmIndexConfig, alphabet  # noqa: E402
from genedex_amd.device import DeviceEngine, DeviceQueries  # noqa: E402
from oracle.oracle import OracleIndex, pack_queries  # noqa: E402


def draw_texts(rng, symbols, total, n_texts, mode):
    cuts = np.sort(rng.integers(0, total + 1, n_texts - 1)) if n_texts > 1 else np.array([], dtype=np.int64)
    lens = np.diff(np.concatenate([[0], cuts, [total]]))
    texts = []
    for ln in lens:
        ln = int(ln)
        if mode == "random":
            t = bytes(symbols[i] for i in rng.integers(0, len(symbols), ln))
        elif mode == "repeats":  # a short unit repeated with a few substitutions
            unit = bytes(symbols[i] for i in rng.integers(0, len(symbols), int(rng.integers(1, 200))))
            t = bytearray((unit * (ln // len(unit) + 1))[:ln])
            for pos in rng.integers(0, max(ln, 1), ln // 300):
                t[pos] = symbols[int(rng.integers(0, len(symbols)))]
            t = bytes(t)
        else:  # "runs": long runs of single symbols
            t = bytearray()
            while len(t) < ln:
                t += bytes([symbols[int(rng.integers(0, len(symbols)))]]) * int(rng.integers(1, 400))
            t = bytes(t[:ln])
        texts.append(t)
    return texts


def draw_queries(rng, texts, symbols, n, max_len):
    qs = []
    nonempty = [t for t in texts if t]
    for _ in range(n):
        r = rng.random()
        if r < 0.7 and nonempty:
            t = nonempty[int(rng.integers(0, len(nonempty)))]
            pos = int(rng.integers(0, len(t)))
            q = bytearray(t[pos:pos + int(rng.integers(0, max_len + 1))])
            if q and rng.random() < 0.15:  # one substitution somewhere
                q[int(rng.integers(0, len(q)))] = symbols[int(rng.integers(0, len(symbols)))]
            qs.append(bytes(q))
        else:
            qs.append(bytes(symbols[i] for i in rng.integers(0, len(symbols), int(rng.integers(0, max_len + 1)))))
    return qs


ALPHABETS = [
    ("ascii_dna", alphabet.ascii_dna, b"ACGT"),
    ("ascii_dna_with_n", alphabet.ascii_dna_with_n, b"ACGTACGTACGTACGTN"),
    ("ascii_dna_iupac_as_dna_with_n", alphabet.ascii_dna_iupac_as_dna_with_n, b"ACGTACGTACGTRYN"),
    ("ascii_dna_iupac", alphabet.ascii_dna_iupac, b"ACGTNRYKMSWBDHV"),
    ("ascii_amino_acid", alphabet.ascii_amino_acid, b"ACDEFGHIKLMNPQRSTVWY"),
]


def one_round(rng, stats):
    name, make, symbols = ALPHABETS[int(rng.choice(len(ALPHABETS), p=[0.2, 0.45, 0.15, 0.1, 0.1]))]
    a = make()
    total = int(rng.choice([300, 5_000, 60_000, 400_000, 2_000_000], p=[0.15, 0.25, 0.3, 0.2, 0.1]))
    n_texts = int(rng.choice([1, 2, 7, 60, 800], p=[0.3, 0.2, 0.2, 0.2, 0.1]))
    mode = str(rng.choice(["random", "repeats", "runs"], p=[0.6, 0.25, 0.15]))
    rate = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 33]))
    k = a.num_searchable_dense_symbols()
    depth = int(rng.integers(0, 6 if k <= 4 else 3))
    build = {"jump_entry_bytes": [32, 32, 16, 8, 0][int(rng.integers(0, 5))],
             "top_table_depth": [None, None, 0, 3, 6, 8, 10][int(rng.integers(0, 7))],
             "pair_lines": [None, None, None, False][int(rng.integers(0, 4))],
             # the text itself and the full suffix array (count / locate searches compare with the text when there is no jump
             # table) in a third of the rounds
             "text_units": [None, None, True][int(rng.integers(0, 3))],
             "full_suffix_array": [None, True][int(rng.integers(0, 2))],
             # seed table (its own kernel, then the fast / verify / general kernels on what it lists) and inverse suffix array
             # (exact intervals through the seed table) in two fifths of the rounds
             "seed_symbols": [None, None, None, True, 8, 12][int(rng.integers(0, 6))],
             "seed_load_percent": [None, 100, 45][int(rng.integers(0, 3))],
             "inverse_suffix_array": [None, True][int(rng.integers(0, 2))]}
    # (round 6) three rounds in ten leave every structure to the library: the default shape (seed table + text units + full and
    # inverse suffix array + pair lines + top table, no jump table) wherever the alphabet allows it, with the top table's depth
    # forced now and then -- exact intervals and cursors then take the text route of search_exact_kernel4
    if rng.random() < 0.3:
        build = {"top_table_depth": [None, None, 0, 3, 6, 10][int(rng.integers(0, 6))],
                 "seed_load_percent": [None, 100][int(rng.integers(0, 2))]}
    query = {"search_lanes": [4, 4, 8][int(rng.integers(0, 3))], "load_policy": int(rng.integers(0, 2)),
             "length_schedule": int(rng.integers(0, 2)), "locate_jump_walk": int(rng.integers(0, 4)) != 0,
             "search_defer_after": [None, 0, 1, 2, 5][int(rng.integers(0, 5))],
             "search_fast": int(rng.integers(0, 3)), "search_exact": int(rng.integers(0, 3)) != 0,
             "search_seed": int(rng.integers(0, 4)) != 0}
    storage = str(rng.choice(["i32", "u32"]))
    cfg = {"alphabet": name, "total": total, "n_texts": n_texts, "mode": mode, "sa_rate": rate, "depth": depth,
           "storage": storage, **build, **query}
    texts = draw_texts(rng, symbols, total, n_texts, mode)
    g = (FmIndexConfig(storage).suffix_array_sampling_rate(rate).lookup_table_depth(depth)
         .acceleration_structures(**build).construct_index(texts, a))
    g.set_query_options(**query)
    c = OracleIndex.build(texts, a.io_to_dense_table, a.num_dense_symbols(), k, sa_rate=rate, lookup_depth=depth,
                          width=-32 if storage == "i32" else 32)
    qs = draw_queries(rng, texts, symbols, int(rng.integers(200, 3000)), int(rng.choice([8, 40, 70, 150, 400])))
    qbuf, qoff = pack_queries(qs)
    s, e, st = g.cursors_raw(qbuf, qoff, strict=False)
    cs, ce, cst = c.cursors_single(qbuf, qoff)
    assert st.tolist() == cst.tolist(), ("status", cfg)
    ok = st == 0
    assert s[ok].tolist() == cs[ok].tolist() and e[ok].tolist() == ce[ok].tolist(), ("intervals", cfg)
    small = ok & ((ce - cs) < 5000)
    ls, le = np.where(small, cs, 0), np.where(small, ce, 0)
    co, ct, cp = c.locate_intervals(ls, le)
    off, t, p = g.locate_intervals_raw(ls, le)
    assert off.tolist() == co.tolist() and t.tolist() == ct.tolist() and p.tolist() == cp.tolist(), ("locate", cfg)
    # device path: search with hints -> offsets -> hinted locate; and the same intervals without hints
    keep = np.nonzero(small)[0]
    sub = [qs[i] for i in keep]
    if sub:
        qb2, qo2 = pack_queries(sub)
        dq = DeviceQueries.from_host(qb2, qo2)
        eng = DeviceEngine(g)
        for hint in (True, False):
            out = eng.alloc_outputs(dq.nq, hint=hint)
            eng.search(dq, out)
            eng.hit_offsets(out, dq.nq)
            torch.cuda.synchronize()
            tot = int(out["hit_offsets"][dq.nq].item())
            hits = torch.empty((max(tot, 1), 2), dtype=torch.int32, device="cuda")
            ws = torch.empty(max(eng.locate_workspace_bytes(tot), 16), dtype=torch.uint8, device="cuda")
            if tot:
                eng.locate(out, dq.nq, tot, hits, ws)
            torch.cuda.synchronize()
            assert out["start"].cpu().numpy().astype(np.uint32).tolist() == cs[keep].astype(np.uint32).tolist(), ("dev start", cfg)
            assert out["end"].cpu().numpy().astype(np.uint32).tolist() == ce[keep].astype(np.uint32).tolist(), ("dev end", cfg)
            h = hits[:tot].cpu().numpy().astype(np.uint32)
            assert h[:, 0].tolist() == ct.astype(np.uint32).tolist() and h[:, 1].tolist() == cp.astype(np.uint32).tolist(), \
                ("dev hits", hint, cfg)
            if hint:
                stats["hinted_queries"] += int(((out["hint"] & 0xffffffff) != 0xffffffff).sum().item())
        # fused count + locate over search records (lazy tails): counts, offsets and hits are the reference's
        rec = eng.alloc_records(dq.nq)
        off_t = torch.empty(dq.nq + 1, dtype=torch.int64, device="cuda")
        eng.locate_search(dq, rec)
        eng.locate_offsets(rec, dq.nq, off_t)
        torch.cuda.synchronize()
        tot = int(off_t[dq.nq].item())
        exp_off = np.concatenate([[0], np.cumsum((ce[keep] - cs[keep]).astype(np.uint64))]).astype(np.uint64)
        assert off_t.cpu().numpy().astype(np.uint64).tolist() == exp_off.tolist(), ("rec offsets", cfg)
        hits = torch.empty((max(tot, 1), 2), dtype=torch.int32, device="cuda")
        ws = torch.empty(max(eng.locate_workspace_bytes(tot), 16), dtype=torch.uint8, device="cuda")
        if tot:
            eng.locate_hits(rec, dq.nq, off_t, tot, hits, ws)
        counts = torch.empty(dq.nq, dtype=torch.int32, device="cuda")
        stat = torch.empty(dq.nq, dtype=torch.uint8, device="cuda")
        eng.unpack_records(rec, dq.nq, counts, stat)
        torch.cuda.synchronize()
        h = hits[:tot].cpu().numpy().astype(np.uint32)
        assert h[:, 0].tolist() == ct.astype(np.uint32).tolist() and h[:, 1].tolist() == cp.astype(np.uint32).tolist(), \
            ("rec hits", cfg)
        assert counts.cpu().numpy().astype(np.uint32).tolist() == (ce[keep] - cs[keep]).astype(np.uint32).tolist(), ("rec counts", cfg)
        assert not stat.any().item(), ("rec status", cfg)
        # the same through the compact results and the fused totals -> offsets + hits calls (gdx_locate_many_*_compact_dev)
        rec2 = eng.alloc_records(dq.nq)
        rec2.fill_(0x5a5a5a5a)
        cmp2 = eng.alloc_compact(dq.nq)
        off2 = torch.empty(dq.nq + 1, dtype=torch.int64, device="cuda")
        eng.locate_search(dq, rec2, compact=cmp2)
        sws = torch.empty(max(eng.totals_workspace_bytes(dq.nq), 16), dtype=torch.uint8, device="cuda")
        totals = torch.zeros(2, dtype=torch.int64, device="cuda")
        eng.locate_totals(rec2, dq.nq, sws, totals, compact=cmp2)
        tot2, rest2 = (int(x) for x in totals.tolist())
        assert tot2 == tot, ("compact total", cfg)
        hits2 = torch.full((max(tot2, 1), 2), -7, dtype=torch.int32, device="cuda")
        ws2 = torch.empty(max(eng.locate_workspace_bytes(tot2), 16), dtype=torch.uint8, device="cuda")
        eng.locate_offsets_hits(rec2, dq.nq, sws, off2, tot2, rest2, hits2, ws2, compact=cmp2)
        counts2 = torch.empty(dq.nq, dtype=torch.int32, device="cuda")
        stat2 = torch.empty(dq.nq, dtype=torch.uint8, device="cuda")
        eng.unpack_records(rec2, dq.nq, counts2, stat2, compact=cmp2)
        torch.cuda.synchronize()
        assert torch.equal(off2, off_t), ("compact offsets", cfg)
        assert torch.equal(hits2[:tot2], hits[:tot]), ("compact hits", cfg)
        assert torch.equal(counts2, counts) and not stat2.any().item(), ("compact counts", cfg)
        # round 5: the whole step as ONE call without a host round trip into a roomy and into too small a buffer, the shard
        # through the found-bitmap wire and back, the narrow host-pointer call
        for cap5 in (tot2 + int(rng.integers(1, 50)), max(tot2 // 3, 1)):
            rec5 = eng.alloc_records(dq.nq)
            rec5.fill_(0x77777777)
            cmp5 = eng.alloc_compact(dq.nq)
            narrow5 = bool(rng.integers(0, 2))
            off5 = torch.full((dq.nq + 1,), -1, dtype=torch.int32 if narrow5 else torch.int64, device="cuda")
            hits5 = torch.full((cap5, 2), -3, dtype=torch.int32, device="cuda")
            tot5 = torch.full((2,), -1, dtype=torch.int64, device="cuda")
            sws5 = torch.empty_like(sws)
            ws5 = torch.empty(max(eng.locate_workspace_bytes(cap5), 16), dtype=torch.uint8, device="cuda")
            eng.locate_step(dq, rec5, cmp5, sws5, tot5, off5, hits5, ws5)
            torch.cuda.synchronize()
            assert tot5.tolist() == [tot2, rest2], ("step totals", cap5, cfg)
            assert torch.equal(off5.to(torch.int64), off_t), ("step offsets", cap5, cfg)
            n5 = min(cap5, tot2)
            assert torch.equal(hits5[:n5], hits[:n5]), ("step hits", cap5, cfg)
            stats["one_call_steps"] = stats.get("one_call_steps", 0) + 1
        if int(g.info.num_texts) <= 256 and dq.nq:
            from genedex_amd import dist as gdist

            n_exc5, n_exc_hits5 = gdist.exception_sizes(cmp2[:dq.nq], off2, dq.nq)
            n_found5 = int(((cmp2[:dq.nq] >= 0) | (cmp2[:dq.nq] < -2)).sum().item())
            layout5 = gdist.WireLayout(dq.nq, max(n_found5, 1), max(n_exc5, 1), max(n_exc_hits5, 1))
            buf5 = torch.full((layout5.nbytes,), 0x5A, dtype=torch.uint8, device="cuda")
            v5 = layout5.views(buf5)
            wws5 = torch.empty(max(eng.wire_pack_workspace_bytes(dq.nq), 16), dtype=torch.uint8, device="cuda")
            eng.wire_pack(cmp2, off2, hits2, dq.nq, v5, wws5)
            ids5 = torch.empty(dq.nq, dtype=torch.uint8, device="cuda")
            pos5 = torch.empty(dq.nq, dtype=torch.int32, device="cuda")
            eng.wire_split(v5, dq.nq, ids5, pos5)
            torch.cuda.synchronize()
            cnt5, hh5 = gdist.expand_split_results(ids5, pos5, v5["exc_cnt"], v5["exc_ids"], v5["exc_pos"], v5["meta"], dq.nq)
            assert torch.equal(cnt5.to(torch.int32), counts), ("wire counts", cfg)
            assert torch.equal(hh5, hits[:tot]), ("wire hits", cfg)
            stats["wire_round_trips"] = stats.get("wire_round_trips", 0) + 1
        if tot2 < (1 << 31):
            o32, t32, p32, s32 = g.locate_layout32_raw(qb2, qo2, dq.nq)
            assert o32.astype(np.uint64).tolist() == exp_off.tolist() and not s32.any(), ("narrow host offsets", cfg)
            assert t32.tolist() == ct.astype(np.uint32).tolist() and p32.tolist() == cp.astype(np.uint32).tolist(), ("narrow host hits", cfg)
            stats["narrow_host_calls"] = stats.get("narrow_host_calls", 0) + 1
        # ... and from the batch as 2-bit codes (gdx_query_layout_t), the totals out of the search call itself, narrow offsets
        dq_p = None
        if int(g.info.table_layout) == 0 and k >= 4:  # (packed queries: the rank-line layout with dense 1..4 searchable)
            try:
                dq_p = dq.as_packed(g)
            except ValueError:
                dq_p = None  # (a query with a symbol outside the four: not expressible in 2 bits)
        if dq_p is not None:
            rec3 = eng.alloc_records(dq.nq)
            rec3.fill_(0x3c3c3c3c)
            cmp3 = eng.alloc_compact(dq.nq)
            totals3 = torch.zeros(2, dtype=torch.int64, device="cuda")
            eng.locate_search_totals(dq_p, rec3, cmp3, sws, totals3)
            assert int(totals3[0].item()) == tot2, ("packed totals", cfg)
            tot3, rest3 = (int(x) for x in totals3.tolist())
            off3 = torch.full((dq.nq + 1,), -1, dtype=torch.int32, device="cuda")
            hits3 = torch.full((max(tot3, 1), 2), -7, dtype=torch.int32, device="cuda")
            eng.locate_offsets_hits(rec3, dq.nq, sws, off3, tot3, rest3, hits3, ws2, compact=cmp3)
            torch.cuda.synchronize()
            assert torch.equal(off3.to(torch.int64), off_t), ("packed offsets", cfg)
            assert torch.equal(hits3[:tot3], hits[:tot]), ("packed hits", cfg)
            stats["packed_batches"] = stats.get("packed_batches", 0) + 1
        # a UNIFORM batch (every read the same length, no offsets read: gdx_query_layout_t.uniform_len) of reads drawn from the
        # texts and at random, as IO symbols and as 2-bit codes, through the fused search + totals call: the oracle's hits
        if int(g.info.table_layout) == 0 and k >= 4:
            ulen = int(rng.choice([1, 9, 24, 33, 50, 56, 57, 90, 151]))
            us = []
            nonempty = [t for t in texts if len(t) >= ulen]
            for _ in range(int(rng.integers(100, 1500))):
                if nonempty and rng.random() < 0.75:
                    t = nonempty[int(rng.integers(0, len(nonempty)))]
                    pos = int(rng.integers(0, len(t) - ulen + 1))
                    us.append(t[pos:pos + ulen])
                else:
                    us.append(bytes(symbols[i] for i in rng.integers(0, len(symbols), ulen)))
            ub, uo = pack_queries(us)
            ucs, uce, ust = c.cursors_single(ub, uo)
            uok = (ust == 0) & ((uce - ucs) < 5000)
            us = [q for q, keep_ in zip(us, uok) if keep_]
            if us:
                ub, uo = pack_queries(us)
                ucs, uce = c.cursors_for_many(ub, uo)
                uco, uct, ucp = c.locate_intervals(ucs, uce)
                plain_u = DeviceQueries.from_host(ub, uo)
                forms = [plain_u.as_uniform(ulen)]
                try:
                    forms.append(plain_u.as_packed(g).as_uniform(ulen))
                except ValueError:
                    pass
                for dq_u in forms:
                    rec4 = eng.alloc_records(dq_u.nq)
                    cmp4 = eng.alloc_compact(dq_u.nq)
                    sws4 = torch.empty(max(eng.totals_workspace_bytes(dq_u.nq), 16), dtype=torch.uint8, device="cuda")
                    tot4 = torch.zeros(2, dtype=torch.int64, device="cuda")
                    eng.locate_search_totals(dq_u, rec4, cmp4, sws4, tot4)
                    t4, r4 = (int(x) for x in tot4.tolist())
                    assert t4 == int(uco[-1]), ("uniform total", ulen, dq_u.packed, cfg)
                    off4 = torch.full((dq_u.nq + 1,), -1, dtype=torch.int32, device="cuda")
                    hits4 = torch.full((max(t4, 1), 2), -7, dtype=torch.int32, device="cuda")
                    ws4 = torch.empty(max(eng.locate_workspace_bytes(t4), 16), dtype=torch.uint8, device="cuda")
                    eng.locate_offsets_hits(rec4, dq_u.nq, sws4, off4, t4, r4, hits4, ws4, compact=cmp4)
                    torch.cuda.synchronize()
                    assert off4.cpu().numpy().astype(np.uint64).tolist() == uco.tolist(), ("uniform offsets", ulen, dq_u.packed, cfg)
                    h4 = hits4[:t4].cpu().numpy().astype(np.uint32)
                    assert h4[:, 0].tolist() == uct.astype(np.uint32).tolist() and h4[:, 1].tolist() == ucp.astype(np.uint32).tolist(), \
                        ("uniform hits", ulen, dq_u.packed, cfg)
                    stats["uniform_batches"] = stats.get("uniform_batches", 0) + 1
        stats["compact_answers"] = stats.get("compact_answers", 0) + int((cmp2[:dq.nq] != -2).sum().item())
        stats["lazy_or_hinted_records"] = stats.get("lazy_or_hinted_records", 0) + int((rec[:dq.nq, 2] != -1).sum().item())
        stats["masked_records"] = stats.get("masked_records", 0) + int((((rec[:dq.nq, 3] >> 23) & 1) == 1).sum().item())
        # the same queries through the batched cursor API, fed in chunks of `chunk` symbols from the right, with
        # device-side active lists: identical intervals (cursor.rs:34-51 applied symbol by symbol)
        chunk = int(rng.choice([1, 5, 8, 16, 32, 50]))
        m = dq.nq
        qoff_t = dq.qoff
        beg, endq = qoff_t[:-1].clone(), qoff_t[1:].clone()
        cur_s = torch.zeros(m, dtype=torch.int32, device="cuda")
        cur_e = torch.full((m,), g.total_text_len(), dtype=torch.int64, device="cuda").to(torch.int32)
        cur_st = torch.zeros(m, dtype=torch.uint8, device="cuda")
        act = [torch.arange(m, dtype=torch.int32, device="cuda"), torch.empty(m, dtype=torch.int32, device="cuda")]
        n_act = [torch.tensor([m], dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")]
        hi_edge = endq.clone()
        rounds = 0
        while True:
            lo_edge = torch.maximum(beg, hi_edge - chunk)
            eng.cursor_extend_strings(cur_s, cur_e, dq.qbuf, lo_edge, hi_edge, m, cur_st, act[0], n_act[0], act[1], n_act[1])
            hi_edge = lo_edge
            act.reverse()
            n_act.reverse()
            rounds += 1
            if int(n_act[0].item()) == 0 or bool((hi_edge <= beg).all().item()):
                break
        torch.cuda.synchronize()
        # a cursor whose query still had symbols left when it emptied is frozen exactly like the fused search
        assert cur_s.cpu().numpy().astype(np.uint32).tolist() == cs[keep].astype(np.uint32).tolist(), ("cursor start", chunk, cfg)
        assert cur_e.cpu().numpy().astype(np.uint32).tolist() == ce[keep].astype(np.uint32).tolist(), ("cursor end", chunk, cfg)
        assert not cur_st.any().item()
        stats["cursor_rounds"] = stats.get("cursor_rounds", 0) + rounds
    stats["default_shape_rounds"] = stats.get("default_shape_rounds", 0) + int(DeviceEngine(g).aux_info()["default_shape"])
    stats["queries"] += len(qs)
    stats["hits"] += int(co[-1])
    stats["status_nonzero"] += int((st != 0).sum())
    return cfg


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    stats = {"queries": 0, "hits": 0, "status_nonzero": 0, "hinted_queries": 0}
    t0 = time.time()
    seen = {}
    for _ in range(rounds):
        cfg = one_round(rng, stats)
        for key in ("alphabet", "mode", "jump_entry_bytes", "top_table_depth", "pair_lines", "search_lanes", "load_policy",
                    "length_schedule", "locate_jump_walk", "search_defer_after", "search_fast", "search_exact", "text_units",
                    "full_suffix_array", "seed_symbols", "seed_load_percent", "inverse_suffix_array", "search_seed", "sa_rate",
                    "depth"):
            seen.setdefault(key, {}).setdefault(str(cfg.get(key, "library default")), 0)
            seen[key][str(cfg.get(key, "library default"))] += 1
    print(json.dumps({"rounds": rounds, "seed": seed, "all_equal": True,
                      "seconds": round(time.time() - t0, 1), **stats, "configurations_seen": seen}))


if __name__ == "__main__":
    main()
