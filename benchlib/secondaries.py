"""benchlib.secondaries -- secondary design points, never `value`: the speed-vs-HBM ladder, exact intervals, workload 5, configs[1], the reference's own tables, the genome-like text (split out of bench.py in round 6; bench.py re-exports everything)."""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .common import *  # noqa: F401,F403
from .pmc import *  # noqa: F401,F403
from .line import *  # noqa: F401,F403
from .baseline import *  # noqa: F401,F403
from .multi import *  # noqa: F401,F403
from .end_to_end import *  # noqa: F401,F403

__all__ = ['secondaries', 'cfg2_secondary', 'genome_like_secondary', 'exact_intervals_secondary', 'mixed_length_secondary']

def secondaries(torch, owned, io_text, lengths, alpha, queries, base_counts, nq, do_locate, args, wl, pmc_ref=None,
                pmc_text=None, e2e=None, res=None, pmc_lookup=None):
    """Secondary design points, never `value`.  (1) The speed-vs-HBM ladder: the same step with the jump / top tables
    rebuilt at other sizes on the same suffix array (gdx_index_rebuild_aux), down to the arrays with the reference's
    information content only; every rung must reproduce the headline's counts exactly.  (2) BASELINE.json configs[4]:
    50 M reads of mixed length through the fused call and through the batched cursor API.  (3) The reference's
    lookup-table knob at the depth BASELINE.md names."""
    from genedex_amd.device import DeviceEngine, build_index_from_device_text

    eng, index = owned["eng"], owned["index"]
    res = res if res is not None else []
    seed_family = args.index in ("default", "seed")
    if not args.no_extras and seed_family:
        # BASELINE configs[4] -- fused and through the cursor API -- and exact intervals of the headline's reads ON THE HEADLINE
        # INDEX itself (the default shape serves them through seed entry / text / ISA; the lean seed index of rounds 3b-5 has
        # nothing for them but the rank lines: one pass there says so)
        if args.index == "default":
            res.append(exact_intervals_secondary(torch, eng, queries, base_counts, nq, "on the HEADLINE index"))
            res[-1]["aux_structures"] = eng.aux_info()
            res[-1]["index_bytes"] = int(index.info.device_bytes)
        res.append(mixed_length_secondary(torch, eng, io_text, lengths, light=args.index != "default", headline=True))
        res[-1]["aux_structures"] = eng.aux_info()
        res[-1]["index_bytes"] = int(index.info.device_bytes)
    text = dict(jump_entry_bytes=0, pair_lines=False, text_units=True)  # the rest of a read against the text at SA[row]
    ladder = [("top16_sa_text", dict(top_table_depth=16, full_suffix_array=True, **text)),
              ("top15_sa_text", dict(top_table_depth=15, full_suffix_array=True, **text)),
              ("top14_sa_text", dict(top_table_depth=14, full_suffix_array=True, **text)),
              ("top14_text", dict(top_table_depth=14, **text)),
              ("top12_text", dict(top_table_depth=12, **text)),
              ("top14_jump32", dict(top_table_depth=14, jump_entry_bytes=32)),
              ("top16_jump16", dict(jump_entry_bytes=16)),
              ("top14_jump16", dict(top_table_depth=14, jump_entry_bytes=16)),
              ("top12_jump8", dict(top_table_depth=12, jump_entry_bytes=8)),
              ("pair_lines_only", dict(top_table_depth=0, jump_entry_bytes=0)),
              ("reference_arrays_only", dict(top_table_depth=0, jump_entry_bytes=0, pair_lines=False))]
    if seed_family:
        # the headline is the default shape; the lean seed index (the headline of rounds 3b-5: what the inverse suffix array,
        # pair lines and top table of the default shape cost a count + locate step -- nothing -- and what they buy the other
        # calls), the tables of rounds 1-3 and the seed table without the full suffix array come first
        ladder = ([("seed_lean_no_isa_no_pairs", dict(SEED_INDEX))] if args.index == "default" else []) + \
                 [("tables_top16_jump32_pairs", dict(jump_entry_bytes=32)),
                  ("seed_text_no_sa", {k: v for k, v in SEED_INDEX.items() if k != "full_suffix_array"})] + ladder
    if args.no_extras:
        ladder = [r for r in ladder if r[0] in ("tables_top16_jump32_pairs", "top16_sa_text", "top14_text", "pair_lines_only",
                                                "reference_arrays_only")]
    for name, opts in ladder:
        t0 = time.time()
        index.rebuild_aux(**opts)
        t_aux = time.time() - t0
        ms, s_ms, l_ms, counts = time_config(torch, eng, queries, nq, do_locate, args)
        same = bool(torch.equal(counts, base_counts))
        if not same:
            raise SystemExit(f"PARITY FAILURE: secondary configuration {name} changed the counts")
        r = {"name": name, "aux_structures": eng.aux_info(), "value": nq / (ms / 1e3), "unit": "queries/s",
             "ms_per_step": ms, "search_ms": s_ms, "locate_ms": l_ms, "counts_identical_to_headline": same,
             "aux_rebuild_seconds": t_aux, "index_bytes": int(index.info.device_bytes)}
        if name == ("tables_top16_jump32_pairs" if seed_family else "top16_sa_text"):
            t_txt = traffic_of(pmc_text, "search_fast_kernel|search_pair_kernel" if seed_family
                               else "search_verify_kernel|search_kernel")
            if t_txt:  # measured HBM traffic of this rung's search (PMC child passes of this run on the same configuration)
                r["roofline"] = {"bound": "hbm", "kernel": t_txt["kernel"], "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                                 "traffic": t_txt["bytes"], "achieved": t_txt["bytes"] / (s_ms / 1e3) / 1e9,
                                 "frac": t_txt["bytes"] / (s_ms / 1e3) / 1e9 / HBM_PEAK_GBPS,
                                 "dram_read_requests_per_query": t_txt.get("read_requests", 0) / nq, "avg_launch_ms": s_ms}
        if name == "reference_arrays_only":
            # like-for-like roofline: the reference's information content, its algorithmic bytes per LF step
            lf_steps, _, _ = eng.search_step_stats(queries)
            b = queries.total_bytes + 60 * lf_steps + 8 * nq
            r["roofline_reference_layout"] = {
                "bound": "hbm", "kernel": "search_kernel<QuadLineTable, 4>", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                "algorithmic_bytes_per_launch": b, "achieved": b / (s_ms / 1e3) / 1e9,
                "frac": b / (s_ms / 1e3) / 1e9 / HBM_PEAK_GBPS,
                "note": "algorithmic bytes of SURVEY.md 8d / kernel time / 8 TB/s on the 4.65 GB index without any "
                        "acceleration structure (every 30-byte rank costs one 128-byte DRAM request there)"}
            t_ref = traffic_of(pmc_ref, "search_kernel")
            if t_ref:  # the same kernel's measured HBM traffic (PMC child passes of this run on the same configuration)
                rl = r["roofline_reference_layout"]
                rl["algorithmic_ratio"] = rl.pop("frac")
                rl["traffic"] = t_ref["bytes"]
                rl["achieved"] = t_ref["bytes"] / (s_ms / 1e3) / 1e9
                rl["frac"] = rl["achieved"] / HBM_PEAK_GBPS
                rl["dram_read_requests_per_query"] = t_ref.get("read_requests", 0) / nq
                rl["note"] = ("frac = measured HBM traffic / kernel time / 8 TB/s on the 4.65 GB index without any acceleration "
                              "structure; algorithmic_ratio = the logical bytes of SURVEY.md 8d over the same time (every "
                              "30-byte rank costs one 128-byte request)")
        log(f"[bench] secondary {name}: {r}")
        res.append(r)
        del counts
    if seed_family:
        # every structure at once (214 GB): the tables of rounds 1-3 for the exact-interval and cursor calls below, plus seed
        # table and inverse suffix array (exact intervals of reads that occur once: seed entry + one ISA fetch)
        index.rebuild_aux(**FULL_INDEX)
    else:
        index.rebuild_aux(jump_entry_bytes=32)  # the tables of rounds 1-3
    if not args.no_extras:
        if args.index == "seed" and e2e:  # the packed-query calls run on the pair-line kernels (the lean index has none)
            import numpy as np
            ms_t, s_ms_t, _, _ = time_config(torch, eng, queries, nq, do_locate, args)
            res.append({"name": "packed_queries_end_to_end (index with every structure)", "aux_structures": eng.aux_info(),
                        "index_bytes": int(index.info.device_bytes),
                        **packed_end_to_end(np, torch, index, queries, nq, base_counts, e2e["pcie_h2d_GBps"], e2e["pcie_d2h_GBps"],
                                            s_ms_t), "device_search_ms_on_ascii_input": s_ms_t})
        res.append(exact_intervals_secondary(torch, eng, queries, base_counts, nq, "on the index with every structure"))
        res[-1]["aux_structures"] = eng.aux_info()
        res[-1]["index_bytes"] = int(index.info.device_bytes)
        res.append(mixed_length_secondary(torch, eng, io_text, lengths))
        res[-1]["aux_structures"] = eng.aux_info()
        res[-1]["index_bytes"] = int(index.info.device_bytes)
    # the reference's lookup-table knob needs its own index (the lookup tables are part of the reference's arrays): the
    # like-for-like rung again -- the reference's arrays and NOTHING else (no seed table, text units, suffix array, pair
    # lines, jump or top table) -- with its lookup tables of depth 10 and 13 in front of the LF steps
    # (lookup_table.rs:51-161): a len-50 read starts from the interval of its last d symbols, one 8-byte fetch, and takes
    # 50 - d steps instead of 50.  Algorithmic bytes per SURVEY 8(d): len + 8 (the lookup entry) + 60 x steps + 8.
    owned.clear()
    del eng, index
    torch.cuda.empty_cache()
    eng2 = index2 = counts = None
    for depth in ((args.secondary_depth,) if args.no_extras else LOOKUP_RUNGS):
        del eng2, index2, counts
        torch.cuda.empty_cache()
        t0 = time.time()
        index2 = build_index_from_device_text(io_text, lengths, alpha, sa_rate=args.sa_rate, lookup_depth=depth,
                                              index_storage=wl["storage"], options=build_options_of(args, **REFERENCE_ARRAYS))
        apply_query_options(index2, args)
        t_build = time.time() - t0
        eng2 = DeviceEngine(index2)
        ms, s_ms, l_ms, counts = time_config(torch, eng2, queries, nq, do_locate, args)
        same = bool(torch.equal(counts, base_counts))
        if not same:
            raise SystemExit(f"PARITY FAILURE: the lookup-depth-{depth} secondary changed the counts")
        lf_steps, _, _ = eng2.search_step_stats(queries)
        b = queries.total_bytes + 8 * nq + 60 * lf_steps + 8 * nq
        r = {"name": f"reference_arrays_d{depth}", "lookup_depth": depth, "aux_structures": eng2.aux_info(),
             "value": nq / (ms / 1e3), "unit": "queries/s", "ms_per_step": ms, "search_ms": s_ms, "locate_ms": l_ms,
             "counts_identical_to_headline": same, "index_build_seconds": t_build, "index_bytes": int(index2.info.device_bytes),
             "lf_steps_per_query": lf_steps / nq,
             "roofline": {"bound": "hbm", "kernel": "search_kernel<QuadLineTable, 4>", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                          "algorithmic_bytes_per_launch": b, "achieved_algorithmic": b / (s_ms / 1e3) / 1e9,
                          "frac_algorithmic": b / (s_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, "avg_launch_ms": s_ms}}
        t_l = traffic_requests_of((pmc_lookup or {}).get(depth), "search_kernel", (pmc_lookup or {}).get("queries", LOOKUP_PMC_READS))
        if t_l:  # DRAM requests of the same kernel on LOOKUP_PMC_READS of these reads (a PMC child pass of this run)
            rq = t_l["read_requests"] / t_l["queries"]
            r["roofline"].update({"dram_read_requests_per_query": rq, "dram_write_requests_per_query": t_l["write_requests"] / t_l["queries"],
                                  "l2_hit_rate": t_l["l2_hits"] / max(t_l["l2_requests"], 1), "pmc_queries": t_l["queries"],
                                  # every DRAM request of this GPU moves 128 bytes (profiles/r01/fetch_size_calibration.json)
                                  "traffic_from_requests": 128.0 * (t_l["read_requests"] + t_l["write_requests"]) / t_l["queries"] * nq,
                                  "frac_traffic_from_requests": 128.0 * (t_l["read_requests"] + t_l["write_requests"]) / t_l["queries"] * nq
                                  / (s_ms / 1e3) / 1e9 / HBM_PEAK_GBPS})
        res.append(r)
        log(f"[bench] secondary {res[-1]}")
    if not args.no_extras and wl["total"] >= 1 << 24:
        del eng2, index2, counts
        torch.cuda.empty_cache()
        # the reference's own occurrence tables, exactly as genedex lays them out and queried in place (one lane per query,
        # gdx_build_options_t.reference_table_layout): its speed / memory points on this GPU, on a fifth of the batch
        from genedex_amd.index import build_options as _bo
        n_sub = min(nq, 20_000_000)
        q_sub = queries.slice(0, n_sub)
        for layout in ("condensed64", "flat64"):
            t0 = time.time()
            ix_r = build_index_from_device_text(io_text, lengths, alpha, sa_rate=args.sa_rate, lookup_depth=args.lookup_depth,
                                                index_storage=wl["storage"], options=_bo(reference_table_layout=layout))
            t_build = time.time() - t0
            eng_r = DeviceEngine(ix_r)
            ms, s_ms, l_ms, counts_r = time_config(torch, eng_r, q_sub, n_sub, do_locate, args, steps=2)
            same = bool(torch.equal(counts_r, base_counts[:n_sub]))
            if not same:
                raise SystemExit(f"PARITY FAILURE: the {layout} table changed the counts")
            res.append({"name": f"reference_table_{layout} (genedex's own layout, queried in place)", "queries": n_sub,
                        "value": n_sub / (ms / 1e3), "unit": "queries/s", "ms_per_step": ms, "search_ms": s_ms, "locate_ms": l_ms,
                        "counts_identical_to_headline": same, "index_build_seconds": t_build,
                        "index_bytes": int(ix_r.info.device_bytes)})
            log(f"[bench] secondary {res[-1]}")
            del eng_r, ix_r, counts_r
            torch.cuda.empty_cache()
        # BASELINE configs[1]: 256 MB text, 10 M len-50 reads, count() -- parity is tests/test_gpu_parity.py's
        # test_full_size_properties_workload2; this is its throughput on the library's default index
        try:
            res.append(cfg2_secondary(torch, alpha, args))
        except Exception as e:  # noqa: BLE001
            log(f"[bench] cfg2 secondary failed: {e!r}")
        # the genome-like text: a read from a repeat goes on from its seed entry's interval -- the verify kernel compares up to four
        # rows with the text, the pair-line kernel steps the wide ones -- and its hits are consecutive rows of the full suffix array
        # (round 6: on the index the headline runs on -- the library's defaults.  Rounds 3b-5 gave this text an index of its own,
        # seed table AND 32-byte jump entries AND depth-16 top table AND full suffix array, 200 GB: 9.0-9.3 G q/s there)
        res.append(genome_like_secondary(torch, alpha, wl, args))
    return res


def cfg2_secondary(torch, alpha, args, steps=20):
    """BASELINE.json configs[1]: one text of 2^28 symbols, 10 M len-50 reads (90 % sampled), count() on one GPU, the library's
    default index (i32 storage: n < 2^31), reads resident as IO symbols + u64 offsets; every sampled read must be found."""
    from genedex_amd.device import DeviceEngine, DeviceQueries, build_index_from_device_text, synth_text

    w = WORKLOADS["cfg2"]
    dev = torch.device("cuda", torch.cuda.current_device())
    text = synth_text(w["total"], seed=42, n_per_million=10_000, device=dev)
    t0 = time.time()
    index = build_index_from_device_text(text, [w["total"]], alpha, sa_rate=args.sa_rate, lookup_depth=args.lookup_depth,
                                         index_storage=w["storage"])
    t_build = time.time() - t0
    eng = DeviceEngine(index)
    nq = w["nq"]
    q = DeviceQueries.synth(text, [w["total"]], nq, w["len_min"], w["len_max"], w["sampled_ppm"], seed=43)
    out = {}
    for form, qq in (("ascii", q), ("packed+uniform", q.as_packed(index).as_uniform(w["len_min"]))):
        runner = StepRunner(torch, eng, qq, nq, False, "records")
        runner.size()
        for _ in range(3):
            runner.step(0, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.step(0, False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        counts = runner.counts(runner.outs[0])
        found = int((counts > 0).sum().item())
        out[form] = {"ms_per_step": ms, "value": nq / (ms / 1e3), "queries_found": found}
        del runner
    if out["ascii"]["queries_found"] != out["packed+uniform"]["queries_found"] or out["ascii"]["queries_found"] < 0.895 * nq:
        raise SystemExit(f"PARITY FAILURE: cfg2 counts {out}")
    r = {"name": "cfg2_256MB_10M_len50_count (BASELINE configs[1])", "text_len": w["total"], "queries": nq, "op": "count",
         "value": out["ascii"]["value"], "unit": "queries/s", "ms_per_step": out["ascii"]["ms_per_step"],
         "input": "IO symbols + u64 offsets", "packed_input": out["packed+uniform"], "queries_found": out["ascii"]["queries_found"],
         "index_bytes": int(index.info.device_bytes), "index_build_seconds": t_build, "aux_structures": eng.aux_info()}
    log(f"[bench] secondary {r}")
    return r


def genome_like_secondary(torch, alpha, wl, args, max_hits=1000):
    """The hard case for the jump tables: a text of the same size with the repeat structure of a genome (30 %
    segmental duplications with 0.5 % divergence, tandem repeats, poly-A, long N gaps; genome_like_text) instead of
    i.i.d. symbols, the same 100 M len-50 reads (90 % drawn from the text).  Reads from repeats have intervals that
    stay wider than four rows after the top table and fall back to pair-line steps; reads with more than `max_hits`
    occurrences (poly-A, tandem repeats: up to tens of millions each) are counted but not located, as a read mapper
    would do.  Hits are verified against the text."""
    from genedex_amd.device import (DeviceEngine, DeviceQueries, build_index_from_device_text, genome_like_text,
                                    hg38_text_lengths)

    dev = torch.device("cuda", torch.cuda.current_device())
    total, nq = wl["total"], wl["nq"]
    t0 = time.time()
    text = genome_like_text(total, dev)
    lengths = hg38_text_lengths(total, wl["n_texts"])
    torch.cuda.synchronize()
    t_text = time.time() - t0
    t0 = time.time()
    index = build_index_from_device_text(text, lengths, alpha, sa_rate=args.sa_rate, lookup_depth=args.lookup_depth,
                                         index_storage=wl["storage"], options=build_options_of(args))
    apply_query_options(index, args)
    t_build = time.time() - t0
    eng = DeviceEngine(index)
    q = DeviceQueries.synth(text, lengths, nq, wl["len_min"], wl["len_max"], wl["sampled_ppm"], seed=43)
    # the step over 16-byte records with the per-query limit: search -> offsets -> read-back of the total -> hits.  (The
    # headline's compact results + one-call step cost more than they save here -- a third of the reads is listed for the next
    # kernel and keeps its record anyway: 13.3 against 12.4 ms on one box, profiles/r05/README.md; `path` = records switches)
    # the batch in the headline's form -- 2-bit codes of uniform length -- unless asked otherwise (genome_input=ascii) or a read
    # holds a symbol 2 bits cannot name: 11.6 against 12.2 ms (profiles/r05/README.md)
    q_run, input_form = q, "ascii"
    if getattr(args, "genome_input", None) != "ascii" and wl["len_min"] == wl["len_max"]:
        try:
            q_run, input_form = q.as_packed(index).as_uniform(wl["len_min"]), "packed+uniform"
        except ValueError:
            pass
    runner = StepRunner(torch, eng, q_run, nq, True, getattr(args, "genome_path", None) or "records16")
    runner.max_hits = max_hits
    total_hits = runner.size()
    runner.step(0, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        runner.step(0, True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    runner.check_totals()
    runner.widen_offsets()
    total_hits = runner.total_hits
    out = runner.outs[0]
    off, hits = out["hit_offsets"], runner.hits[0]
    counts = runner.counts(out).to(torch.int64) & 0xFFFFFFFF
    ev = [(a, b, c) for (a, b), (_, c) in zip(runner.ev_search, runner.ev_locate)]
    chk = verify_hits(torch, text, lengths, q, {"hit_offsets": off}, hits, total_hits, nq, 1_000_000) if total_hits else {}
    if chk and chk["hits_checked"] != chk["hits_matching_text"]:
        raise SystemExit(f"PARITY FAILURE on the genome-like text: {chk}")
    # oracle gate at bench size, like the headline's: a >= 1 M-query prefix against the CPU restatement on the same index
    # (counts of every query; hits, in order, of the queries under the limit)
    import numpy as np

    from oracle import oracle as orc

    n_gate = min(nq, 1_000_000)
    avail, _ = host_threads()
    cpu = oracle_from_index(np, index, alpha, args, wl, avail)
    qb, qo = q.host_slice(0, n_gate)
    cs, ce = cpu.cursors_for_many(qb, qo, n_threads=avail)
    g_counts = counts[:n_gate].cpu().numpy().astype(np.uint64)
    gate_counts = bool(np.array_equal(g_counts, ce - cs))
    keep = (ce - cs) <= max_hits
    co, ct, cp = cpu.locate_intervals(np.where(keep, cs, 0), np.where(keep, ce, 0), n_threads=avail)
    n_h = int(co[-1])
    gh = hits[:n_h].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    goff = off[:n_gate + 1].cpu().numpy().astype(np.uint64)
    gate_hits = bool(np.array_equal(goff, co) and np.array_equal(gh[:, 0], ct.astype(np.int64))
                     and np.array_equal(gh[:, 1], cp.astype(np.int64)))
    del cpu
    if not gate_counts or not gate_hits:
        raise SystemExit(f"PARITY FAILURE on the genome-like text vs the CPU oracle: counts {gate_counts}, hits {gate_hits}")
    lf_steps, fetches, slots = eng.search_step_stats(q)
    res = {"name": "genome_like_text (repeats, tandem repeats, poly-A, N gaps)", "text_len": total, "queries": nq,
           "oracle_gate": {"queries": n_gate, "hits": n_h, "counts_identical": gate_counts, "hits_identical": gate_hits},
           "text_checksum": int(text[: total // 8 * 8].view(torch.int64).sum().item()),  # the same text in every run
           "max_hits_located_per_query": max_hits, "input": input_form, "value": nq / (ms / 1e3), "unit": "queries/s", "ms_per_step": ms,
           "search_ms": sum(a.elapsed_time(b) for a, b, _ in ev) / len(ev),
           "scan_and_locate_ms": sum(b.elapsed_time(c) for _, b, c in ev) / len(ev),
           "queries_found": int((counts > 0).sum().item()), "occurrences_of_all_queries": int(counts.sum().item()),
           "queries_over_the_limit": int((counts > max_hits).sum().item()), "hits_located": total_hits,
           "mean_hits_per_located_query": total_hits / max(int(((counts > 0) & (counts <= max_hits)).sum().item()), 1),
           # how the located hits spread over interval sizes: {rows per query: [queries, hits]} -- a query of 2..31 rows costs a
           # whole 128-byte line of the suffix array for 8..124 bytes of it
           "located_queries_by_hits": {name: [int(((counts >= lo) & (counts <= hi)).sum().item()),
                                              int(counts[(counts >= lo) & (counts <= hi)].sum().item())]
                                       for name, lo, hi in (("1", 1, 1), ("2-3", 2, 3), ("4-31", 4, 31), ("32-255", 32, 255),
                                                            (f"256-{max_hits}", 256, max_hits))},
           "lf_steps": lf_steps, "line_fetches_per_query_exact_mode": fetches / nq,
           "active_lane_fraction_exact_mode": fetches / slots if slots else None,
           "index_build_seconds": t_build, "text_seconds": t_text, "build_stats": index.build_stats(),
           "aux_structures": eng.aux_info(), **chk}
    log(f"[bench] secondary {res}")
    return res


def exact_intervals_secondary(torch, eng, queries, base_counts, nq, where):
    """exact intervals of the headline's reads (cursors_for_many_queries): bit-identical to the reference's, frozen empty ones
    included (tests); here their widths must be the headline's counts"""
    xo = eng.alloc_outputs(nq)
    eng.search(queries, xo)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        eng.search(queries, xo)
    ev[1].record()
    torch.cuda.synchronize()
    x_ms = ev[0].elapsed_time(ev[1]) / 3
    x_same = bool(torch.equal(torch.sub(xo["end"], xo["start"]), base_counts))
    if not x_same:
        raise SystemExit(f"PARITY FAILURE: exact interval widths {where} differ from the headline's counts")
    r = {"name": f"exact_intervals_len50 (cursors_for_many_queries on the headline's reads) {where}", "queries": nq,
         "ms": x_ms, "value": nq / (x_ms / 1e3), "unit": "queries/s", "widths_identical_to_headline_counts": x_same}
    log(f"[bench] secondary {r}")
    return r


def mixed_length_secondary(torch, eng, io_text, lengths, light=False, headline=False):
    """BASELINE.json configs[4]: 50 M reads of length 20..150, 70 % sampled / 30 % random (early termination), through
    (a) the fused cursors_for_many_queries call and (b) the batched cursor API: cursor_empty, then
    gdx_cursor_extend_front_strings_dev with 32 symbols per call and device-side active lists.  Identical intervals."""
    from genedex_amd.device import DeviceQueries

    w = WORKLOADS["mixed"]
    nq = w["nq"]
    dev = io_text.device
    q = DeviceQueries.synth(io_text, lengths, nq, w["len_min"], w["len_max"], w["sampled_ppm"], seed=47)
    out = eng.alloc_outputs(nq)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    fused_ms = timed(lambda: eng.search(q, out))
    lf_steps, fetches, slots = eng.search_step_stats(q)
    chunk = 32
    n = eng.index.total_text_len()
    beg, end = q.qoff[:-1], q.qoff[1:]
    cur_s = torch.empty(nq, dtype=torch.int32, device=dev)
    cur_e = torch.empty(nq, dtype=torch.int32, device=dev)
    cur_st = torch.empty(nq, dtype=torch.uint8, device=dev)
    act = [torch.empty(nq, dtype=torch.int32, device=dev) for _ in range(2)]
    n_act = [torch.empty(1, dtype=torch.int32, device=dev) for _ in range(2)]
    edges = [torch.empty(nq, dtype=torch.int64, device=dev) for _ in range(2)]
    rounds = -(-w["len_max"] // chunk)
    live = {"strings": [], "chunks": []}

    def reset():
        cur_s.zero_()
        cur_e.fill_(n if n < (1 << 31) else n - (1 << 32))  # cursor_empty for every read
        cur_st.zero_()

    def cursor_api_strings(record_live=False):
        """gdx_cursor_extend_front_strings_dev: the caller computes the chunk edges; a read stays in the live list as
        long as its interval is non-empty (it gets empty strings once it has ended)"""
        reset()
        hi = end
        a, na = None, None  # first call: all cursors
        for r in range(rounds):
            lo = edges[r % 2]
            torch.sub(hi, chunk, out=lo)
            torch.maximum(lo, beg, out=lo)
            eng.cursor_extend_strings(cur_s, cur_e, q.qbuf, lo, hi, nq, cur_st, a, na, act[r % 2], n_act[r % 2])
            a, na = act[r % 2], n_act[r % 2]
            hi = lo
            if record_live:
                live["strings"].append(int(na.item()))

    def cursor_api_chunks(record_live=False):
        """gdx_cursor_extend_front_chunk_dev: chunk k of every read, the live list drops reads that have ended"""
        reset()
        a, na = None, None
        for r in range(rounds):
            eng.cursor_extend_chunk(cur_s, cur_e, q.qbuf, q.qoff, nq, chunk, r, cur_st, a, na, act[r % 2], n_act[r % 2])
            a, na = act[r % 2], n_act[r % 2]
            if record_live:
                live["chunks"].append(int(na.item()))

    def check(what):
        torch.cuda.synchronize()
        if not (torch.equal(cur_s, out["start"]) and torch.equal(cur_e, out["end"]) and not bool(cur_st.any().item())):
            raise SystemExit(f"PARITY FAILURE: the batched cursor API ({what}) and the fused search disagree on workload 5")

    if light:  # (the headline index: no pair lines, the calls run on the rank-line kernel -- one pass each is enough to say so)
        cursor_ms = timed(cursor_api_chunks, reps=1)
        check("chunks")
        res = {"name": "mixed_lengths_20_150 on the HEADLINE index (BASELINE configs[4])", "queries": nq, "op": "count (intervals)",
               "fused_value": nq / (fused_ms / 1e3), "fused_ms": fused_ms, "cursor_api_value": nq / (cursor_ms / 1e3),
               "cursor_api_ms": cursor_ms, "unit": "queries/s", "intervals_identical": True,
               "note": "exact intervals and cursor extension need the pair-line / jump structures (the 214 GB index of the "
                       "`mixed_lengths_20_150` secondary); on the 84 GB headline index (reference arrays + seed table + text units "
                       "+ full SA) both calls fall to the rank-line kernel -- the seed table serves count / locate, where no "
                       "interval has to come out"}
        log(f"[bench] secondary {res}")
        return res
    strings_ms = timed(cursor_api_strings)
    cursor_api_strings(record_live=True)
    check("strings")
    cursor_ms = timed(cursor_api_chunks)
    cursor_api_chunks(record_live=True)
    check("chunks")
    # the same API with more symbols per call: a cursor extension costs its own fixed lines (list entry, state, offsets, the
    # line of query bytes) beside one jump entry per 32 symbols, so fewer, longer calls move fewer bytes
    by_chunk = {str(chunk): cursor_ms}
    for c2 in (64, 80):
        chunk, rounds = c2, -(-w["len_max"] // c2)
        by_chunk[str(c2)] = timed(cursor_api_chunks)
        check(f"chunks of {c2}")
    chunk, rounds = 32, -(-w["len_max"] // 32)
    same = True
    res = {"name": "mixed_lengths_20_150 on the HEADLINE index (BASELINE configs[4])" if headline else
           "mixed_lengths_20_150 (BASELINE configs[4])", "queries": nq, "op": "count (intervals)",
           "fused_value": nq / (fused_ms / 1e3), "fused_ms": fused_ms,
           "cursor_api_value": nq / (cursor_ms / 1e3), "cursor_api_ms": cursor_ms, "unit": "queries/s",
           "cursor_api": f"cursor_empty + {rounds} x gdx_cursor_extend_front_chunk_dev ({chunk} symbols per call, "
                         f"device-side live lists, no host round trip inside a pass)",
           "cursor_api_ms_by_symbols_per_call": by_chunk,
           "cursor_api_strings_ms": strings_ms,
           "cursor_api_strings": "the same through gdx_cursor_extend_front_strings_dev (chunk edges computed by the "
                                 "caller, reads that have ended stay in the live list)",
           "live_cursors_after_each_call": live["chunks"], "live_cursors_after_each_call_strings": live["strings"],
           "intervals_identical": same,
           "lf_steps": lf_steps, "active_lane_fraction_fused": fetches / slots if slots else None}
    log(f"[bench] secondary {res}")
    return res
