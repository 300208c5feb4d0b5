"""benchlib.baseline -- the CPU baseline (the oracle on the host cores, compared bit for bit with the GPU output) and the hit check against the text (split out of bench.py in round 6; bench.py re-exports everything)."""
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

__all__ = ['verify_hits', 'oracle_from_index', 'cpu_baseline']

def verify_hits(torch, io_text, lengths, queries, out, hits, total_hits, nq, n_check):
    """Every checked hit (text_id, position) must spell its query in the text: independent of the oracle."""
    dev = io_text.device
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    h = torch.randint(0, total_hits, (min(n_check, total_hits),), device=dev, generator=g)
    off = out["hit_offsets"]
    q = torch.searchsorted(off, h, right=True) - 1
    qb, qe = queries.qoff[q], queries.qoff[q + 1]
    qlen = qe - qb
    toff = torch.zeros(len(lengths) + 1, dtype=torch.int64, device=dev)
    toff[1:] = torch.cumsum(torch.tensor(lengths, dtype=torch.int64, device=dev), 0)
    tid = hits[h, 0].to(torch.int64)
    pos = hits[h, 1].to(torch.int64) & 0xFFFFFFFF
    base = toff[tid] + pos
    inside = (pos + qlen) <= (toff[tid + 1] - toff[tid])
    max_len = int(qlen.max().item())
    ok = inside.clone()
    for j0 in range(0, max_len, 64):
        j = torch.arange(j0, min(j0 + 64, max_len), device=dev)
        valid = j[None, :] < qlen[:, None]
        ti = (base[:, None] + j[None, :]).clamp_(max=io_text.numel() - 1)
        qi = (qb[:, None] + j[None, :]).clamp_(max=queries.qbuf.numel() - 1)
        same = (io_text[ti] == queries.qbuf[qi]) | ~valid
        ok &= same.all(dim=1)
    return {"hits_checked": int(h.numel()), "hits_matching_text": int(ok.sum().item())}


def oracle_from_index(np, index, alpha, args, wl, n_threads, lib=None):
    """The CPU restatement's index (reference layout) from the arrays the GPU build exports."""
    from oracle import oracle as orc

    bwt = index.export_bwt()
    samples = index.export_sa_samples()
    bk, bv = index.export_borders()
    sent = index.export_sentinel_indices()
    width = {"u32": 32, "i32": -32, "i64": 64}[wl["storage"]]
    return orc.OracleIndex.from_bwt(bwt, samples, args.sa_rate, bk, bv, sent, alpha.io_to_dense_table, 6, 4,
                                    lookup_depth=args.lookup_depth, width=width, n_threads=n_threads, lib=lib)


def cpu_baseline(np, torch, index, alpha, queries, runner, do_locate, args, wl):
    """The CPU restatement of genedex's batched path (oracle/), timed on the host cores of this box on a bounded sample
    of the same queries against the same index, and compared bit for bit with the GPU results.  The thread count is
    swept (1, 8, 32, 64, 128, all usable CPUs) and the best is reported, with the sweep."""
    from oracle import oracle as orc

    avail, avail_note = host_threads()
    # (libgomp reads these when it is loaded: threads spread over the cores and stay there)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "threads")
    lib = None
    try:  # rebuild for this host's CPU; fall back to the shipped generic build
        path = orc.build_oracle(out=f"/tmp/libgdx_oracle_native_{os.getpid()}.so",
                                cflags="-O3 -march=native -fopenmp -fPIC -std=c11")
        lib = orc.load(path)
    except Exception as e:  # noqa: BLE001
        log(f"[bench] native oracle build failed ({e}); using the shipped build")
        lib = orc.load()
    t0 = time.time()
    cpu = oracle_from_index(np, index, alpha, args, wl, avail, lib)
    log(f"[bench] CPU index (reference layout, huge pages, parallel first touch) ready in {time.time() - t0:.1f}s; {avail_note}")

    def run(first, count, threads):
        qbuf, qoff = queries.host_slice(first, count)
        t0 = time.perf_counter()
        s, e = cpu.cursors_for_many(qbuf, qoff, n_threads=threads)
        t_count = time.perf_counter() - t0
        t_loc, loc = 0.0, None
        if do_locate:
            t0 = time.perf_counter()
            loc = cpu.locate_intervals(s, e, n_threads=threads)
            t_loc = time.perf_counter() - t0
        return s, e, loc, t_count, t_loc

    # thread sweep, ~1.5 s of CPU work each (sized from a one-thread calibration)
    calib = min(queries.nq, 50_000)
    _, _, _, tc, tl = run(0, calib, 1)
    rate1 = calib / max(tc + tl, 1e-6)
    sweep = {}
    for th in sorted({t for t in (1, 8, 32, 64, 128, avail) if t <= avail}):
        m = int(min(queries.nq, max(calib, rate1 * min(th, 48) * 1.5)))
        run(0, min(m, 20_000 * th), th)  # threads started, pages of the outputs touched
        _, _, _, tc, tl = run(0, m, th)
        sweep[th] = {"queries": m, "count_s": tc, "locate_s": tl, "qps": m / (tc + tl), "count_only_qps": m / tc}
        log(f"[bench] CPU baseline sweep: {th} threads -> {sweep[th]['qps']:.3e} q/s (count only {sweep[th]['count_only_qps']:.3e})")
    best = max(sweep, key=lambda t: sweep[t]["qps"])
    n_sample = int(min(queries.nq, max(calib, sweep[best]["qps"] * args.cpu_seconds)))
    s, e, loc, tc, tl = run(0, n_sample, best)
    value = n_sample / (tc + tl)
    # bit-exactness at full index size: the timed path's counts and hits (same order), and the exact intervals of
    # the interval call (cursors_for_many_queries) on the same prefix
    out = runner.outs[0]
    g_counts = runner.counts(out)[:n_sample].cpu().numpy().astype(np.uint64) & np.uint64(0xFFFFFFFF)
    same_counts = bool(np.array_equal(g_counts, e - s))
    exact = runner.eng.alloc_outputs(n_sample)
    runner.eng.search(queries.slice(0, n_sample), exact)
    torch.cuda.synchronize()
    gs = exact["start"].cpu().numpy().astype(np.uint64) & np.uint64(0xFFFFFFFF)
    ge = exact["end"].cpu().numpy().astype(np.uint64) & np.uint64(0xFFFFFFFF)
    same_intervals = bool(np.array_equal(gs, s) and np.array_equal(ge, e))
    del exact
    same_hits = None
    if do_locate:
        runner.step(0, False)  # the accounting pass rewrote the hit buffer (same values); make it the timed path's again
        torch.cuda.synchronize()
        off, t, p = loc
        n_h = int(off[-1])
        gh = runner.hits[0][:n_h].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        goff = out["hit_offsets"][:n_sample + 1].cpu().numpy().astype(np.uint64)
        same_hits = bool(np.array_equal(goff, off) and np.array_equal(gh[:, 0], t.astype(np.int64))
                         and np.array_equal(gh[:, 1], p.astype(np.int64)))
    if not same_intervals or not same_counts or same_hits is False:
        raise SystemExit(f"PARITY FAILURE vs CPU oracle: intervals {same_intervals}, counts {same_counts}, hits {same_hits}")
    # the author's "batching gives about 2x" (src/lib.rs:37-40): batched vs single-query path on ONE thread
    m1 = min(queries.nq, 100_000)
    qbuf1, qoff1 = queries.host_slice(0, m1)
    t0 = time.perf_counter()
    cpu.cursors_for_many(qbuf1, qoff1, n_threads=1)
    t_batched1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    cpu.cursors_single(qbuf1, qoff1, n_threads=1)
    t_single1 = time.perf_counter() - t0
    log(f"[bench] CPU baseline: {n_sample} queries, count {tc:.2f}s + locate {tl:.2f}s on {best} threads "
        f"-> {value:.3e} q/s; GPU results identical: intervals {same_intervals}, counts {same_counts}, hits {same_hits}")
    one = sweep[min(sweep)]["qps"]
    return {"value": value, "unit": "queries/s", "cores": best, "kind": "port",
            "sample": f"first {n_sample} queries, count {tc:.2f}s + locate {tl:.2f}s, {best} of {avail} usable threads",
            "sample_long": f"first {n_sample} queries of the GPU batch, same index (BWT + samples exported from the GPU "
                           f"build, occurrence table rebuilt in the reference layout on huge pages, lookup depth "
                           f"{args.lookup_depth}, no acceleration structures), count {tc:.2f}s + locate {tl:.2f}s",
            "usable_threads": avail, "usable_threads_note": avail_note,
            "speedup_over_one_thread": value / one if one else None,
            "thread_sweep_qps": {str(t): round(v["qps"]) for t, v in sweep.items()},
            "thread_sweep_count_only_qps": {str(t): round(v["count_only_qps"]) for t, v in sweep.items()},
            "count_only_value": n_sample / tc,
            "bit_exact_vs_gpu": {"intervals": same_intervals, "counts": same_counts, "hits": same_hits},
            "one_thread": {"batched_path_count_qps": m1 / t_batched1, "single_query_path_count_qps": m1 / t_single1,
                           "batching_speedup": t_single1 / t_batched1, "queries": m1}}
