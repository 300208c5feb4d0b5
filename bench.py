#!/usr/bin/env python3
"""bench.py -- headline benchmark: queries/sec (count+locate) on an hg38-scale synthetic DNA text.

One "step" = one pass of the query hot path over one batch of synthetic reads that already sit in HBM:
backward search of every query (lookup jump + LF loop) -> exclusive scan of the interval sizes ->
locate walk of every hit (+ the result gather to rank 0 when N > 1).  The index (3.1 G symbols incl.
24 sentinels, u32, sampling rate 4) is built on the GPU before the timed region and is not timed.

    python bench.py --gpus N --steps K --warmup W [--workload hg38|cfg2|small] [--op count+locate|count]

For N > 1 launch one rank per GPU with torch.distributed.run; the index is replicated, every rank
searches its own shard of N x nq queries (weak scaling), results are gathered to rank 0 over RCCL.
Rank 0 prints ONE JSON line.  Everything else goes to stderr.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.md section 3, workload 3/4: 24 texts proportional to hg38, 100 M reads of length 50
    "hg38": dict(total=3_100_000_000, n_texts=24, nq=100_000_000, len_min=50, len_max=50, sampled_ppm=900_000,
                 storage="u32", label="hg38-scale 3.1G DNA-N text (24 texts), 100M len-50 reads 90% sampled / 10% random"),
    # workload 5: mixed lengths, early termination
    "mixed": dict(total=3_100_000_000, n_texts=24, nq=50_000_000, len_min=20, len_max=150, sampled_ppm=700_000,
                  storage="u32", label="hg38-scale text, 50M reads of length 20..150, 70% sampled / 30% random"),
    # workload 2
    "cfg2": dict(total=1 << 28, n_texts=1, nq=10_000_000, len_min=50, len_max=50, sampled_ppm=900_000,
                 storage="i32", label="256 MB DNA-N text, 10M len-50 reads"),
    "small": dict(total=1 << 24, n_texts=3, nq=1_000_000, len_min=50, len_max=50, sampled_ppm=900_000,
                  storage="i32", label="16 MB DNA-N text, 1M len-50 reads (plumbing check)"),
}

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="hg38", choices=sorted(WORKLOADS))
    ap.add_argument("--op", default="count+locate", choices=["count+locate", "count"])
    ap.add_argument("--lookup-depth", type=int, default=0, help="reference default 0 (config.rs:76)")
    ap.add_argument("--sa-rate", type=int, default=4, help="reference default 4 (config.rs:75)")
    ap.add_argument("--nq", type=int, default=None, help="override the number of queries per GPU")
    ap.add_argument("--total", type=int, default=None, help="override the text length")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bandwidth", action="store_true")
    ap.add_argument("--no-hint", action="store_true", help="locate without the hints of the search (A/B)")
    ap.add_argument("--overlap", action="store_true",
                    help="run the locate of batch k on a second stream beside the search of batch k + 1 (measured: "
                         "20.8 instead of 21.2 ms per step, both kernels contend for DRAM requests; off by default so "
                         "that the per-kernel durations stay those of the kernels alone)")
    ap.add_argument("--verify-hits", type=int, default=1_000_000)
    ap.add_argument("--path", default="records", choices=["records", "arrays"],
                    help="records: fused count + locate over 16-byte search records (gdx_locate_many_*_dev, lazy "
                         "tails); arrays: exact intervals + hints (gdx_cursors_for_many_queries_hint_dev + "
                         "gdx_locate_intervals_hint_dev), the round-1 path")
    ap.add_argument("--secondary-depth", type=int, default=10,
                    help="N=1 only: after the headline run (reference-default lookup depth), rebuild the index with "
                         "this lookup-table depth, time the same step and check the counts are identical; 0 = skip")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: using WORLD_SIZE")

    import numpy as np
    import torch  # before libgdx.so: both must share torch's HIP runtime
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the query path has no CPU fallback")
    # Dry-run hooks for boxes with one GPU (tools/dryrun_two_ranks.sh): GDX_BENCH_ONE_GPU=1 puts every rank on device
    # 0 and GDX_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device; the N > 1 control flow
    # (sharding, size exchange, double-buffered gather) is then exercised end to end on real kernels.
    device_index = 0 if os.environ.get("GDX_BENCH_ONE_GPU") == "1" else local_rank
    backend = os.environ.get("GDX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from genedex_amd import alphabet
    from genedex_amd import dist as gdist
    from genedex_amd.device import (DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths,
                                    measure_bandwidth, synth_text)

    wl = dict(WORKLOADS[args.workload])
    if args.nq:
        wl["nq"] = args.nq
    if args.total:
        wl["total"] = args.total
    nq = wl["nq"]
    alpha = alphabet.ascii_dna_with_n()

    # ---- inputs into HBM, index build (untimed) ----------------------------------------------------
    t0 = time.time()
    io_text = synth_text(wl["total"], seed=42, n_per_million=10_000, device=dev)
    lengths = hg38_text_lengths(wl["total"], wl["n_texts"])
    torch.cuda.synchronize()
    t_text = time.time() - t0
    t0 = time.time()
    index = build_index_from_device_text(io_text, lengths, alpha, sa_rate=args.sa_rate,
                                         lookup_depth=args.lookup_depth, index_storage=wl["storage"])
    t_build = time.time() - t0
    stats = index.build_stats()
    log(f"[bench r{rank}] text {t_text:.1f}s, index build {t_build:.1f}s {stats}, "
        f"index {index.info.device_bytes / 1e9:.2f} GB in HBM, n = {index.total_text_len()}")
    t0 = time.time()
    queries = DeviceQueries.synth(io_text, lengths, nq, wl["len_min"], wl["len_max"], wl["sampled_ppm"],
                                  seed=43 + 1000 * rank)
    log(f"[bench r{rank}] {nq} queries ({queries.total_bytes / 1e9:.2f} GB) generated in {time.time() - t0:.1f}s")

    eng = DeviceEngine(index)
    do_locate = args.op == "count+locate"
    use_rec = args.path == "records" and do_locate
    out = eng.alloc_outputs(nq, hint=do_locate and not args.no_hint)
    if use_rec:
        out["rec"] = eng.alloc_records(nq)

    def run_search(o):
        if use_rec:
            eng.locate_search(queries, o["rec"])
        else:
            eng.search(queries, o)

    def run_offsets(o):
        if use_rec:
            eng.locate_offsets(o["rec"], nq, o["hit_offsets"])
        else:
            eng.hit_offsets(o, nq)

    def run_locate(o, h, ws):
        if use_rec:
            eng.locate_hits(o["rec"], nq, o["hit_offsets"], total_hits, h, ws)
        else:
            eng.locate(o, nq, total_hits, h, ws)

    # sizing pass (also the first warm-up of the kernels)
    run_search(out)
    run_offsets(out)
    torch.cuda.synchronize()
    total_hits = int(out["hit_offsets"][nq].item())
    if use_rec:
        cnt32 = torch.empty(nq, dtype=torch.int32, device=dev)
        eng.unpack_records(out["rec"], nq, cnt32, out["status"])
        out["start"].zero_()
        out["end"].copy_(cnt32)  # end - start = count for the checks below (the exact intervals are not computed)
        del cnt32
    n_status = int((out["status"] != 0).sum().item())
    hits = torch.empty((max(total_hits, 1), 2), dtype=torch.int32, device=dev)
    workspace = torch.empty(max(eng.locate_workspace_bytes(total_hits), 16), dtype=torch.uint8, device=dev)
    # Per-query counts travel to rank 0 in the narrowest integer type that holds the largest count of any rank
    # (known from the sizing pass; lossless): 1 instead of 4 bytes per query on this workload, which keeps the
    # gather of a batch (counts + 8 bytes per hit, over one xGMI link per peer) shorter than the step it hides behind.
    max_count = int((out["end"] - out["start"]).max().item()) if nq else 0
    if world > 1:
        max_count = gdist.max_int_over_ranks(max_count, dev)
    count_dtype = torch.uint8 if max_count <= 0xff else (torch.int16 if max_count <= 0x7fff else torch.int32)
    counts = torch.empty(nq, dtype=count_dtype, device=dev)
    log(f"[bench r{rank}] {total_hits} hits, {n_status} queries with non-zero status")

    ev_search, ev_locate = [], []

    # N > 1: results are gathered to rank 0 over RCCL asynchronously, double-buffered, so that the gather of
    # batch k overlaps the kernels of batch k+1 (payloads padded to the largest shard up front).
    # Result slots: one, or two when the gather of batch k (N > 1) or its locate (--overlap) runs beside the search
    # of batch k + 1.
    overlap = do_locate and args.overlap
    slots = [(out, hits, counts, workspace)]
    gather = None
    if world > 1:
        max_hits = gdist.max_int_over_ranks(total_hits, dev)
        hits = torch.zeros((max(max_hits, 1), 2), dtype=torch.int32, device=dev)
        slots = [(out, hits, counts, workspace)]
    if world > 1 or overlap:
        o2 = eng.alloc_outputs(nq, hint="hint" in out)
        if use_rec:
            o2["rec"] = eng.alloc_records(nq)
        slots.append((o2, torch.zeros_like(hits), torch.empty_like(counts), torch.empty_like(workspace)))
    if world > 1:
        gather = gdist.PipelinedGather([[c, h] if do_locate else [c] for (_, h, c, _w) in slots], dst=0)
    main_stream = torch.cuda.current_stream()
    side_stream = torch.cuda.Stream() if overlap else main_stream
    slot_free = [None] * len(slots)  # event: the side stream is done with the slot's buffers
    step_no = [0]

    def step(record):
        slot = step_no[0] % len(slots)
        step_no[0] += 1
        o, h, cnt, ws = slots[slot]
        if gather:
            gather.acquire(slot)
        if slot_free[slot] is not None:
            main_stream.wait_event(slot_free[slot])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run_search(o)
        b.record()
        if record:
            ev_search.append((a, b))
        with torch.cuda.stream(side_stream):
            if overlap:
                side_stream.wait_event(b)
            if do_locate:
                run_offsets(o)
                c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c.record()
                run_locate(o, h, ws)
                d.record()
                if record:
                    ev_locate.append((c, d))
            if gather:
                if use_rec:
                    cnt.copy_(torch.sub(o["rec"][:nq, 1], o["rec"][:nq, 0]))
                else:
                    cnt.copy_(torch.sub(o["end"], o["start"]))  # copy_ narrows to the gather's count type
                gather.submit(slot)
            if overlap:
                slot_free[slot] = torch.cuda.Event()
                slot_free[slot].record()

    for _ in range(args.warmup):
        step(False)
    if gather:
        gather.drain()
    gdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    if gather:
        gather.drain()
    torch.cuda.synchronize()
    gdist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = gdist.max_over_ranks(elapsed, dev)
    ms_per_step = elapsed / args.steps * 1e3
    value = nq * world / (ms_per_step / 1e3)

    search_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_search]))
    locate_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_locate])) if ev_locate else None

    # ---- algorithmic bytes, counted by an extra, untimed pass ------------------
    lf_steps, fetches, fetch_slots = eng.search_step_stats(queries)
    search_bytes = queries.total_bytes + (8 * nq if args.lookup_depth > 0 else 0) + 60 * lf_steps + 8 * nq
    variant = os.environ.get("GDX_SEARCH_VARIANT", "pair")
    uniform = wl["len_max"] - wl["len_min"] <= wl["len_min"] // 4
    lanes = os.environ.get("GDX_SEARCH_LANES", "4")
    aux = eng.aux_info()
    policy = os.environ.get("GDX_LOAD_POLICY", "0")
    kernel_name = {"pair": f"search_pair_kernel{lanes}<{policy}, {aux['jump_entry_bytes'] or 8}>",
                   "quad": "search_kernel<QuadLineTable,4>", "lane": "search_kernel<LineTable,1>"}[variant]
    roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": search_bytes / (search_ms / 1e3) / 1e9,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": search_bytes / (search_ms / 1e3) / 1e9 / HBM_PEAK_GBPS,
                "traffic": None, "algorithmic_bytes_per_launch": search_bytes, "lf_steps_per_launch": lf_steps,
                "avg_launch_ms": search_ms,
                "note": "achieved = algorithmic bytes of the reference's algorithm (60 B per LF step it would execute, "
                        "BASELINE.md section 4) / kernel time; frac > 1 means the jump / top tables deliver those LF "
                        "steps with fewer bytes than the reference layout needs, not that HBM exceeds its peak: see "
                        "traffic (measured DRAM bytes per launch) and random_request_model for the bound of the kernel "
                        "as built",
                "line_fetches_per_query": fetches / nq if fetches else None,
                "active_lane_fraction": fetches / fetch_slots if fetch_slots else None}
    roofline.update(pmc_traffic(kernel_name, args, wl, nq))
    locate_roofline = None
    if do_locate and total_hits:
        acct = out
        if use_rec:  # the accounting pass counts the reference's walk steps from the exact intervals, without hints
            acct = eng.alloc_outputs(nq, hint=False)
            eng.search(queries, acct)
            eng.hit_offsets(acct, nq)
        walk_steps = eng.locate_walk_steps(acct, nq, total_hits, hits, workspace)
        if use_rec:
            del acct
        locate_bytes = 30 * walk_steps + 4 * total_hits + 8 * total_hits
        locate_roofline = {"bound": "hbm", "kernel": "locate_queue_kernel<LineTable> (+ slot -> query map)",
                           "achieved": locate_bytes / (locate_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": locate_bytes / (locate_ms / 1e3) / 1e9 / HBM_PEAK_GBPS,
                           "algorithmic_bytes_per_launch": locate_bytes, "walk_steps_per_launch": walk_steps,
                           "hits_per_launch": total_hits, "avg_launch_ms": locate_ms}

    # ---- size-independent parity properties at full size ---------------------------------------------------
    parity = {"queries_with_status": n_status}
    found = int(((out["end"] - out["start"]) > 0).sum().item())
    parity["queries_found"] = found
    parity["found_fraction"] = found / nq
    if do_locate and total_hits and args.verify_hits:
        parity.update(verify_hits(torch, io_text, lengths, queries, out, hits, total_hits, nq, args.verify_hits))
        if parity["hits_checked"] != parity["hits_matching_text"]:
            raise SystemExit(f"PARITY FAILURE: {parity}")

    result = {
        "metric": "queries/sec (count+locate), hg38-scale text, 100M len-50 reads" if do_locate
        else "queries/sec (count), hg38-scale text, 100M len-50 reads",
        "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": wl["label"], "name": args.workload, "op": args.op, "queries_per_gpu": nq,
                   "text_len": wl["total"], "n_texts": wl["n_texts"], "lookup_depth": args.lookup_depth,
                   "sa_rate": args.sa_rate, "index_storage": wl["storage"], "hits_per_gpu": total_hits,
                   "aux_structures": eng.aux_info(),
                   "parallelism": f"index replicated x{world}, queries sharded, gather to rank 0",
                   "gathered_bytes_per_rank_and_step": (nq * counts.element_size() + (hits.numel() * 4 if do_locate else 0))
                   if world > 1 else 0},
        "roofline": roofline,
        "locate_roofline": locate_roofline,
        "parity": parity,
        "index_build_seconds": t_build,
        "index_bytes": int(index.info.device_bytes),
    }

    if rank == 0 and not args.no_bandwidth:
        result["measured_bandwidth"] = measure_bandwidth(dev)
        result["roofline"]["frac_of_measured_stream"] = (result["roofline"]["achieved"]
                                                          / result["measured_bandwidth"]["stream_copy_GBps"])
        log(f"[bench] measured bandwidth: {result['measured_bandwidth']}")
        # The kernel's own bound: it is made of dependent random 128-byte requests, whose measured ceiling on this
        # GPU (gather128_group of measure_bandwidth) is well below the streaming peak.
        rq = result["roofline"].get("dram_read_requests_per_query")
        if rq:
            rate = rq * nq / (search_ms / 1e3) / 1e9
            ceiling = result["measured_bandwidth"]["gather128_group_Glines_per_s"]
            result["roofline"]["random_request_model"] = {
                "dram_requests_per_launch": rq * nq, "achieved_Greq_per_s": rate, "measured_ceiling_Greq_per_s": ceiling,
                "frac_of_ceiling": rate / ceiling, "traffic_frac_of_hbm_peak": result["roofline"]["traffic"]
                / (search_ms / 1e3) / 1e9 / HBM_PEAK_GBPS}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(np, index, alpha, queries, out, hits, do_locate, args, wl)
    else:
        result["cpu_baseline"] = None

    if rank == 0 and world == 1 and args.secondary_depth > 0 and wl["len_min"] >= 16:
        # Secondary design points, never `value`: (1) the reference's lookup-table knob at the depth BASELINE.md names;
        # (2) pair lines but no jump / top table; (3) every acceleration structure of this build switched off (rank
        # lines with the reference's information content only).  The HBM each rung spends is in `index_bytes`.
        base_counts = (out["end"] - out["start"]).clone()
        del eng, index
        torch.cuda.empty_cache()
        common = (torch, io_text, lengths, alpha, queries, out, hits, workspace, base_counts, nq, total_hits,
                  do_locate, args, wl)
        result["secondary"] = [
            secondary_run(f"lookup_depth_{args.secondary_depth}", args.secondary_depth, {}, *common),
            secondary_run("pair_lines_only", args.lookup_depth, {"GDX_TOP_DEPTH": "0", "GDX_NO_JUMP_TABLE": "1"}, *common),
            secondary_run("no_acceleration_structures", args.lookup_depth,
                          {"GDX_TOP_DEPTH": "0", "GDX_NO_JUMP_TABLE": "1", "GDX_NO_PAIR_LINES": "1"}, *common),
        ]

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def secondary_run(name, lookup_depth, env, torch, io_text, lengths, alpha, queries, out, hits, workspace, base_counts,
                  nq, total_hits, do_locate, args, wl):
    """Rebuild the index with another configuration, time the same step and require identical interval sizes.
    `lookup_depth` is the reference's knob (config.rs:36-47; README.md:101-115 recommends a deeper table for large
    texts); `env` switches build-time structures of this implementation (fm_index.hip)."""
    from genedex_amd.device import DeviceEngine, build_index_from_device_text

    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        t0 = time.time()
        index = build_index_from_device_text(io_text, lengths, alpha, sa_rate=args.sa_rate, lookup_depth=lookup_depth,
                                             index_storage=wl["storage"])
        t_build = time.time() - t0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    eng = DeviceEngine(index)
    ev_s, ev_l = [], []

    def step(record):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.search(queries, out)
        b.record()
        if do_locate:
            eng.hit_offsets(out, nq)
            c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c.record()
            eng.locate(out, nq, total_hits, hits, workspace)
            d.record()
            if record:
                ev_l.append((c, d))
        if record:
            ev_s.append((a, b))

    step(False)
    torch.cuda.synchronize()
    steps = min(args.steps, 3)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    same = bool(torch.equal(out["end"] - out["start"], base_counts))
    if not same:
        raise SystemExit(f"PARITY FAILURE: secondary configuration {name} changed interval sizes")
    res = {"name": name, "lookup_depth": lookup_depth, "aux_structures": eng.aux_info(), "value": nq / (ms / 1e3),
           "unit": "queries/s", "ms_per_step": ms, "search_ms": sum(a.elapsed_time(b) for a, b in ev_s) / len(ev_s),
           "locate_ms": sum(a.elapsed_time(b) for a, b in ev_l) / len(ev_l) if ev_l else None,
           "counts_identical_to_headline": same, "index_build_seconds": t_build,
           "index_bytes": int(index.info.device_bytes)}
    log(f"[bench] secondary {name}: {res}")
    del eng, index
    torch.cuda.empty_cache()
    return res


def pmc_traffic(kernel_name, args, wl, nq):
    """HBM traffic of the search kernel per launch from the committed rocprofv3 PMC summary of this very
    configuration (FETCH_SIZE cannot be read from inside the process).  Correction per
    MI355X_MICROARCH.md section HBM and tools/calibrate_fetch_size.sh: every DRAM request of this GPU is
    128 bytes (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ also for 64-byte gathers) and FETCH_SIZE tallies 64 bytes
    per request, so read bytes = 2 * FETCH_SIZE[KB] * 1024.  The PMC pass runs fewer queries; traffic per
    query is scaled to this launch."""
    path = os.path.join(ROOT, "profiles", "r01", "search_pmc_final.json")
    try:
        with open(path) as f:
            pmc = json.load(f)
        if (pmc["workload"], pmc["lookup_depth"], pmc["kernel"]) != (args.workload, args.lookup_depth, kernel_name):
            return {"traffic": None}
        per_query = (2 * pmc["FETCH_SIZE_KB_per_launch"] + pmc["WRITE_SIZE_KB_per_launch"]) * 1024 / pmc["queries_per_launch"]
        res = {"traffic": per_query * nq, "traffic_source": "profiles/r01/search_pmc_final.json (rocprofv3 --pmc FETCH_SIZE / "
               "WRITE_SIZE, separate passes, FETCH_SIZE doubled: all requests are 128 B)"}
        if "TCC_EA0_RDREQ_per_launch" in pmc:
            res["dram_read_requests_per_query"] = pmc["TCC_EA0_RDREQ_per_launch"] / pmc["queries_per_launch"]
        return res
    except (OSError, KeyError, ValueError):
        return {"traffic": None}


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


def cpu_baseline(np, index, alpha, queries, out, hits, do_locate, args, wl):
    """The CPU restatement of genedex's batched path (oracle/), timed on all host cores on a bounded sample
    of the same queries against the same index, and compared bit for bit with the GPU results."""
    from oracle import oracle as orc

    cores = os.cpu_count() or 1
    lib = None
    try:  # rebuild for this host's CPU; fall back to the shipped generic build
        path = orc.build_oracle(out=f"/tmp/libgdx_oracle_native_{os.getpid()}.so",
                                cflags="-O3 -march=native -fopenmp -fPIC -std=c11")
        lib = orc.load(path)
    except Exception as e:  # noqa: BLE001
        log(f"[bench] native oracle build failed ({e}); using the shipped build")
        lib = orc.load()
    t0 = time.time()
    bwt = index.export_bwt()
    samples = index.export_sa_samples()
    bk, bv = index.export_borders()
    sent = index.export_sentinel_indices()
    width = {"u32": 32, "i32": -32, "i64": 64}[wl["storage"]]
    cpu = orc.OracleIndex.from_bwt(bwt, samples, args.sa_rate, bk, bv, sent, alpha.io_to_dense_table, 6, 4,
                                   lookup_depth=args.lookup_depth, width=width, n_threads=cores, lib=lib)
    del bwt, samples
    log(f"[bench] CPU index (reference layout) ready in {time.time() - t0:.1f}s on {cores} threads")

    def run(first, count):
        qbuf, qoff = queries.host_slice(first, count)
        t0 = time.perf_counter()
        s, e = cpu.cursors_for_many(qbuf, qoff, n_threads=cores)
        t_count = time.perf_counter() - t0
        t_loc, loc = 0.0, None
        if do_locate:
            t0 = time.perf_counter()
            loc = cpu.locate_intervals(s, e, n_threads=cores)
            t_loc = time.perf_counter() - t0
        return s, e, loc, t_count, t_loc

    calib = min(queries.nq, 200_000)
    _, _, _, tc, tl = run(0, calib)
    rate = calib / max(tc + tl, 1e-6)
    n_sample = int(min(queries.nq, max(calib, rate * args.cpu_seconds)))
    s, e, loc, tc, tl = run(0, n_sample)
    value = n_sample / (tc + tl)
    # bit-exactness at full index size: same intervals, same hits in the same order
    gs = out["start"][:n_sample].cpu().numpy().astype(np.uint64) & np.uint64(0xFFFFFFFF)
    ge = out["end"][:n_sample].cpu().numpy().astype(np.uint64) & np.uint64(0xFFFFFFFF)
    same_intervals = bool(np.array_equal(gs, s) and np.array_equal(ge, e))
    same_hits = None
    if do_locate:
        off, t, p = loc
        n_h = int(off[-1])
        gh = hits[:n_h].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        goff = out["hit_offsets"][:n_sample + 1].cpu().numpy().astype(np.uint64)
        same_hits = bool(np.array_equal(goff, off) and np.array_equal(gh[:, 0], t.astype(np.int64))
                         and np.array_equal(gh[:, 1], p.astype(np.int64)))
    if not same_intervals or same_hits is False:
        raise SystemExit(f"PARITY FAILURE vs CPU oracle: intervals {same_intervals}, hits {same_hits}")
    # the author's "batching gives about 2x" (src/lib.rs:37-40): batched vs single-query path on ONE thread
    m1 = min(queries.nq, 100_000)
    qbuf1, qoff1 = queries.host_slice(0, m1)
    t0 = time.perf_counter()
    cpu.cursors_for_many(qbuf1, qoff1, n_threads=1)
    t_batched1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    cpu.cursors_single(qbuf1, qoff1, n_threads=1)
    t_single1 = time.perf_counter() - t0
    log(f"[bench] CPU baseline: {n_sample} queries, count {tc:.2f}s + locate {tl:.2f}s on {cores} threads "
        f"-> {value:.3e} q/s; GPU results identical: intervals {same_intervals}, hits {same_hits}")
    return {"value": value, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"first {n_sample} queries of the GPU batch, same index (BWT + samples exported from the GPU "
                      f"build, occurrence table rebuilt in the reference layout), count {tc:.2f}s + locate {tl:.2f}s",
            "count_only_value": n_sample / tc, "bit_exact_vs_gpu": {"intervals": same_intervals, "hits": same_hits},
            "one_thread": {"batched_path_count_qps": m1 / t_batched1, "single_query_path_count_qps": m1 / t_single1,
                           "batching_speedup": t_single1 / t_batched1, "queries": m1}}


if __name__ == "__main__":
    main()
