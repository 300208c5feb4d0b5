#!/usr/bin/env python3
"""bench.py -- headline benchmark: queries/sec (count+locate) on an hg38-scale synthetic DNA text.

One "step" = one pass of the query hot path over one batch of synthetic reads that already sit in HBM:
backward search of every query -> exclusive scan of the counts -> locate of every hit (+ the result gather to
rank 0 when N > 1).  The index (3.1 G symbols incl. 24 sentinels, u32, sampling rate 4 = BASELINE.json configs[2])
is built on the GPU before the timed region and is not timed.

    python bench.py --gpus N --steps K --warmup W [--workload hg38|mixed|cfg2|small] [--op count+locate|count]

For N > 1 launch one rank per GPU with torch.distributed.run; the index is replicated and results are gathered to rank 0
over RCCL.  The run first gives every rank its own batch of nq queries (`weak_scaling`), then shards ONE batch of nq
queries over the ranks -- BASELINE.json configs[3], which is what `value`, `ms_per_step` and `config.workload` report at
N > 1 (`scaling: "strong"`).  Rank 0 prints ONE JSON line of less than 4 KB (the contract's keys, `roofline`,
`cpu_baseline`); every other measurement (ladder, secondaries, end to end, bandwidths) goes to --side-file
(gpurun_out/bench_secondary.json) and to stderr.

Roofline (N = 1): `roofline.traffic` = HBM bytes of the dominant kernel per launch, measured by rocprofv3 PMC passes
of this very workload that bench.py itself starts as child processes BEFORE it touches the GPU (FETCH_SIZE and
WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); `roofline.frac` =
traffic / live HIP-event duration / 8 TB/s, at most 1 by construction; `roofline.frac_rocprof` = the same traffic over
rocprofv3's own average duration of the kernels in a `--kernel-trace --stats` child pass of the same workload (its
kernel_stats.csv is kept beside the side file: the summary under profiles/ is that file).  The ratio of the reference algorithm's
logical bytes to the time is reported separately as `algorithmic_ratio`.
"""
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

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the parts (benchlib/): common (workloads, StepRunner), pmc (rocprofv3 child passes), line (the stdout line), baseline (CPU
# baseline, hit check), multi (N > 1), end_to_end (host-pointer calls, FASTQ), secondaries -- all re-exported here
from benchlib.common import *  # noqa: E402,F401,F403
from benchlib.pmc import *  # noqa: E402,F401,F403
from benchlib.line import *  # noqa: E402,F401,F403
from benchlib.baseline import *  # noqa: E402,F401,F403
from benchlib.multi import *  # noqa: E402,F401,F403
from benchlib.end_to_end import *  # noqa: E402,F401,F403
from benchlib.secondaries import *  # noqa: E402,F401,F403


def parse_args():
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
    ap.add_argument("--no-hint", action="store_true", help="arrays path: locate without the hints of the search (A/B)")
    ap.add_argument("--path", default="records", choices=["records", "records16", "arrays"],
                    help="records: fused count + locate over search records with the compact results beside them "
                         "(gdx_locate_many_*_compact_dev: 4 bytes per read the seed kernel answers, 16-byte records for the "
                         "rest); records16: 16-byte records only (gdx_locate_many_*_dev, rounds 2..3a); arrays: exact "
                         "intervals + hints (gdx_cursors_for_many_queries_hint_dev + gdx_locate_intervals_hint_dev), the "
                         "round-1 path")
    ap.add_argument("--overlap", action="store_true",
                    help="run the locate of batch k on a second stream beside the search of batch k + 1 (measured in "
                         "round 1: 2 %%, both kernels contend for DRAM requests; off by default so that the per-kernel "
                         "durations stay those of the kernels alone)")
    ap.add_argument("--verify-hits", type=int, default=1_000_000)
    ap.add_argument("--secondary-depth", type=int, default=10,
                    help="N=1 only: lookup-table depth of the `lookup_depth_D` secondary design point; 0 = no secondaries")
    ap.add_argument("--index", default="default", choices=["default", "seed", "tables"],
                    help="headline index: default = what gdx_index_build makes with every build option left at its default -- "
                         "the DEFAULT SHAPE: the reference's arrays + seed table + text units + full and inverse suffix array + "
                         "pair lines + depth-14 top table (104 GB at hg38 scale); count / locate, exact intervals and cursors "
                         "are all measured on this ONE index.  seed = rounds 3b-5's lean headline index (seed table + text "
                         "units + full suffix array, 84 GB: count / locate only); tables = pair lines + 32-byte jump entries "
                         "+ depth-16 top table (144 GB: the headline of rounds 1..3a).  Both are ladder rungs now")
    ap.add_argument("--jump-bytes", type=int, default=None, help="gdx_build_options_t.jump_entry_bytes")
    ap.add_argument("--top-depth", type=int, default=None, help="gdx_build_options_t.top_table_depth")
    ap.add_argument("--no-pair-lines", action="store_true", help="gdx_build_options_t.pair_lines = 0")
    ap.add_argument("--full-sa", action="store_true", help="gdx_build_options_t.full_suffix_array = 1")
    ap.add_argument("--text-units", action="store_true", help="gdx_build_options_t.text_units = 1")
    ap.add_argument("--lanes", type=int, default=None, help="gdx_query_options_t.search_lanes")
    ap.add_argument("--load-policy", type=int, default=None, help="gdx_query_options_t.load_policy")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="skip the rocprofv3 PMC child passes (roofline.traffic then falls back to the committed "
                         "summary under profiles/ and says so)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling measurement")
    ap.add_argument("--root-weight", type=float, default=None,
                    help="N > 1, the sharded batch: rank 0's shard as a fraction of every other rank's (rank 0 also splits the "
                         "shards it receives).  Default: dist.root_weight_for from a probe of the links; 1 = equal shards")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the cfg 5, ladder and genome-like secondaries")
    ap.add_argument("--input", default="ascii", choices=["ascii", "uniform", "packed", "packed+uniform"],
                    help="how the batch lies in HBM when the timed region starts (gdx_query_layout_t): ascii (default) = IO "
                         "symbols + u64 offsets -- the reference's own input, byte slices (lib.rs:155,179), so that the alphabet "
                         "translation (alphabet.rs:195-204, SURVEY 8 row a14) happens INSIDE the timed region; uniform = the same "
                         "bytes declared uniform (every read len symbols, no offsets read); packed = 2-bit codes + offsets; "
                         "packed+uniform = 2-bit codes, no offsets (a batch translated beforehand).  uniform forms need "
                         "len_min == len_max.  With ascii the same step on the pre-translated form is measured beside it "
                         "(`packed_input`), with any other form the ascii step (`ascii_input`)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child-steps", type=int, default=2, help=argparse.SUPPRESS)
    ap.add_argument("--side-file", default=os.environ.get("GDX_BENCH_SIDE_FILE", os.path.join("gpurun_out", "bench_secondary.json")),
                    help="where everything beside the contract line goes (secondaries, ladder, end-to-end, bandwidths, notes): "
                         "the one stdout line stays below 4 KB")
    args = ap.parse_args()
    explicit = (args.jump_bytes is not None or args.top_depth is not None or args.no_pair_lines or args.full_sa or args.text_units)
    if explicit:  # hand-picked structures (ladder rungs of the PMC children, experiments)
        args.index = "tables"
    return args


# ======================================================================================================
# live PMC: rocprofv3 child processes of this same script (--pmc-child), one counter group per pass


# ======================================================================================================


def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child(args)

    # (the pool's host driver shares device memory between processes through dmabuf only: RCCL needs this, and inherits it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: using WORLD_SIZE")

    # PMC passes first: they are separate processes that each need the GPU's memory for their own index, and starting
    # them before this process initialises the GPU keeps every exec clear of a process that holds the device
    pmc, pmc_note, pmc_ref, pmc_text, pmc_lookup = None, "live PMC passes run at N = 1 only", None, None, {}
    profiled = any(k in os.environ for k in ("ROCPROFILER_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH")) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "")
    if profiled:  # under rocprofv3 already (its preload initialised the GPU): no nested profiler children
        pmc_note = "bench.py itself runs under a profiler"
    elif world == 1 and not args.no_live_pmc:
        pmc, pmc_note = run_live_pmc(args, kernel_trace=os.path.join(os.path.dirname(os.path.abspath(args.side_file)),
                                                                     "bench_kernel_stats.csv"))
        if pmc is None:
            log(f"[bench] live PMC unavailable: {pmc_note}")
        elif args.secondary_depth > 0 and not args.no_extras and not args.no_pair_lines:
            pmc_ref, _ = run_live_pmc(args, reference_layout=True)
            pmc_text, _ = run_live_pmc(args, rung="tables" if args.index != "tables" else "top16_sa_text")
            # the reference's arrays WITH its lookup tables (depth 10: BASELINE.md cfg 3's secondary; 13: the deepest that
            # SURVEY 8 sizes): DRAM requests of a fifth of the batch -- one pass each, the counters scale with the reads
            if not args.no_extras:
                pmc_lookup["queries"] = min(LOOKUP_PMC_READS, workload_of(args)["nq"])
                for d in LOOKUP_RUNGS:
                    pmc_lookup[d], _ = run_live_pmc(args, reference_layout=True, lookup_depth=d, only=("requests",),
                                                    nq=pmc_lookup["queries"])

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

    wl = workload_of(args)
    nq = wl["nq"]
    alpha = alphabet.ascii_dna_with_n()
    do_locate = args.op == "count+locate"

    # ---- inputs into HBM, index build (untimed) ----------------------------------------------------
    t0 = time.time()
    io_text = synth_text(wl["total"], seed=42, n_per_million=10_000, device=dev)
    lengths = hg38_text_lengths(wl["total"], wl["n_texts"])
    torch.cuda.synchronize()
    t_text = time.time() - t0
    t0 = time.time()
    index = build_index_from_device_text(io_text, lengths, alpha, sa_rate=args.sa_rate,
                                         lookup_depth=args.lookup_depth, index_storage=wl["storage"],
                                         options=build_options_of(args))
    apply_query_options(index, args)
    t_build = time.time() - t0
    stats = index.build_stats()
    log(f"[bench r{rank}] text {t_text:.1f}s, index build {t_build:.1f}s {stats}, "
        f"index {index.info.device_bytes / 1e9:.2f} GB in HBM, n = {index.total_text_len()}")
    t0 = time.time()
    # weak scaling: every rank has its own batch (rank 0's is the N = 1 batch)
    queries = DeviceQueries.synth(io_text, lengths, nq, wl["len_min"], wl["len_max"], wl["sampled_ppm"],
                                  seed=43 + 1000 * rank)
    log(f"[bench r{rank}] {nq} queries ({queries.total_bytes / 1e9:.2f} GB) generated in {time.time() - t0:.1f}s")

    eng = DeviceEngine(index)
    aux = eng.aux_info()
    if args.index == "default" and wl["total"] >= 1 << 20 and not aux["default_shape"]:
        # the headline runs on what a caller of gdx_index_build gets: nothing was asked for, and the library must have chosen
        raise SystemExit(f"[bench] the library did not build its default shape (no room in HBM?): {aux}")
    # N > 1, before anything is timed: ONE small sharded, gathered step whose concatenated shards must equal rank 0's own output of
    # the same reads (strong_scaling raises SystemExit otherwise: the run ends non-zero, without a number), the backend's rank
    # count and the rate a probe gather measures on this run's links.  The RCCL path has never met several GPUs before the
    # driver's run: if it is wrong there, it says so here instead of printing a throughput.
    preflight = None
    if world > 1:
        pre_args = argparse.Namespace(**{**vars(args), "steps": 1, "warmup": 0})
        n_pre = max(min(nq, 1 << 18), 8 * world)
        pre = strong_scaling(torch, gdist, eng, io_text, lengths, wl, n_pre, do_locate, pre_args, rank, world, dev)
        preflight = {"queries": n_pre, "ranks": dist.get_world_size(), "backend": dist.get_backend(),
                     "gather_link_GBps": pre.get("gather_probe_GBps_per_link"), "gather_wire": pre.get("gather_wire"),
                     "shards_equal_single_rank_output": pre.get("shards_equal_single_rank_output")}
        log(f"[bench r{rank}] preflight: {preflight}")
        torch.cuda.empty_cache()
    n_slots = 2 if (world > 1 or (do_locate and args.overlap)) else 1
    run_queries = input_form(queries, index, args, wl)
    runner = StepRunner(torch, eng, run_queries, nq, do_locate, args.path, hint=not args.no_hint, n_slots=n_slots)
    total_hits = runner.size()
    out = runner.outs[0]
    n_status = int((runner.status(out) != 0).sum().item())
    # (queries whose compact result says "see the record" and their hits: what travels beside the 4 bytes per query at N > 1)
    exceptions = dict(zip(("queries", "hits"), gdist.exception_sizes(out["compact"], out["hit_offsets"], nq))) \
        if runner.use_compact else None
    log(f"[bench r{rank}] {total_hits} hits, {n_status} queries with non-zero status")

    # N > 1: results are gathered to rank 0 over RCCL asynchronously, double-buffered, so that the gather of batch k
    # overlaps the kernels of batch k+1 (payloads padded to the largest shard up front).  Per-query counts travel in
    # the narrowest integer type that holds the largest count of any rank (known from the sizing pass; lossless).
    gather, count_of, gathered_bytes = None, None, 0
    if world > 1:
        gather, count_of, gathered_bytes = make_gather(torch, gdist, runner, dev, do_locate)
    elapsed, _ = timed_steps(torch, gdist, runner, args.steps, args.warmup, dev, gather, count_of,
                             overlap=do_locate and args.overlap)
    runner.check_totals()
    narrow_offsets = runner.widen_offsets()
    ms_per_step = elapsed / args.steps * 1e3
    value = nq * world / (ms_per_step / 1e3)
    search_ms = runner.mean_ms(runner.ev_search)
    locate_ms = runner.mean_ms(runner.ev_locate)
    hits = runner.hits[0]

    # ---- the same step on the other form of the batch, beside the headline: `value` is timed on the reference's own input (IO
    # symbols + u64 offsets: the translation of SURVEY row a14 inside the timed region) and the batch translated beforehand
    # (2-bit codes, uniform length when every read has one) is `packed_input`; with --input <another form> the ascii step is
    # `ascii_input` --------------
    ascii_input = packed_input = None
    if world == 1:
        other_form = ("packed+uniform" if wl["len_min"] == wl["len_max"] else "packed") if args.input == "ascii" else "ascii"
        other_q = queries if other_form == "ascii" else input_form(queries, index, argparse.Namespace(input=other_form, workload=args.workload), wl)
        r2 = StepRunner(torch, eng, other_q, nq, do_locate, args.path, hint=not args.no_hint)
        if r2.size() != total_hits:
            raise SystemExit(f"PARITY FAILURE: the {other_form} form of the batch gives another number of hits")
        e2, _ = timed_steps(torch, gdist, r2, args.steps, args.warmup, dev)
        r2.check_totals()
        r2.widen_offsets()
        same = (bool(torch.equal(r2.outs[0]["hit_offsets"], out["hit_offsets"])) and
                bool(torch.equal(r2.hits[0][:total_hits], runner.hits[0][:total_hits]))) if do_locate else \
            bool(torch.equal(r2.counts(r2.outs[0]), runner.counts(out)))
        if not same:
            raise SystemExit(f"PARITY FAILURE: the {other_form} form of the batch gives other offsets or hits")
        other = {"value": nq / (e2 / args.steps), "unit": "queries/s", "ms_per_step": e2 / args.steps * 1e3,
                 "search_ms": r2.mean_ms(r2.ev_search), "locate_ms": r2.mean_ms(r2.ev_locate), "input": other_form,
                 "query_bytes": other_q.total_bytes + (0 if other_q.uniform_len else 8 * (nq + 1)),
                 "offsets_and_hits_identical_to_headline": same}
        if other_form == "ascii":
            ascii_input = other
        else:
            packed_input = other
            packed_input["note"] = "the batch translated to 2-bit codes BEFORE the timed region (row a14 outside it): not `value`"
        log(f"[bench] the same step on {other_form} input: {other}")
        del r2, other_q
        torch.cuda.empty_cache()

    # ---- the step on what a rank of 2 / 4 / 8 GPUs gets of this batch (N = 1; BASELINE configs[3] shards ONE batch) ----------
    shard_step = None
    if world == 1 and do_locate and not args.no_extras and nq >= 8_000_000:
        shard_step = {}
        for parts in (2, 4, 8):
            n_part = nq // parts
            part = input_form(queries.copy_slice(0, n_part), index, args, wl)
            r3 = StepRunner(torch, eng, part, n_part, do_locate, args.path, hint=not args.no_hint)
            r3.size()
            for _ in range(3):
                r3.step(0, False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                r3.step(0, False)  # (no events: an event record between two kernels costs the step 5 us)
            torch.cuda.synchronize()
            ms3 = (time.perf_counter() - t0) / args.steps * 1e3
            r3.check_totals()
            shard_step[str(n_part)] = {"ms_per_step": ms3, "value": n_part / (ms3 / 1e3), "ranks": parts, "hits": r3.total_hits,
                                       "all_ranks_value_if_kernels_bound": nq / (ms3 / 1e3)}
            del r3, part
            torch.cuda.empty_cache()
        log(f"[bench] shard steps (what a rank of 2 / 4 / 8 runs of this batch, results left on the GPU): {shard_step}")

    # ---- algorithmic bytes, counted by an extra, untimed pass in the exact mode -----------
    lf_steps, fetches, fetch_slots = eng.search_step_stats(queries)
    search_bytes = queries.total_bytes + (8 * nq if args.lookup_depth > 0 else 0) + 60 * lf_steps + 8 * nq
    if aux["seed"]["k"]:  # the seed kernel, the seed-aware verify kernel on what it listed, the general kernel on the rest
        kernel_pattern = ("search_seed_lane_kernel|search_seed_kernel|seed_text_kernel|tile_sums_lists_kernel|search_verify_kernel|" +
                          ("search_pair_kernel" if aux["pair_lines"] else "search_kernel"))
    else:
        kernel_pattern = "search_fast_kernel|search_pair_kernel" if aux["pair_lines"] else "search_kernel"
    search_traffic = traffic_of(pmc, kernel_pattern)
    traffic_source = "live: rocprofv3 --pmc child passes of this run (FETCH_SIZE x 2 + WRITE_SIZE, separate passes)"
    if search_traffic is None:
        search_traffic, traffic_source = committed_traffic(args, nq, aux, pmc_note)
    roofline = {"bound": "hbm", "kernel": (search_traffic or {}).get("kernel", kernel_pattern), "unit": "GB/s",
                "peak": HBM_PEAK_GBPS, "avg_launch_ms": search_ms}
    if search_traffic:
        roofline["traffic"] = search_traffic["bytes"]
        roofline["achieved"] = search_traffic["bytes"] / (search_ms / 1e3) / 1e9
        roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBPS
        # the same traffic over rocprofv3's own average durations of these kernels (the kernel-trace child pass of this run):
        # what profiles/ reproduces; the spread between the two is process to process (where the structures land in HBM)
        rp_ms = rocprof_ms_of(pmc, kernel_pattern)
        if rp_ms:
            roofline["avg_launch_ms_rocprof"] = rp_ms
            roofline["frac_rocprof"] = search_traffic["bytes"] / (rp_ms / 1e3) / 1e9 / HBM_PEAK_GBPS
        roofline["traffic_by_kernel"] = search_traffic.get("by_kernel")
        roofline["traffic_read_bytes"] = search_traffic["read_bytes"]
        roofline["traffic_write_bytes"] = search_traffic["write_bytes"]
        if "read_requests" in search_traffic:
            roofline["dram_read_requests_per_query"] = search_traffic["read_requests"] / nq
            roofline["dram_write_requests_per_query"] = search_traffic["write_requests"] / nq
            roofline["l2_hit_rate"] = search_traffic["l2_hits"] / max(search_traffic["l2_requests"], 1)
            roofline["dram_read_requests_per_launch"] = search_traffic["read_requests"]
            roofline["dram_write_requests_per_launch"] = search_traffic["write_requests"]
            roofline["l2_requests_per_launch"] = search_traffic["l2_requests"]
            roofline["l2_hits_per_launch"] = search_traffic["l2_hits"]
    else:
        roofline.update({"traffic": None, "achieved": None, "frac": None})
    roofline["traffic_source"] = traffic_source
    roofline["algorithmic_bytes_per_launch"] = search_bytes
    roofline["lf_steps_per_launch"] = lf_steps
    roofline["algorithmic_ratio"] = search_bytes / (search_ms / 1e3) / 1e9 / HBM_PEAK_GBPS
    # SURVEY 8(d) literally: algorithmic bytes / kernel time (/ peak).  Above 1 on this index, because the top and jump
    # tables deliver the reference's LF steps with far fewer fetches; the fraction that IS a roofline fraction of the
    # reference's own work is `reference_layout.frac_algorithmic` below (the same kernel family on the reference's arrays)
    roofline["achieved_algorithmic"] = search_bytes / (search_ms / 1e3) / 1e9
    roofline["frac_algorithmic"] = roofline["algorithmic_ratio"]
    # the same number under the name the round-5 review asked for: it is NOT a fraction of anything the kernel moves
    roofline["frac_section8d_headline"] = roofline["algorithmic_ratio"]
    roofline["frac_section8d_headline_label"] = (
        "algorithm substituted: SURVEY 8(d) bytes of the reference's LF steps / time / peak -- above 1 because one seed-table "
        "bucket stands for a read's ~47 LF steps and the full suffix array for the locate walk; `frac` is measured traffic, "
        "reference_layout.frac_algorithmic the 8(d) fraction of the reference's own work")
    # bytes the kernels actually consume per launch: query bytes + one 8-byte offset + one 8-byte top entry + 32 bytes per
    # jump entry (ceil((len - D) / 40) per read: 32 steps + an 8-symbol lookahead each) + the 16-byte record written
    mean_len = queries.total_bytes / max(nq, 1)
    entries = max(0.0, -(-(mean_len - aux["top_table_depth"]) // 40)) if aux["jump_entry_bytes"] == 32 else None
    if aux["seed"]["k"]:
        # query bytes + one 8-byte offset + one 16-byte seed entry + the result written (4 bytes compact, else a 16-byte
        # record; reads longer than k + 32 symbols also compare with text units: not counted)
        useful = run_queries.total_bytes + nq * ((0 if run_queries.uniform_len else 8) + 16 + (4 if runner.use_compact else 16))
        roofline["useful_bytes_per_launch"] = useful
        roofline["useful_bytes_per_query"] = useful / nq
        if search_traffic:
            roofline["wasted_traffic_ratio"] = search_traffic["bytes"] / useful
            roofline["frac_useful"] = useful / (search_ms / 1e3) / 1e9 / HBM_PEAK_GBPS
    elif entries is not None and aux["top_table_depth"]:
        useful = queries.total_bytes + nq * (8 + 8 + 32 * entries + 16)
        roofline["useful_bytes_per_launch"] = useful
        roofline["useful_bytes_per_query"] = useful / nq
        if search_traffic:
            roofline["wasted_traffic_ratio"] = search_traffic["bytes"] / useful
            roofline["frac_useful"] = useful / (search_ms / 1e3) / 1e9 / HBM_PEAK_GBPS
    roofline["note"] = "frac=PMC traffic/time/peak; frac_algorithmic=SURVEY 8d bytes/time/peak (>1: tables replace LF steps)"
    if aux["seed"]["k"]:
        roofline["note_long_seed"] = (
            "The search step is the seed kernel (one 128-byte bucket of the seed table per read: the last k symbols, and for a "
            "k-mer that occurs once its position and the 32 symbols in front), the seed-aware verify kernel on the reads it "
            "listed (k-mers on several rows) and the general kernel on what is left (symbols outside A C G T); traffic and "
            "avg_launch_ms are those of all three launches.  Per read the step moves ~202 bytes of HBM traffic (128 bucket + 58 "
            "query bytes and offset + 16 record): it is bound by HBM bytes -- every random access costs a whole 128-byte line "
            "-- not by instructions or requests in flight.")
    roofline["note_long"] = ("The search step is the fast-path kernel (top table + jumps + lazy tail) followed by the general "
                             "kernel on the few queries it left over; traffic, requests and avg_launch_ms are those of both "
                             "launches together (rocprofv3's kernel stats list them separately). "
                             "frac = measured HBM traffic of the kernel / its live HIP-event duration / 8 TB/s. "
                             "frac_algorithmic = the reference algorithm's logical bytes (60 B per LF step it would execute "
                             "+ query bytes + 8 B result, SURVEY.md 8d) / the same time / 8 TB/s: it exceeds 1 because the "
                             "top table, the jump table and the lazy tail deliver those LF steps with far fewer fetches. "
                             "wasted_traffic_ratio = traffic / the bytes the kernels consume (every 8-byte top entry and "
                             "32-byte jump entry costs a 128-byte DRAM request). The kernel is a chain of dependent random "
                             "128-byte requests; see random_request_model for that bound; reference_layout = the same "
                             "measurement on the reference's arrays alone (frac_algorithmic there is a true 8d fraction).")
    roofline["line_fetches_per_query_exact_mode"] = fetches / nq if fetches else None
    roofline["active_lane_fraction_exact_mode"] = fetches / fetch_slots if fetch_slots else None
    locate_roofline = None
    if do_locate and total_hits:
        acct = out
        if runner.use_rec:  # the accounting pass counts the reference's walk steps from the exact intervals, without hints
            acct = eng.alloc_outputs(nq, hint=False)
            eng.search(queries, acct)
            eng.hit_offsets(acct, nq)
        walk_steps = eng.locate_walk_steps(acct, nq, total_hits, hits, runner.ws[0])
        if runner.use_rec:
            del acct
        locate_bytes = 30 * walk_steps + 4 * total_hits + 8 * total_hits
        if runner.use_compact:  # offsets and the compactly answered hits in one pass, the queue kernel on flagged chunks only
            lt = traffic_of(pmc, "scan2_tile_scan_kernel<true, false>|locate_stream_kernel|locate_queue_kernel")
        else:
            lt = traffic_of(pmc, "locate_stream_kernel|locate_queue_kernel")
        locate_roofline = {"bound": "hbm", "kernel": (lt or {}).get("kernel", "locate_stream_kernel"), "peak": HBM_PEAK_GBPS,
                           "what": "gdx_locate_many_offsets_hits_compact_dev: hit offsets + the hits of the compactly answered reads "
                                   "in one pass over 4 bytes per query (scan2_tile_scan_kernel<true, false>), then the queue kernel "
                                   "on the chunks with slots left open; the totals pass before the host round trip "
                                   "(scan2_tile_sums_kernel) is kernel_ms.totals" if runner.use_compact else
                                   "locate_queue_kernel after the separate scan",
                           "unit": "GB/s", "avg_launch_ms": locate_ms,
                           "traffic": lt["bytes"] if lt else None,
                           "achieved": lt["bytes"] / (locate_ms / 1e3) / 1e9 if lt else None,
                           "frac": lt["bytes"] / (locate_ms / 1e3) / 1e9 / HBM_PEAK_GBPS if lt else None,
                           "dram_read_requests_per_hit": lt["read_requests"] / total_hits if lt and "read_requests" in lt else None,
                           "algorithmic_bytes_per_launch": locate_bytes, "walk_steps_per_launch": walk_steps,
                           "algorithmic_ratio": locate_bytes / (locate_ms / 1e3) / 1e9 / HBM_PEAK_GBPS,
                           "hits_per_launch": total_hits}

    # ---- size-independent parity properties at full size ---------------------------------------------------
    counts = runner.counts(out)
    parity = {"queries_with_status": n_status}
    found = int((counts > 0).sum().item())
    parity["queries_found"] = found
    parity["found_fraction"] = found / nq
    parity["sum_of_counts_equals_hits"] = int(counts.to(torch.int64).sum().item()) == total_hits
    if do_locate and total_hits and args.verify_hits:
        parity.update(verify_hits(torch, io_text, lengths, queries, out, hits, total_hits, nq, args.verify_hits))
        if parity["hits_checked"] != parity["hits_matching_text"]:
            raise SystemExit(f"PARITY FAILURE: {parity}")
    if do_locate and not parity["sum_of_counts_equals_hits"]:
        raise SystemExit(f"PARITY FAILURE: {parity}")

    result = {
        # (BASELINE.json's metric for its configs[2]; the other workloads name themselves)
        "metric": (f"queries/sec ({args.op}), hg38-scale text, 100M len-50 reads" if args.workload == "hg38"
                   else f"queries/sec ({args.op}), {wl['short']}"),
        "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": (f"{wl['short']}, resident in HBM as {INPUT_FORMS[args.input]}; index "
                                f"{index.info.device_bytes / 1e9:.0f} GB/replica = "
                                + ("the library's DEFAULT shape (every build option at its default): " if aux["default_shape"] else "")
                                + f"reference arrays + seed table (k={aux['seed']['k']}) + text units + full SA"
                                + (" + inverse SA" if aux["inverse_suffix_array"] else "")
                                + (" + pair lines" if aux["pair_lines"] else "")
                                + (f" + depth-{aux['top_table_depth']} top table" if aux["top_table_depth"] else "")
                                + f"; {wl['label']}") if aux["seed"]["k"] else
                               (f"{wl['short']}; index {index.info.device_bytes / 1e9:.0f} GB/replica = reference arrays + pair "
                                f"lines + {aux['jump_entry_bytes']}-byte jump entries (with SA) + depth-{aux['top_table_depth']} "
                                f"top table; {wl['label']}"),
                   "index_gb_per_replica": index.info.device_bytes / 1e9,
                   "name": args.workload, "op": args.op, "path": args.path, "input": args.input, "index": args.index,
                   "index_is_library_default": bool(aux["default_shape"]), "queries_per_gpu": nq,
                   "hit_offsets": "u32" if narrow_offsets else "u64",
                   "query_bytes_per_gpu": run_queries.total_bytes + (0 if run_queries.uniform_len else 8 * (nq + 1)),
                   "text_len": wl["total"], "n_texts": wl["n_texts"], "lookup_depth": args.lookup_depth,
                   "sa_rate": args.sa_rate, "index_storage": wl["storage"], "hits_per_gpu": total_hits,
                   "aux_structures": aux,
                   "parallelism": f"index replicated x{world}, queries sharded, gather to rank 0",
                   "rccl_ranks": preflight["ranks"] if preflight else None,
                   "gather_backend": preflight["backend"] if preflight else None,
                   "gather_link_GBps": preflight["gather_link_GBps"] if preflight else None,
                   "preflight": preflight,
                   "gathered_bytes_per_rank_and_step": gathered_bytes,
                   "compact_exceptions": exceptions,
                   "gather_wire": (("found bitmap + positions + exceptions" if getattr(gather, "wire_name", "") == "bitmap" else
                                    "compact results + exceptions" if getattr(gather, "compact_wire", False) else "arrays")
                                   if gather is not None else None)},
        "roofline": roofline,
        "ascii_input": ascii_input,
        "packed_input": packed_input,
        "shard_step": shard_step,
        "locate_roofline": locate_roofline,
        "kernel_ms": {"search": search_ms, "locate": locate_ms, "totals": runner.mean_ms(runner.ev_scan)},
        "parity": parity,
        "index_build_seconds": t_build,
        "index_bytes": int(index.info.device_bytes),
    }

    # ---- BASELINE.json configs[3]: ONE batch of nq queries sharded over the ranks (strong scaling) ---------
    if world > 1 and not args.no_strong:
        del gather, count_of
        runner.outs, runner.hits, runner.ws = [], [], []
        del out, hits, counts
        torch.cuda.empty_cache()
        result["strong_scaling"] = strong_scaling(torch, gdist, eng, io_text, lengths, wl, nq, do_locate, args, rank,
                                                  world, dev)

    if rank == 0 and not args.no_bandwidth:
        result["measured_bandwidth"] = measure_bandwidth(dev)
        bw = result["measured_bandwidth"]
        log(f"[bench] measured bandwidth: {bw}")
        if roofline.get("achieved"):
            roofline["frac_of_measured_stream_read"] = roofline["achieved"] / bw["stream_read_GBps"]
        # The kernel's own bound: dependent random 128-byte requests, whose measured ceiling on this GPU
        # (gather128_group of measure_bandwidth) is well below the streaming peak.
        rq = roofline.get("dram_read_requests_per_query")
        if rq:
            rate = rq * nq / (search_ms / 1e3) / 1e9
            ceiling = bw["gather128_group_Glines_per_s"]
            roofline["random_request_model"] = {"dram_requests_per_launch": rq * nq, "achieved_Greq_per_s": rate,
                                                "measured_ceiling_Greq_per_s": ceiling, "frac_of_ceiling": rate / ceiling}

    # The measurements beside the headline must not take the line with them when the box runs out of something (memory for
    # the 214 GB index of the secondaries, a host allocation): an ordinary exception is reported in the line; a PARITY FAILURE
    # (SystemExit) still ends the run without a number.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(np, torch, index, alpha, queries, runner, do_locate, args, wl)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] cpu_baseline failed: {e!r}")
            result["cpu_baseline"] = {"error": repr(e)}
    else:
        result["cpu_baseline"] = None

    if rank == 0 and world == 1 and not args.no_extras:
        try:
            result["end_to_end"] = end_to_end(np, torch, index, queries, nq, counts, total_hits, ms_per_step, search_ms,
                                               has_pair_lines=aux["pair_lines"])
        except Exception as e:  # noqa: BLE001
            log(f"[bench] end_to_end failed: {e!r}")
            result["end_to_end"] = {"error": repr(e)}

    if rank == 0 and world == 1 and args.secondary_depth > 0 and wl["len_min"] >= 16:
        base_counts = counts.clone()
        runner.outs, runner.hits, runner.ws = [], [], []
        del out, hits, counts, runner
        torch.cuda.empty_cache()
        owned = {"eng": eng, "index": index}  # handed over: the last rung frees the index before building another
        del eng, index
        result["secondary"] = []
        try:
            secondaries(torch, owned, io_text, lengths, alpha, queries, base_counts, nq, do_locate, args, wl, pmc_ref, pmc_text,
                        result.get("end_to_end") if "error" not in (result.get("end_to_end") or {}) else None,
                        res=result["secondary"], pmc_lookup=pmc_lookup)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] secondaries stopped: {e!r}")
            result["secondary_error"] = repr(e)
        for r in result["secondary"]:  # the like-for-like rung, in the keys the driver keeps
            rl = r.get("roofline_reference_layout")
            if rl:
                roofline["reference_layout"] = {
                    "index_bytes": r["index_bytes"], "search_ms": r["search_ms"], "value": r["value"],
                    "frac_traffic": rl.get("frac") if "traffic" in rl else None,
                    "frac_algorithmic": rl.get("algorithmic_ratio", rl.get("frac")),
                    "traffic": rl.get("traffic"), "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"],
                    "dram_read_requests_per_query": rl.get("dram_read_requests_per_query")}
            if r.get("name", "").startswith("reference_arrays_d") and "roofline" in r:  # ... with the reference's lookup tables
                roofline[f"reference_layout_d{r['lookup_depth']}"] = {
                    "value": r["value"], "search_ms": r["search_ms"], "frac_algorithmic": r["roofline"]["frac_algorithmic"],
                    "dram_read_requests_per_query": r["roofline"].get("dram_read_requests_per_query"),
                    "frac_traffic": r["roofline"].get("frac_traffic_from_requests")}

    if rank == 0:
        if world > 1 and "strong_scaling" in result:
            report_strong_scaling(result, wl)
        side = write_side_file(args.side_file, result)
        print(json.dumps(compact_line(result, side)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
