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

WORKLOADS = {
    # BASELINE.md section 3, workload 3/4: 24 texts proportional to hg38, 100 M reads of length 50
    "hg38": dict(total=3_100_000_000, n_texts=24, nq=100_000_000, len_min=50, len_max=50, sampled_ppm=900_000,
                 storage="u32", short="hg38-scale 3.1G text, 100M len-50 reads",
                 label="hg38-scale 3.1G DNA-N text (24 texts), 100M len-50 reads 90% sampled / 10% random"),
    # workload 5: mixed lengths, early termination
    "mixed": dict(total=3_100_000_000, n_texts=24, nq=50_000_000, len_min=20, len_max=150, sampled_ppm=700_000,
                  storage="u32", short="hg38-scale text, 50M reads len 20..150",
                  label="hg38-scale text, 50M reads of length 20..150, 70% sampled / 30% random"),
    # workload 2
    "cfg2": dict(total=1 << 28, n_texts=1, nq=10_000_000, len_min=50, len_max=50, sampled_ppm=900_000,
                 storage="i32", short="256 MB text, 10M len-50 reads", label="256 MB DNA-N text, 10M len-50 reads"),
    "small": dict(total=1 << 24, n_texts=3, nq=1_000_000, len_min=50, len_max=50, sampled_ppm=900_000,
                  storage="i32", short="16 MB text, 1M len-50 reads",
                  label="16 MB DNA-N text, 1M len-50 reads (plumbing check)"),
}

INPUT_FORMS = {"ascii": "IO symbols + u64 offsets", "uniform": "IO symbols, uniform length (no offsets)",
               "packed": "2-bit codes + u64 offsets", "packed+uniform": "2-bit codes, uniform length (no offsets)"}

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


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


# (seed_load_percent: slots of the seed table filled on average.  The library's default of 70 leaves 15 % of the buckets overflowing
# into their neighbours -- a second 128-byte fetch for the reads that land there; at 60 it is 6.6 %: 9 GB more of the 288, the
# step 4.5 % shorter on 100 M reads and 8 % on the 12.5 M a rank of eight runs (profiles/r05/seed_load_sweep.txt))
SEED_INDEX = dict(pair_lines=False, jump_entry_bytes=0, top_table_depth=0, full_suffix_array=True, seed_symbols=True, seed_load_percent=60)
REFERENCE_ARRAYS = dict(pair_lines=False, jump_entry_bytes=0, top_table_depth=0)  # the reference's information content, nothing else
LOOKUP_RUNGS = (10, 13)        # lookup-table depths of the `reference_arrays_dD` secondaries (lookup_table.rs:51-161)
LOOKUP_PMC_READS = 20_000_000  # reads of their PMC child passes
FULL_INDEX = dict(seed_symbols=True, inverse_suffix_array=True, aux_budget_bytes=250_000_000_000)


# ======================================================================================================
# live PMC: rocprofv3 child processes of this same script (--pmc-child), one counter group per pass

PMC_PASSES = [
    ("requests", ["TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_REQ_sum", "TCC_HIT_sum"]),
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
    # no counters: rocprofv3's own kernel durations of the same child workload (`--kernel-trace --stats`), so that the time
    # under the traffic can be the profiler's as well as this process's HIP events (roofline.avg_launch_ms_rocprof)
    ("kernel_trace", None),
]
KERNEL_REGEX = ("search_seed_kernel|search_seed_lane_kernel|seed_text_kernel|tile_sums_lists_kernel|scan2_sums_kernel|search_fast_kernel|search_pair_kernel|locate_queue_kernel|locate_stream_kernel|scan2_tile|"
                "search_kernel|search_verify_kernel|search_exact_kernel")


def pmc_child(args):
    """The workload of the parent, once, without any of its measurements: what rocprofv3 observes."""
    import torch

    from genedex_amd import alphabet
    from genedex_amd.device import DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths, synth_text

    wl = workload_of(args)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    io_text = synth_text(wl["total"], seed=42, n_per_million=10_000, device=dev)
    lengths = hg38_text_lengths(wl["total"], wl["n_texts"])
    index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), sa_rate=args.sa_rate,
                                         lookup_depth=args.lookup_depth, index_storage=wl["storage"],
                                         options=build_options_of(args))
    apply_query_options(index, args)
    nq = wl["nq"]
    queries = DeviceQueries.synth(io_text, lengths, nq, wl["len_min"], wl["len_max"], wl["sampled_ppm"], seed=43)
    eng = DeviceEngine(index)
    runner = StepRunner(torch, eng, input_form(queries, index, args, wl), nq, args.op == "count+locate", args.path,
                        hint=not args.no_hint)
    runner.size()
    for _ in range(args.pmc_child_steps):
        runner.step(0, False)
    torch.cuda.synchronize()
    print(json.dumps({"pmc_child": True, "nq": nq, "hits": runner.total_hits}), flush=True)


def read_kernel_stats(path, out, keep=None):
    """rocprofv3's kernel_stats.csv -> out[kernel short name]["rocprof_avg_ms"] (+ launches); `keep`: copy the file there"""
    import re

    agg = {}
    for row in csv.DictReader(open(path)):
        name = row.get("Name") or row.get("Kernel_Name") or ""
        if not re.search(KERNEL_REGEX, name):
            continue
        a = agg.setdefault(short_kernel_name(name), [0, 0.0])
        a[0] += int(float(row["Calls"]))
        a[1] += float(row["TotalDurationNs"])
    for kern, (calls, total_ns) in agg.items():
        if calls:
            out.setdefault(kern, {})["rocprof"] = {"avg_ms": total_ns / calls / 1e6, "launches": calls}
    if keep:
        try:
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            shutil.copyfile(path, keep)
        except OSError:
            pass


def rocprof_ms_of(pmc, pattern):
    """sum of rocprofv3's average durations (ms) of the kernels `pattern` names ('|'-separated), or None"""
    if not pmc:
        return None
    total = 0.0
    for pat in pattern.split("|"):
        names = [k for k in pmc if pat in k and "stats" not in k]
        if len(names) > 1:
            return None
        if names:
            r = pmc[names[0]].get("rocprof")
            if not r:
                return None
            total += r["avg_ms"]
    return total or None


def run_live_pmc(args, reference_layout=False, rung=None, kernel_trace=False, lookup_depth=None, only=None, nq=None):
    """-> ({kernel short name: {counter: per-launch value}}, None) or (None, reason).  Runs before the parent touches
    the GPU: every pass is `rocprofv3 --pmc <group> -- python3 bench.py --pmc-child ...` in its own process.
    reference_layout: the same workload on an index without any acceleration structure (the ladder's last rung);
    rung = "top16_sa_text": on the 53 GB rung (top table + full suffix array + text units, no jump table, no pair lines).
    lookup_depth / nq: override the parent's; only: the names of the PMC_PASSES to run (default: all)."""
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    child_args = ["--pmc-child", "--workload", args.workload, "--op", args.op, "--path", args.path,
                  "--input", args.input if not (reference_layout or rung) else "ascii",
                  "--lookup-depth", str(args.lookup_depth if lookup_depth is None else lookup_depth), "--sa-rate", str(args.sa_rate),
                  "--index", "tables" if (reference_layout or rung) else args.index]
    jump_bytes, top_depth, no_pairs = args.jump_bytes, args.top_depth, args.no_pair_lines
    if reference_layout:
        jump_bytes, top_depth, no_pairs = 0, 0, True
    if rung == "top16_sa_text":
        jump_bytes, top_depth, no_pairs = 0, 16, True
        child_args += ["--full-sa", "--text-units"]
    # rung == "tables": the library's default structures (--index tables, nothing else)
    for flag, v in (("--nq", args.nq if nq is None else nq), ("--total", args.total), ("--jump-bytes", jump_bytes),
                    ("--top-depth", top_depth), ("--lanes", args.lanes), ("--load-policy", args.load_policy)):
        if v is not None:
            child_args += [flag, str(v)]
    if no_pairs:
        child_args.append("--no-pair-lines")
    if args.no_hint:
        child_args.append("--no-hint")
    out = {}
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.time()
    for name, counters in PMC_PASSES:
        if only is not None and name not in only:
            continue
        d = tempfile.mkdtemp(prefix=f"gdx_pmc_{name}_", dir="/tmp")
        if counters is None:
            if not kernel_trace:
                continue
            cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--",
                   "python3", os.path.join(ROOT, "bench.py"), *child_args, "--pmc-child-steps", "12"]
        else:
            cmd = ["rocprofv3", "--pmc", *counters, "--kernel-include-regex", KERNEL_REGEX,
                   "--output-format", "csv", "-d", d, "--",
                   "python3", os.path.join(ROOT, "bench.py"), *child_args]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
            if counters is None:
                # a failed timing pass does not take the traffic with it
                stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
                if r.returncode == 0 and stats:
                    read_kernel_stats(stats[0], out, keep=kernel_trace if isinstance(kernel_trace, str) else None)
                else:
                    log(f"[bench] kernel-trace child pass failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-300:]}")
                continue
            files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                tail = r.stderr.decode(errors="replace")[-400:]
                return None, f"PMC pass '{name}' failed (rc {r.returncode}): {tail}"
            agg = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    k = (short_kernel_name(row["Kernel_Name"]), row["Counter_Name"])
                    a = agg.setdefault(k, [0, 0.0])
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
            for (kern, counter), (n, s) in agg.items():
                out.setdefault(kern, {})[counter] = {"per_launch": s / n, "launches": n}
        except subprocess.TimeoutExpired:
            return None, f"PMC pass '{name}' timed out"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    log(f"[bench] live PMC passes took {time.time() - t0:.0f}s: {sorted(out)}")
    return out, None


def short_kernel_name(name: str) -> str:
    import re

    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"(gdx::[A-Za-z0-9_]+(?:<[^>(]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][-60:]


def traffic_of(pmc, pattern):
    """HBM bytes per launch of the kernel whose name contains `pattern`: 2 * FETCH_SIZE[KB] * 1024 (every DRAM request
    of gfx950 is 128 B and FETCH_SIZE tallies 64 B each: MI355X_MICROARCH.md section HBM, re-checked on this kernel's
    own access pattern by tools/calibrate_fetch_size.sh) + WRITE_SIZE[KB] * 1024."""
    if not pmc:
        return None
    # a search step may be two launches (the fast-path kernel, then the general kernel on the queries it left over):
    # `pattern` may name several kernels separated by '|'; their per-launch counters are added
    total = None
    for pat in pattern.split("|"):
        names = [k for k in pmc if pat in k and "stats" not in k]
        if len(names) > 1:
            return None
        if not names:
            continue
        c = pmc[names[0]]
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            return None
        res = {"kernel": names[0], "read_bytes": 2.0 * c["FETCH_SIZE"]["per_launch"] * 1024.0,
               "write_bytes": c["WRITE_SIZE"]["per_launch"] * 1024.0}
        res["bytes"] = res["read_bytes"] + res["write_bytes"]
        if "TCC_EA0_RDREQ_sum" in c:
            res["read_requests"] = c["TCC_EA0_RDREQ_sum"]["per_launch"]
            res["write_requests"] = c["TCC_EA0_WRREQ_sum"]["per_launch"]
            res["l2_requests"] = c["TCC_REQ_sum"]["per_launch"]
            res["l2_hits"] = c["TCC_HIT_sum"]["per_launch"]
        if total is None:
            total = res
            total["by_kernel"] = {res["kernel"]: res["bytes"]}
        else:
            total["by_kernel"][res["kernel"]] = res["bytes"]
            total["kernel"] += " + " + res["kernel"]
            for key in ("read_bytes", "write_bytes", "bytes", "read_requests", "write_requests", "l2_requests", "l2_hits"):
                if key in total and key in res:
                    total[key] += res[key]
    return total


def traffic_requests_of(pmc, pattern, queries=LOOKUP_PMC_READS):
    """request counters of the one kernel `pattern` names out of a "requests"-only PMC pass, or None"""
    if not pmc:
        return None
    names = [k for k in pmc if pattern in k and "stats" not in k]
    if len(names) != 1 or "TCC_EA0_RDREQ_sum" not in pmc[names[0]]:
        return None
    c = pmc[names[0]]
    return {"kernel": names[0], "read_requests": c["TCC_EA0_RDREQ_sum"]["per_launch"], "write_requests": c["TCC_EA0_WRREQ_sum"]["per_launch"],
            "l2_requests": c["TCC_REQ_sum"]["per_launch"], "l2_hits": c["TCC_HIT_sum"]["per_launch"], "queries": queries}


# ======================================================================================================

def workload_of(args):
    wl = dict(WORKLOADS[args.workload])
    if args.nq:
        wl["nq"] = args.nq
    if args.total:
        wl["total"] = args.total
    return wl


def input_form(queries, index, args, wl):
    """the batch in the form --input names (made before the timed region; `queries` stays the plain form)"""
    form = getattr(args, "input", "ascii")
    q = queries
    if "uniform" in form and wl["len_min"] != wl["len_max"]:
        raise SystemExit(f"--input {form}: workload {args.workload} has reads of {wl['len_min']}..{wl['len_max']} symbols")
    if "packed" in form:
        q = q.as_packed(index)
    if "uniform" in form:
        q = q.as_uniform(wl["len_min"])
    return q


def build_options_of(args, **override):
    from genedex_amd.index import build_options

    kind = getattr(args, "index", "tables")
    if kind == "default" and not override:
        return build_options()  # nothing asked for: the library's default shape (main() requires aux_structures.default_shape)
    if kind == "seed" and not override:
        return build_options(**SEED_INDEX)
    # (tables: the structures of rounds 1-3 are asked for by name -- with every option at its default the library builds the
    # default shape)
    kw = dict(jump_entry_bytes=args.jump_bytes if (args.jump_bytes is not None or override) else 32, top_table_depth=args.top_depth,
              pair_lines=False if args.no_pair_lines else None,
              full_suffix_array=True if getattr(args, "full_sa", False) else None,
              text_units=True if getattr(args, "text_units", False) else None,
              seed_symbols=getattr(args, "seed_symbols", None), seed_load_percent=getattr(args, "seed_load_percent", None),
              aux_budget_bytes=getattr(args, "aux_budget_bytes", None))
    kw.update(override)
    return build_options(**kw)


def apply_query_options(index, args):
    if args.lanes is not None or getattr(args, "load_policy", None) is not None:
        index.set_query_options(search_lanes=args.lanes, load_policy=getattr(args, "load_policy", None))


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class StepRunner:
    """One timed step of the hot path on resident inputs: search -> offsets scan -> locate, on `n_slots` result sets."""

    def __init__(self, torch, eng, queries, nq, do_locate, path, hint=True, n_slots=1):
        self.torch, self.eng, self.q, self.nq = torch, eng, queries, nq
        self.do_locate = do_locate
        self.use_rec = path in ("records", "records16") and do_locate
        # (compact results only where something fills them: on an index without seed table every entry would say "see the
        # record" and the extra array would only cost its fill and its reads)
        self.use_compact = path == "records" and do_locate and eng.index.seed_info()["k"] != 0
        self.hint = hint and do_locate
        self.n_slots = n_slots
        self.outs = [self._alloc() for _ in range(n_slots)]
        self.total_hits = 0
        self.hits, self.ws = [], []
        self.ev_search, self.ev_locate = [], []
        self.sized_in_step = True  # the timed step reads the number of hits back and sizes the hit buffer itself
        self.scan_ws, self.totals = [], []
        self.ev_scan = []
        self.max_hits = 0  # != 0: queries with more occurrences are counted but not located (gdx.h max_hits)
        # "fused": the step is ONE library call without a host round trip; "split": search + totals, read-back of the totals,
        # offsets + hits (rounds 3-4)
        self.step_mode = "fused"

    def _alloc(self):
        o = self.eng.alloc_outputs(self.nq, hint=self.hint and not self.use_rec)
        if self.use_rec:
            o["rec"] = self.eng.alloc_records(self.nq)
            o["compact"] = self.eng.alloc_compact(self.nq) if self.use_compact else None
        return o

    def search(self, o):
        if self.use_rec:
            self.eng.locate_search(self.q, o["rec"], compact=o["compact"])
        else:
            self.eng.search(self.q, o)

    def offsets(self, o):
        if self.use_rec:
            self.eng.locate_offsets(o["rec"], self.nq, o["hit_offsets"], self.max_hits, compact=o["compact"])
        else:
            self.eng.hit_offsets(o, self.nq)

    def locate(self, o, h, ws):
        if self.use_rec:
            self.eng.locate_hits(o["rec"], self.nq, o["hit_offsets"], self.total_hits, h, ws, compact=o["compact"])
        else:
            self.eng.locate(o, self.nq, self.total_hits, h, ws)

    def counts(self, o):
        """per-query number of occurrences (int32 tensor)"""
        if self.use_rec:
            d = self.torch.sub(o["rec"][:self.nq, 1], o["rec"][:self.nq, 0])
            if o["compact"] is not None:  # -2: see the record; -1: no occurrence; else the position of the only hit
                c = o["compact"][:self.nq]
                d = self.torch.where(c == -2, d, (c != -1).to(self.torch.int32))
            return d
        return self.torch.sub(o["end"], o["start"])

    def status(self, o):
        if self.use_rec:
            s = (o["rec"][:self.nq, 3] >> 24) & 0xff
            if o["compact"] is not None:
                s = self.torch.where(o["compact"][:self.nq] == -2, s, self.torch.zeros_like(s))
            return s
        return o["status"]

    def size(self):
        """sizing pass (also the first warm-up of the kernels): total hits, result buffers"""
        torch = self.torch
        o = self.outs[0]
        self.search(o)
        self.offsets(o)
        torch.cuda.synchronize()
        self.total_hits = int(o["hit_offsets"][self.nq].item()) if self.nq else 0
        dev = o["hit_offsets"].device
        self.hits = [torch.zeros((max(self.total_hits, 1), 2), dtype=torch.int32, device=dev) for _ in range(self.n_slots)]
        nbytes = max(self.eng.locate_workspace_bytes(self.total_hits), 16)
        self.ws = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(self.n_slots)]
        return self.total_hits

    def _step_fused(self, slot, record, after):
        """The whole step as ONE library call without a host round trip (gdx_locate_many_step_compact_layout_dev): the hit
        buffer is the one the sizing pass made (a pipeline offers what its earlier batches needed); the totals stay on the
        device and are checked against the capacity by check_totals() after the timed region."""
        torch = self.torch
        o, h, ws = self.outs[slot], self.hits[slot], self.ws[slot]
        dev = h.device
        if slot >= len(self.scan_ws):
            self.scan_ws = [torch.empty(max(self.eng.totals_workspace_bytes(self.nq), 16), dtype=torch.uint8, device=dev)
                            for _ in range(self.n_slots)]
            self.totals = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(self.n_slots)]
        need = self.eng.locate_workspace_bytes(h.shape[0])
        if need > ws.numel():
            self.ws[slot] = ws = torch.empty(need, dtype=torch.uint8, device=dev)
        narrow = self.n_slots == 1 and h.shape[0] < (1 << 31)
        if narrow and "hit_offsets32" not in o:
            o["hit_offsets32"] = torch.empty(self.nq + 1, dtype=torch.int32, device=dev)
        self.narrow_offsets = narrow
        a, mid, d = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        if record:
            mid.record()  # (creates the event's handle; the library records it again between the step's two halves)
        a.record()
        self.eng.locate_step(self.q, o["rec"], o["compact"], self.scan_ws[slot], self.totals[slot],
                             o["hit_offsets32"] if narrow else o["hit_offsets"], h, ws, max_hits=self.max_hits,
                             event_after_search=mid if record else None)
        d.record()
        if record:
            self.ev_search.append((a, mid))
            self.ev_locate.append((mid, d))
        self.fused_steps = getattr(self, "fused_steps", 0) + 1
        if after is not None:
            after(slot)

    def check_totals(self):
        """after the timed steps of the fused form: every slot's hit total must have fitted the buffer it was offered"""
        if not getattr(self, "fused_steps", 0):
            return
        self.torch.cuda.synchronize()
        for t, h in zip(self.totals, self.hits):
            tot = int(t[0].item())
            if tot > h.shape[0]:
                raise SystemExit(f"PARITY FAILURE: a fused step produced {tot} hits for a buffer of {h.shape[0]}")
            self.total_hits = tot

    def step(self, slot, record, side_stream=None, after=None):
        torch = self.torch
        if (self.do_locate and self.use_compact and side_stream is None
                and getattr(self, "step_mode", "fused") == "fused" and os.environ.get("GDX_BENCH_NO_FOLD") != "1"):
            return self._step_fused(slot, record, after)
        o, h, ws = self.outs[slot], self.hits[slot], self.ws[slot]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # compact path: the hit totals come out of the search call itself (gdx_locate_many_search_totals_compact_layout_dev;
        # GDX_BENCH_NO_FOLD=1: the separate totals pass of round 3)
        fold = self.do_locate and self.use_compact and os.environ.get("GDX_BENCH_NO_FOLD") != "1"
        if fold and slot >= len(self.scan_ws):
            dev_ = h.device
            self.scan_ws = [torch.empty(max(self.eng.totals_workspace_bytes(self.nq), 16), dtype=torch.uint8, device=dev_)
                            for _ in range(self.n_slots)]
            self.totals = [torch.zeros(2, dtype=torch.int64, device=dev_) for _ in range(self.n_slots)]
        a.record()
        if fold:
            self.eng.locate_search_totals(self.q, o["rec"], o["compact"], self.scan_ws[slot], self.totals[slot], self.max_hits)
        else:
            self.search(o)
        b.record()
        if record:
            self.ev_search.append((a, b))
        with (torch.cuda.stream(side_stream) if side_stream is not None else _null()):
            if side_stream is not None:
                side_stream.wait_event(b)
            if self.do_locate and self.use_compact:
                # totals -> the one host round trip (sizes the hit buffer) -> offsets and the hits of the compactly answered
                # reads in ONE pass, the rest from the records (gdx_locate_many_totals_compact_dev / _offsets_hits_compact_dev)
                if slot >= len(self.scan_ws):
                    dev = h.device
                    self.scan_ws = [torch.empty(max(self.eng.totals_workspace_bytes(self.nq), 16), dtype=torch.uint8, device=dev)
                                    for _ in range(self.n_slots)]
                    self.totals = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(self.n_slots)]
                if not fold:
                    ta, tb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ta.record()
                    self.eng.locate_totals(o["rec"], self.nq, self.scan_ws[slot], self.totals[slot], self.max_hits,
                                           compact=o["compact"])
                    tb.record()
                    if record:
                        self.ev_scan.append((ta, tb))
                tot, rest = (int(x) for x in self.totals[slot].tolist())
                self.total_hits = tot
                if tot > self.hits[slot].shape[0]:
                    self.hits[slot] = torch.empty((tot, 2), dtype=torch.int32, device=h.device)
                    h = self.hits[slot]
                need = self.eng.locate_workspace_bytes(tot) if rest else 0
                if need > self.ws[slot].numel():
                    self.ws[slot] = torch.empty(need, dtype=torch.uint8, device=h.device)
                    ws = self.ws[slot]
                c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                # narrow hit offsets (u32[nq + 1], gdx_locate_many_offsets32_hits_compact_dev) when the hits fit 32 bits and
                # nothing else of the step reads them (N = 1); widen_offsets() makes them the step's offsets for every check
                narrow = self.n_slots == 1 and tot < (1 << 31) and os.environ.get("GDX_BENCH_OFFSETS32") != "0"
                if narrow and "hit_offsets32" not in o:
                    o["hit_offsets32"] = torch.empty(self.nq + 1, dtype=torch.int32, device=h.device)
                self.narrow_offsets = narrow
                c.record()
                self.eng.locate_offsets_hits(o["rec"], self.nq, self.scan_ws[slot], o["hit_offsets32"] if narrow else o["hit_offsets"],
                                             tot, rest, h, ws, self.max_hits, compact=o["compact"])
                d.record()
                if record:
                    self.ev_locate.append((c, d))
            elif self.do_locate:
                self.offsets(o)
                if self.sized_in_step:
                    # what a caller cannot skip: the number of hits comes back to the host (one 8-byte copy + a stream
                    # sync) and sizes the hit buffer; the buffers only grow, so a steady state allocates nothing
                    self.total_hits = int(o["hit_offsets"][self.nq].item()) if self.nq else 0
                    if self.total_hits > self.hits[slot].shape[0]:
                        self.hits[slot] = torch.empty((self.total_hits, 2), dtype=torch.int32, device=h.device)
                        h = self.hits[slot]
                    need = self.eng.locate_workspace_bytes(self.total_hits)
                    if need > self.ws[slot].numel():
                        self.ws[slot] = torch.empty(need, dtype=torch.uint8, device=h.device)
                        ws = self.ws[slot]
                c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c.record()
                self.locate(o, h, ws)
                d.record()
                if record:
                    self.ev_locate.append((c, d))
            if after is not None:
                after(slot)

    def widen_offsets(self):
        """after the timed steps: the narrow offsets of the last step become `hit_offsets` (u64), which every check reads"""
        if getattr(self, "narrow_offsets", False):
            for o in self.outs:
                if "hit_offsets32" in o:
                    o["hit_offsets"].copy_(o["hit_offsets32"])
        return getattr(self, "narrow_offsets", False)

    @staticmethod
    def mean_ms(events):
        return float(sum(a.elapsed_time(b) for a, b in events) / len(events)) if events else None


def timed_steps(torch, gdist, runner, steps, warmup, dev, gather=None, count_of=None, overlap=False):
    """W untimed + K timed steps bracketed by barrier + synchronize; -> max-over-ranks seconds"""
    main_stream = torch.cuda.current_stream()
    side_stream = torch.cuda.Stream() if overlap else None
    slot_free = [None] * runner.n_slots
    no = [0]
    # what an N > 1 step does beyond an N = 1 step -- the counts in the gather's type, the hit pairs split into the two
    # arrays that travel -- runs on a stream of its own behind the step's kernels, beside the next step's search, and the
    # gather is enqueued from there (it waits for that stream); acquire(slot) two steps later waits for the gather
    post_stream = torch.cuda.Stream() if gather else None

    def after(slot):
        if gather:
            done = torch.cuda.Event()
            done.record()
            with torch.cuda.stream(post_stream):
                post_stream.wait_event(done)
                count_of(slot)
                gather.submit(slot)
        if overlap:
            slot_free[slot] = torch.cuda.Event()
            slot_free[slot].record()

    def one(record):
        slot = no[0] % runner.n_slots
        no[0] += 1
        if gather:
            gather.acquire(slot)
        if slot_free[slot] is not None:
            main_stream.wait_event(slot_free[slot])
        runner.step(slot, record, side_stream, after)

    for _ in range(warmup):
        one(False)
    if gather:
        gather.drain()
    gdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one(True)
    if gather:
        gather.drain()
    torch.cuda.synchronize()
    gdist.barrier()
    return gdist.max_over_ranks(time.perf_counter() - t0, dev), (no[0] - 1) % runner.n_slots


def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child(args)

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
        same = bool(torch.equal(r2.outs[0]["hit_offsets"], out["hit_offsets"])) and \
            (not do_locate or bool(torch.equal(r2.hits[0][:total_hits], runner.hits[0][:total_hits])))
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
        "metric": "queries/sec (count+locate), hg38-scale text, 100M len-50 reads" if do_locate
        else "queries/sec (count), hg38-scale text, 100M len-50 reads",
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


LINE_LIMIT = 4096  # bytes: the driver keeps a bounded tail of stdout; round 3's 25 KB line was cut and went unparsed


def report_strong_scaling(result, wl):
    """N > 1: BASELINE.json configs[3] is ONE batch of nq reads sharded over the ranks, so `value`, `ms_per_step`,
    `scaling` and `config.workload` become those of the strong-scaling measurement; the every-rank-its-own-batch number
    that was timed first moves to `weak_scaling`."""
    st = result["strong_scaling"]
    result["weak_scaling"] = {"value": result["value"], "ms_per_step": result["ms_per_step"], "unit": "queries/s",
                              "queries_per_gpu": result["config"]["queries_per_gpu"],
                              "gathered_bytes_per_rank_and_step": result["config"]["gathered_bytes_per_rank_and_step"]}
    n = result["n_gpus"]
    result["value"], result["ms_per_step"], result["scaling"] = st["value"], st["ms_per_step"], "strong"
    result["steps"] = st["steps"]
    c = result["config"]
    c["workload"] = (f"{wl['short']}: ONE batch of {st['queries_total']} reads sharded over {n} GPUs (BASELINE configs[3]), index "
                     f"{c['index_gb_per_replica']:.0f} GB replicated, results gathered to rank 0 over RCCL")
    c["queries_per_gpu"] = st["queries_this_rank"]
    c["queries_total"] = st["queries_total"]
    c["gathered_bytes_per_rank_and_step"] = st["gathered_bytes_per_rank_and_step"]
    c["gather_wire"] = st.get("gather_wire", c.get("gather_wire"))
    result["parity"]["shards_equal_single_rank_output"] = st.get("shards_equal_single_rank_output")
    result["results_sharded"] = st.get("results_sharded")


def write_side_file(path, result):
    """Everything measured, in full, beside the contract line (and on stderr); -> the path written or None."""
    log("[bench] full result: " + json.dumps(result))
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(result, f, indent=1)
        return path
    except OSError as e:
        log(f"[bench] side file {path} not written: {e!r}")
        return None


def _pick(d, keys):
    return {k: d[k] for k in keys if d and k in d and d[k] is not None}


def _num(x):
    """numbers at 6 significant digits: the line is for reading and for ratios, the side file keeps every digit"""
    if isinstance(x, float):
        return float(f"{x:.6g}")
    if isinstance(x, dict):
        return {k: _num(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_num(v) for v in x]
    return x


def compact_line(result, side_file=None):
    """The ONE stdout line: the contract's keys, `roofline` and `cpu_baseline` in their short forms, nothing else.
    Guaranteed below LINE_LIMIT bytes (strings are cut, optional parts dropped in a fixed order if it ever grows)."""
    r = result.get("roofline") or {}
    roof = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "avg_launch_ms_rocprof",
                     "frac_rocprof", "frac_algorithmic", "frac_section8d_headline", "algorithmic_bytes_per_launch", "wasted_traffic_ratio",
                     "useful_bytes_per_query", "dram_read_requests_per_query", "l2_hit_rate", "frac_of_measured_stream_read"))
    for k in ("traffic", "achieved", "frac"):  # the contract's keys are there even when nothing was measured (null)
        roof.setdefault(k, r.get(k))
    roof["traffic_source"] = (r.get("traffic_source") or "")[:44]
    if "frac_section8d_headline" in roof:  # (> 1: not a fraction of anything the kernel moves -- named so that nobody takes it for one)
        roof["frac_section8d_headline_label"] = "algorithm substituted: 8d bytes/time/peak, not traffic"
    parts = str(roof.get("kernel") or "").split(" + ")
    if len(parts) > 1:  # the dominant kernel by name, the list kernels of the same step in the side file
        roof["kernel"] = f"{parts[0]} (+ {len(parts) - 1} list kernels of the same step: side file)"
    if r.get("reference_layout"):
        roof["reference_layout"] = _pick(r["reference_layout"], ("index_bytes", "value", "search_ms", "frac_traffic",
                                                                 "frac_algorithmic", "dram_read_requests_per_query"))
        t, a = r["reference_layout"].get("traffic"), r["reference_layout"].get("algorithmic_bytes_per_launch")
        if t and a:
            roof["reference_layout"]["wasted_traffic_ratio"] = t / a
    for d in LOOKUP_RUNGS:
        if r.get(f"reference_layout_d{d}"):
            roof[f"reference_layout_d{d}"] = _pick(r[f"reference_layout_d{d}"], ("value", "search_ms", "frac_algorithmic",
                                                                                 "dram_read_requests_per_query", "frac_traffic"))
    c = result.get("cpu_baseline")
    cpu = c if (c is None or "error" in c) else _pick(c, ("value", "unit", "cores", "kind", "sample", "usable_threads",
                                                             "count_only_value", "bit_exact_vs_gpu"))
    if cpu and isinstance(cpu.get("sample"), str):
        cpu["sample"] = cpu["sample"][:200]
    cfg = result.get("config") or {}
    config = _pick(cfg, ("workload", "index_gb_per_replica", "index_is_library_default", "name", "op", "path", "input", "hit_offsets", "queries_per_gpu", "queries_total", "text_len",
                         "n_texts", "lookup_depth", "sa_rate", "index_storage", "hits_per_gpu", "parallelism", "rccl_ranks",
                         "gather_backend", "gather_link_GBps",
                         "gathered_bytes_per_rank_and_step", "gather_wire", "compact_exceptions"))
    if isinstance(config.get("workload"), str):
        config["workload"] = config["workload"][:260]
    line = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                       "scaling", "vs_baseline", "dtype", "data")}
    # (`value` is timed on the batch in this form -- by default the reference's own: IO symbols + u64 offsets, translated inside
    # the timed region; the same step on a batch translated beforehand is `packed_input`, timed in the same process)
    line["input_form"] = INPUT_FORMS.get(cfg.get("input"), cfg.get("input"))
    line["config"] = config
    line["roofline"] = roof
    line["cpu_baseline"] = cpu
    line["kernel_ms"] = result.get("kernel_ms")
    if result.get("ascii_input"):
        line["ascii_input"] = _pick(result["ascii_input"], ("value", "ms_per_step", "search_ms", "offsets_and_hits_identical_to_headline"))
    if result.get("packed_input"):
        line["packed_input"] = _pick(result["packed_input"], ("value", "ms_per_step", "search_ms", "input", "offsets_and_hits_identical_to_headline"))
    if result.get("shard_step"):
        line["shard_step_ms"] = {k: v["ms_per_step"] for k, v in result["shard_step"].items()}
    if result.get("results_sharded"):
        line["results_sharded"] = _pick(result["results_sharded"], ("value", "ms_per_step"))
    lr = result.get("locate_roofline")
    if lr:
        line["locate_roofline"] = _pick(lr, ("kernel", "avg_launch_ms", "traffic", "frac", "hits_per_launch"))
    line["parity"] = _pick(result.get("parity") or {}, ("queries_with_status", "queries_found", "sum_of_counts_equals_hits",
                                                        "hits_checked", "hits_matching_text", "shards_equal_single_rank_output"))
    if result.get("weak_scaling"):
        line["weak_scaling"] = result["weak_scaling"]
    elif result.get("strong_scaling"):
        line["strong_scaling"] = _pick(result["strong_scaling"], ("value", "ms_per_step", "queries_total"))
    e = result.get("end_to_end")
    if e and "error" not in e:
        line["end_to_end"] = _pick(e, ("count_qps", "locate_qps", "pcie_h2d_GBps", "pcie_d2h_GBps", "pcie_both_directions_GBps_total",
                                            "count_over_bound", "locate_over_bound"))
        if isinstance(e.get("packed_queries"), dict) and "host_packing_GBps_of_ascii" in e["packed_queries"]:
            line["end_to_end"]["host_packing_GBps_of_ascii"] = e["packed_queries"]["host_packing_GBps_of_ascii"]
        if isinstance(e.get("fastq_to_hits"), dict) and "fastq_to_hits_qps" in e["fastq_to_hits"]:
            line["end_to_end"]["fastq_to_hits_qps"] = e["fastq_to_hits"]["fastq_to_hits_qps"]
        if isinstance(e.get("packed_uniform"), dict):
            line["end_to_end"]["packed_uniform"] = _pick(e["packed_uniform"], ("count_qps", "locate_qps", "count_over_bound",
                                                                                  "locate_over_bound", "locate32_qps", "locate32_over_bound",
                                                                                  "locate32_pinned_input_qps"))
    cur = {}
    for r in result.get("secondary") or []:  # BASELINE configs[4]: which index the cursor-API numbers are on
        if str(r.get("name", "")).startswith("exact_intervals_len50") and "HEADLINE" in r["name"]:
            cur["exact_intervals_100M_len50_headline_index_ms"] = r["ms"]
        if str(r.get("name", "")).startswith("mixed_lengths_20_150") and "cursor_api_ms" in r:
            key = "headline_index" if "HEADLINE" in r["name"] else "index_with_every_structure"
            cur[key] = {"index_gb": round(r.get("index_bytes", 0) / 1e9), "cursor_api_ms": r["cursor_api_ms"], "fused_ms": r["fused_ms"]}
    if cur:
        line["cursor_api_50M_len20_150"] = cur
    line["index_build_seconds"] = result.get("index_build_seconds")
    line["side_file"] = side_file
    line = _num(line)
    # (what goes first when the line grows: the side file has everything; the cursor / exact-interval numbers of the headline index
    # and the other input form stay -- they are what makes the headline one index for every BASELINE configuration)
    if len(json.dumps(line)) >= LINE_LIMIT and isinstance(line.get("end_to_end"), dict):
        line["end_to_end"].pop("packed_uniform", None)
    for drop in ("locate_roofline", "shard_step_ms", "end_to_end", "parity", "kernel_ms", "cursor_api_50M_len20_150", "weak_scaling", "strong_scaling"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) >= LINE_LIMIT:  # (cannot happen with the keys above: every string is cut, every list is gone)
        line["config"] = {"workload": config.get("workload", "")[:200]}
    return line


def make_gather(torch, gdist, runner, dev, do_locate):
    """Pads the hit buffers to the largest shard, picks the count type, returns (PipelinedGather, count_of, bytes)."""
    nq = runner.nq
    o = runner.outs[0]
    max_count = int(runner.counts(o).max().item()) if nq else 0
    max_count = gdist.max_int_over_ranks(max_count, dev)
    # (torch's RCCL process group maps int8 / uint8 / int32 / int64 and the float types only: no 16-bit integers)
    count_dtype = torch.uint8 if max_count <= 0xff else torch.int32
    max_hits = gdist.max_int_over_ranks(runner.total_hits, dev)
    max_nq = gdist.max_int_over_ranks(nq, dev)
    # On an index with a seed table the search's compact results travel as they are, with the few queries that have more to
    # say beside them (make_compact_gather) -- whenever that is fewer bytes than the arrays below (it is not on a text of repeats)
    # ... or as a bit per read + 4 bytes per FOUND read (make_bitmap_gather: 3.73 bytes per read where nine in ten are found)
    wire = os.environ.get("GDX_BENCH_GATHER", "auto")
    if do_locate and runner.use_compact and int(runner.eng.index.num_texts()) <= 256 and wire in ("auto", "compact", "bitmap"):
        n_exc, n_exc_hits = gdist.exception_sizes(o["compact"], o["hit_offsets"], nq)
        cap_q = max(gdist.max_int_over_ranks(n_exc, dev), 1)
        cap_h = max(gdist.max_int_over_ranks(n_exc_hits, dev), 1)
        n_found = int((o["compact"][:nq] >= 0).sum().item()) + int((o["compact"][:nq] < -2).sum().item()) if nq else 0
        cap_f = max(gdist.max_int_over_ranks(n_found, dev), 1)
        arrays_bytes = max(max_nq, 1) * (1 if max_count <= 0xff else 4) + 5 * max(max_hits, 1)
        compact_bytes = 4 * max(max_nq, 1) + 4 * cap_q + 5 * cap_h + 8
        layout = gdist.WireLayout(max(max_nq, 1), cap_f, cap_q, cap_h)
        exc = {"queries": n_exc, "hits": n_exc_hits, "found": n_found}
        if wire == "bitmap" or (wire == "auto" and layout.nbytes < min(compact_bytes, arrays_bytes)):
            return make_bitmap_gather(torch, gdist, runner, dev, layout, exc)
        if wire == "compact" or (wire == "auto" and compact_bytes < arrays_bytes):
            return make_compact_gather(torch, gdist, runner, dev, max_nq, cap_q, cap_h, exc)
    runner.hits = [torch.zeros((max(max_hits, 1), 2), dtype=torch.int32, device=dev) for _ in range(runner.n_slots)]
    cnts = [torch.zeros(max(max_nq, 1), dtype=count_dtype, device=dev) for _ in range(runner.n_slots)]
    # Hits travel as two arrays -- text ids as bytes when the collection has at most 256 texts, positions as int32 -- instead
    # of (int32, int32) pairs: 5 instead of 8 bytes per hit over the one xGMI link every rank has to rank 0.  With the seed
    # index a rank produces ~21 G results/s; as pairs that would be 172 GB/s per link, more than a link carries (DESIGN.md
    # section 6), and the gather rather than the kernels would bound the step.  Lossless: rank 0 holds the same hits.
    n_texts = int(runner.eng.index.num_texts())
    split = do_locate and n_texts <= 256
    if split:
        ids = [torch.zeros(max(max_hits, 1), dtype=torch.uint8, device=dev) for _ in range(runner.n_slots)]
        pos = [torch.zeros(max(max_hits, 1), dtype=torch.int32, device=dev) for _ in range(runner.n_slots)]
        gather = gdist.PipelinedGather([[c, i, p] for c, i, p in zip(cnts, ids, pos)], dst=0)
    else:
        gather = gdist.PipelinedGather([[c, h] if do_locate else [c] for c, h in zip(cnts, runner.hits)], dst=0)

    counts32 = [torch.empty(max(nq, 1), dtype=torch.int32, device=dev) for _ in range(runner.n_slots)] if runner.use_rec else None

    def count_of(slot):
        # per-query counts out of the step's results in the gather's count type (copy_ narrows); with records one pass of
        # gdx_locate_many_unpack[_compact]_dev instead of torch arithmetic over the strided 16-byte records -- this runs
        # inside every timed step of an N > 1 run, which an N = 1 run does not have
        if runner.use_rec:
            o_ = runner.outs[slot]
            runner.eng.unpack_records(o_["rec"], nq, counts32[slot], None, compact=o_["compact"])
            cnts[slot][:nq].copy_(counts32[slot][:nq])
        else:
            cnts[slot][:nq].copy_(runner.counts(runner.outs[slot]))
        if split:
            th = min(runner.total_hits, ids[slot].numel())
            h_ = runner.hits[slot]
            ids[slot][:th].copy_(h_[:th, 0])  # (text ids < 256: copy_ narrows)
            pos[slot][:th].copy_(h_[:th, 1])

    if split:
        nbytes = cnts[0].numel() * cnts[0].element_size() + 5 * ids[0].numel()
    else:
        nbytes = cnts[0].numel() * cnts[0].element_size() + (runner.hits[0].numel() * 4 if do_locate else 0)
    gather.hits_are_split = split
    return gather, count_of, nbytes


def make_compact_gather(torch, gdist, runner, dev, max_nq, cap_q, cap_h, exceptions):
    """The gather of a count + locate step on an index with a seed table: the search's compact results travel as they are
    -- 4 bytes per query: the text position of its only hit, "none" or "see the exceptions" -- beside the counts and hits
    of the exceptions (dist.pack_exceptions, sized by the sizing pass: the steps repeat the same batch).  A link into rank 0
    carries one direction of one xGMI link's 153.6 GB/s, so at ~21 G results/s per rank the bytes per result decide the
    step: 4.0x instead of 5.5.  Rank 0 turns every arrived shard into text id + position per query
    (gdx_compact_split_hits_dev, one kernel per shard, enqueued when the gather is acquired) -- inside the timed region."""
    nq = runner.nq
    o = runner.outs[0]
    n = max(max_nq, 1)
    slots = [[torch.full((n,), -1, dtype=torch.int32, device=dev), torch.zeros(cap_q, dtype=torch.int32, device=dev),
              torch.zeros(cap_h, dtype=torch.uint8, device=dev), torch.zeros(cap_h, dtype=torch.int32, device=dev),
              torch.zeros(2, dtype=torch.int32, device=dev)] for _ in range(runner.n_slots)]
    rank, world = gdist.world()
    root_ids = root_pos = None
    if rank == 0:
        root_ids = [[torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(runner.n_slots)]
        root_pos = [[torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(world)] for _ in range(runner.n_slots)]

    def on_gathered(slot, own=False):
        # (rank 0's own shard is on its device in every form already: only a check asks for it in this one)
        for r, words in enumerate(gather.gathered(slot)[0]):
            if (r == 0) == own:
                runner.eng.compact_split_hits(words, n, root_ids[slot][r], root_pos[slot][r])

    gather = gdist.PipelinedGather(slots, dst=0, on_gathered=on_gathered)

    # the search writes its compact results straight into the buffer that travels
    for s_, o_ in zip(slots, runner.outs):
        s_[0][:nq].copy_(o_["compact"][:nq])
        o_["compact"] = s_[0]
    listed = [(torch.zeros(cap_q, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
              for _ in range(runner.n_slots)]

    def pack(slot):
        o_ = runner.outs[slot]
        words, exc_cnt, exc_ids, exc_pos, meta = slots[slot]
        runner.eng.compact_exceptions(words, nq, *listed[slot])
        gdist.pack_exceptions(words, o_["hit_offsets"], runner.hits[slot], nq, exc_cnt, exc_ids, exc_pos, meta, listed[slot])

    gather.hits_are_split = True
    gather.compact_wire = True
    gather.root_ids, gather.root_pos = root_ids, root_pos
    gather.split_own = lambda slot: on_gathered(slot, own=True)
    gather.exceptions = exceptions
    return gather, pack, 4 * n + 4 * cap_q + 5 * cap_h + 8


def make_bitmap_gather(torch, gdist, runner, dev, layout, exceptions):
    """The gather of a count + locate step as a bit per read + the text positions of the found reads (gdx_wire_pack_dev, three
    launches on the rank; dist.WireLayout: everything a rank sends lies in ONE byte buffer, one gather per step): 0.125 + 4 x
    the found fraction bytes per read -- 3.73 where nine reads in ten are found -- instead of the 4 of the compact words.  A
    link into rank 0 carries one direction of an xGMI link, and at ~28 G results/s per rank the bytes per result decide the step
    (DESIGN.md section 6).  Rank 0 turns every arrived shard into text id + position per read (gdx_wire_split_dev, one kernel
    per shard, enqueued when the gather is acquired) -- inside the timed region."""
    nq = runner.nq
    n = layout.n_max
    bufs = [torch.zeros(layout.nbytes, dtype=torch.uint8, device=dev) for _ in range(runner.n_slots)]
    views = [layout.views(b) for b in bufs]
    ws = [torch.empty(max(runner.eng.wire_pack_workspace_bytes(nq), 16), dtype=torch.uint8, device=dev) for _ in range(runner.n_slots)]
    rank, world = gdist.world()
    root_ids = root_pos = None
    if rank == 0:
        root_ids = [[torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(runner.n_slots)]
        root_pos = [[torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(world)] for _ in range(runner.n_slots)]

    def on_gathered(slot, own=False):
        for r, buf in enumerate(gather.gathered(slot)[0]):
            if (r == 0) == own:
                runner.eng.wire_split(layout.views(buf), n, root_ids[slot][r], root_pos[slot][r])

    gather = gdist.PipelinedGather([[b] for b in bufs], dst=0, on_gathered=on_gathered)

    def pack(slot):
        o_ = runner.outs[slot]
        runner.eng.wire_pack(o_["compact"], o_["hit_offsets"], runner.hits[slot], nq, views[slot], ws[slot])

    gather.hits_are_split = True
    gather.compact_wire = True
    gather.wire_name = "bitmap"
    gather.root_ids, gather.root_pos = root_ids, root_pos
    gather.split_own = lambda slot: on_gathered(slot, own=True)
    gather.exceptions = exceptions
    # (exception counts, text ids, positions and the true numbers of one received shard, as expand_split_results takes them)
    gather.exception_parts = lambda slot, r: [layout.views(gather.gathered(slot)[0][r])[k] for k in ("exc_cnt", "exc_ids", "exc_pos", "meta")]
    gather.payload_bytes = layout.payload_bytes(nq, exceptions["found"], exceptions["queries"], exceptions["hits"])
    return gather, pack, layout.nbytes


def gathered_shards(torch, gdist, gather, slot, shard_len, sizes, do_locate):
    """rank 0: (counts, hits or None) of the gathered shards of `slot`, concatenated, as a one-rank run would hold them"""
    parts = gather.gathered(slot)
    world = len(shard_len)
    if getattr(gather, "compact_wire", False):
        gather.split_own(slot)
        cnts, hits = [], []
        for r, (a, b) in enumerate(shard_len):
            exc = gather.exception_parts(slot, r) if hasattr(gather, "exception_parts") else [parts[k][r] for k in (1, 2, 3, 4)]
            c, h = gdist.expand_split_results(gather.root_ids[slot][r], gather.root_pos[slot][r], *exc, b - a)
            if h.shape[0] != sizes[r]:
                raise SystemExit(f"PARITY FAILURE: shard {r} arrived with {h.shape[0]} hits, its rank located {sizes[r]}")
            cnts.append(c)
            hits.append(h)
        return torch.cat(cnts), torch.cat(hits)
    cnt_cat = torch.cat([parts[0][r][: b - a] for r, (a, b) in enumerate(shard_len)])
    if do_locate and getattr(gather, "hits_are_split", False):  # (text ids as bytes, positions as int32: back to pairs)
        hit_cat = torch.cat([torch.stack([parts[1][r][: sizes[r]].to(torch.int32), parts[2][r][: sizes[r]]], dim=1)
                             for r in range(world)])
    else:
        hit_cat = torch.cat([parts[1][r][: sizes[r]] for r in range(world)]) if do_locate else None
    return cnt_cat, hit_cat


def strong_scaling(torch, gdist, eng, io_text, lengths, wl, nq_total, do_locate, args, rank, world, dev):
    """BASELINE.json configs[3]: the N = 1 batch (seed 43) split into `world` contiguous shards (dist.shard_range), one
    per rank, results gathered to rank 0; value = nq_total / max-over-ranks step time.  Rank 0 also runs the whole
    batch alone once and requires the concatenated shard results to equal it bit for bit."""
    from genedex_amd.device import DeviceQueries

    full = DeviceQueries.synth(io_text, lengths, nq_total, wl["len_min"], wl["len_max"], wl["sampled_ppm"], seed=43)
    # rank 0's shard relative to the others': from the rate this run's links deliver into rank 0 (a probe gather) and the
    # one-GPU costs of a step and of the root's split (dist.root_weight_for)
    link_rate = gdist.gather_rate_probe(dev)
    seeded = eng.index.seed_info()["k"] != 0 and do_locate
    root_weight = args.root_weight if args.root_weight is not None else \
        gdist.root_weight_for(world, nq_total, link_rate, gdist.WIRE_BYTES_PER_READ if seeded else 5.5)
    lo, hi = gdist.shard_range(nq_total, rank, world, root_weight)
    # (a rank holds its shard as a batch of its own: the form --input names is made from that)
    shard = input_form(full.copy_slice(lo, hi) if args.input != "ascii" else full.slice(lo, hi), eng.index, args, wl)
    runner = StepRunner(torch, eng, shard, hi - lo, do_locate, args.path, hint=not args.no_hint, n_slots=2)
    runner.size()
    gather, count_of, nbytes = make_gather(torch, gdist, runner, dev, do_locate)
    steps = max(args.steps, 1)
    elapsed, last = timed_steps(torch, gdist, runner, steps, args.warmup, dev, gather, count_of)
    ms = elapsed / steps * 1e3
    res = {"scaling": "strong", "value": nq_total / (ms / 1e3), "unit": "queries/s", "ms_per_step": ms,
           "queries_total": nq_total, "queries_this_rank": hi - lo, "steps": steps, "root_weight": root_weight, "gather_probe_GBps_per_link": link_rate,
           "kernel_ms_rank0": {"search": runner.mean_ms(runner.ev_search), "locate": runner.mean_ms(runner.ev_locate)},
           "gathered_bytes_per_rank_and_step": nbytes}
    runner.check_totals()
    # the same sharded step with the results LEFT on their GPUs (no gather): what the kernels and launches of N ranks give; the
    # gather above adds what one direction of the links into rank 0 carries
    e_ng, _ = timed_steps(torch, gdist, runner, steps, args.warmup, dev)
    runner.check_totals()
    res["results_sharded"] = {"value": nq_total / (e_ng / steps), "unit": "queries/s", "ms_per_step": e_ng / steps * 1e3,
                              "what": "the sharded step without the gather: every rank's offsets and hits stay in its own HBM"}
    # bit-exactness: concatenated shards == the one-rank output
    sizes = gdist.gather_ints(runner.total_hits, dev)
    if rank == 0:
        shard_len = [gdist.shard_range(nq_total, r, world, root_weight) for r in range(world)]
        cnt_cat, hit_cat = gathered_shards(torch, gdist, gather, last, shard_len, sizes, do_locate)
        res["gather_wire"] = getattr(gather, "wire_name", "compact" if getattr(gather, "compact_wire", False) else "arrays")
        del gather, runner
        torch.cuda.empty_cache()
        single = StepRunner(torch, eng, input_form(full, eng.index, args, wl), nq_total, do_locate, args.path, hint=not args.no_hint)
        single.size()
        single.step(0, False)
        torch.cuda.synchronize()
        same_counts = bool(torch.equal(cnt_cat.to(torch.int64), single.counts(single.outs[0]).to(torch.int64)))
        same_hits = bool(torch.equal(hit_cat, single.hits[0][: single.total_hits])) if do_locate else None
        res["shards_equal_single_rank_output"] = {"counts": same_counts, "hits": same_hits}
        if not same_counts or same_hits is False:
            raise SystemExit(f"PARITY FAILURE: sharded results differ from the one-rank output: {res}")
    return res


def committed_traffic(args, nq, aux, why):
    """Fallback when the live PMC passes are unavailable: the committed summary of the same configuration."""
    path = os.path.join(ROOT, "profiles", "r05", "search_pmc_final.json")
    try:
        with open(path) as f:
            p = json.load(f)
        if ((p["workload"], p["lookup_depth"], p["path"], p["jump_entry_bytes"], p["top_table_depth"], p.get("seed_k", 0),
             p.get("input", "ascii"))
                != (args.workload, args.lookup_depth, args.path, aux["jump_entry_bytes"], aux["top_table_depth"], aux["seed"]["k"],
                    getattr(args, "input", "ascii"))):
            return None, f"unavailable ({why}; the committed summary is of another configuration)"
        scale = nq / p["queries_per_launch"]
        t = {"kernel": p["kernel"], "read_bytes": p["read_bytes_per_launch"] * scale,
             "write_bytes": p["write_bytes_per_launch"] * scale, "read_requests": p["read_requests_per_launch"] * scale,
             "write_requests": p["write_requests_per_launch"] * scale, "l2_requests": p["l2_requests_per_launch"] * scale,
             "l2_hits": p["l2_hits_per_launch"] * scale}
        t["bytes"] = t["read_bytes"] + t["write_bytes"]
        return t, (f"NOT measured in this run ({why}); committed summary {os.path.relpath(path, ROOT)} of the same "
                   f"configuration")
    except (OSError, KeyError, ValueError):
        return None, f"unavailable ({why})"


def time_config(torch, eng, queries, nq, do_locate, args, steps=3):
    """(ms per step, search ms, locate ms, counts) of the resident index in its current configuration"""
    runner = StepRunner(torch, eng, queries, nq, do_locate, args.path, hint=not args.no_hint)
    runner.size()
    runner.step(0, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.step(0, True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    counts = runner.counts(runner.outs[0]).clone()
    return ms, runner.mean_ms(runner.ev_search), runner.mean_ms(runner.ev_locate), counts


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
        # (seed table AND the library's default structures: a read from a repeat goes on from its seed entry's interval with
        # one jump round -- search_seed_kernel4 -> search_fast_kernel4 over its list -> the general kernel)
        # + the full suffix array: the hits of a read from a repeat are consecutive rows -- 32 of their SA values per 128-byte
        # line there, 4 per line inside the 32-byte jump entries (scan + locate of 573 M hits 5.35 -> 4.0 ms)
        both = ({"index": "tables", "seed_symbols": 1, "full_sa": True, "aux_budget_bytes": 250_000_000_000}
                if seed_family else {})
        res.append(genome_like_secondary(torch, alpha, wl, argparse.Namespace(**{**vars(args), **both})))
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


def end_to_end(np, torch, index, queries, nq, dev_counts, total_hits, step_ms, search_ms, has_pair_lines=True):
    """SURVEY.md 8d "wall-clock incl. H2D/D2H": the host-pointer calls a genedex caller would make (queries as &[u8] in
    host memory, lib.rs:155-185; results into host arrays), which run as a chunked H2D || kernels || D2H pipeline
    (host_api.hip).  Never `value`.  The PCIe rates are measured here with pinned 1 GiB copies."""
    import ctypes as C

    from genedex_amd import _lib

    lib = _lib.load()
    dev = queries.qbuf.device
    nbytes = queries.total_bytes
    qbuf = queries.qbuf[:nbytes].cpu().numpy()
    qoff = queries.qoff.cpu().numpy().astype(np.uint64)
    pin = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    dbuf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)

    def copy_rate(dst, src):
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return (1 << 30) / best / 1e9

    h2d, d2h = copy_rate(dbuf, pin), copy_rate(pin, dbuf)
    # both directions at once on two streams: what a pipeline that copies in and out together gets of the link (this platform
    # serves the two directions at not much more than ONE direction's rate in all -- profiles/r05/README.md -- so the bound
    # of a host-pointer call is (bytes in + bytes out) / this rate, not the slower of the two directions alone)
    pin2 = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    dbuf2 = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()
    duplex = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(s_in):
            dbuf.copy_(pin, non_blocking=True)
        with torch.cuda.stream(s_out):
            pin2.copy_(dbuf2, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        duplex = dt if duplex is None or dt < duplex else duplex
    duplex = 2 * (1 << 30) / duplex / 1e9
    del pin, dbuf, pin2, dbuf2
    counts = np.empty(nq, dtype=np.uint64)
    status = np.empty(nq, dtype=np.uint8)
    u8p, u64p = _lib.u8p, _lib.u64p

    def count_call():
        _lib.check(lib.gdx_count_many(index._h, qbuf.ctypes.data_as(u8p), qoff.ctypes.data_as(u64p), nq,
                                      counts.ctypes.data_as(u64p), status.ctypes.data_as(u8p)))

    def best_of(fn, reps=2):
        fn()  # the first call also sizes the pinned staging buffers
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return best

    t_count = best_of(count_call)
    same_counts = bool(np.array_equal(counts, dev_counts.cpu().numpy().astype(np.uint64) & np.uint64(0xFFFFFFFF)))
    offs = np.empty(nq + 1, dtype=np.uint64)
    total = C.c_uint64(0)
    last = {}

    def locate_call():
        ptr = C.POINTER(_lib.HitStruct)()
        _lib.check(lib.gdx_locate_many_alloc(index._h, qbuf.ctypes.data_as(u8p), qoff.ctypes.data_as(u64p), nq,
                                             offs.ctypes.data_as(u64p), C.byref(ptr), C.byref(total),
                                             status.ctypes.data_as(u8p)))
        last["ptr"] = ptr

    t_locate = None
    for _ in range(3):  # (the first call also sizes the pinned staging buffers)
        if last.get("ptr"):
            lib.gdx_free_hits(last.pop("ptr"))
        t0 = time.perf_counter()
        locate_call()
        dt = time.perf_counter() - t0
        t_locate = dt if t_locate is None or dt < t_locate else t_locate
    same_total = total.value == total_hits and int(offs[-1]) == total_hits
    if last.get("ptr"):
        lib.gdx_free_hits(last.pop("ptr"))
    if not same_counts or not same_total:
        raise SystemExit("PARITY FAILURE: the host-pointer calls disagree with the device-resident path")
    in_bytes = nbytes + 8 * (nq + 1)
    out_count_bytes = 5 * nq  # u32 count + status byte per query on the wire, widened to u64 by the host threads
    out_locate_bytes = 5 * nq + 8 * total_hits
    def bound(n_in, n_out, kernel_ms):
        return max(n_in / (h2d * 1e9), n_out / (d2h * 1e9), (n_in + n_out) / (duplex * 1e9), kernel_ms / 1e3)

    bound_count = bound(in_bytes, out_count_bytes, search_ms)
    bound_locate = bound(in_bytes, out_locate_bytes, step_ms)
    res = {"count_qps": nq / t_count, "count_seconds": t_count, "locate_qps": nq / t_locate, "locate_seconds": t_locate,
           "pcie_h2d_GBps": h2d, "pcie_d2h_GBps": d2h, "pcie_both_directions_GBps_total": duplex, "h2d_bytes": in_bytes,
           "d2h_bytes_count": out_count_bytes,
           "d2h_bytes_locate": out_locate_bytes,
           "count_over_bound": t_count / bound_count, "locate_over_bound": t_locate / bound_locate,
           "bound": "max(H2D bytes / measured H2D rate, D2H bytes / measured D2H rate, (H2D + D2H bytes) / the rate of both "
                    "directions at once, kernel time)",
           "calls": "gdx_count_many / gdx_locate_many_alloc on pageable host arrays (ASCII queries, u64 offsets), results "
                    "identical to the device-resident path", "query_packing": "none (ASCII) for count_qps / locate_qps",
           "results_identical_to_device_path": {"counts": same_counts, "hits_total": same_total}}
    res["packed_queries"] = packed_end_to_end(np, torch, index, queries, nq, dev_counts, h2d, d2h, search_ms)
    # the same two calls on the batch as 2-bit codes without offsets (gdx_query_layout_t: packed + uniform) when every read has
    # the same length: 12.5 instead of 58 bytes per len-50 read over PCIe, nothing to stage but the codes
    lens = (queries.qoff[1: nq + 1] - queries.qoff[:nq]) if nq else None
    if nq and bool((lens == lens[0]).all().item()) and int(lens[0]) > 0:
        ulen = int(lens[0])
        packed = np.zeros(int(lib.gdx_packed_bytes(int(qoff[-1]))), dtype=np.uint8)
        n_exc = C.c_uint64(0)
        rc = lib.gdx_pack_queries(index._h, qbuf.ctypes.data_as(u8p), qoff.ctypes.data_as(u64p), nq, packed.ctypes.data_as(u8p),
                                  None, 0, C.byref(n_exc))
        if rc == 0 and n_exc.value == 0:
            lay = _lib.QueryLayout()
            lib.gdx_query_layout_init(C.byref(lay))
            lay.packed, lay.uniform_len = 1, ulen

            def count_pu():
                _lib.check(lib.gdx_count_many_layout(index._h, packed.ctypes.data_as(u8p), None, nq, C.byref(lay),
                                                     counts.ctypes.data_as(u64p), status.ctypes.data_as(u8p)))

            t_c = best_of(count_pu)
            same_c = bool(np.array_equal(counts, dev_counts.cpu().numpy().astype(np.uint64) & np.uint64(0xFFFFFFFF)))

            def locate_pu():
                ptr = C.POINTER(_lib.HitStruct)()
                _lib.check(lib.gdx_locate_many_alloc_layout(index._h, packed.ctypes.data_as(u8p), None, nq, C.byref(lay),
                                                            offs.ctypes.data_as(u64p), C.byref(ptr), C.byref(total),
                                                            status.ctypes.data_as(u8p)))
                last["ptr"] = ptr

            t_l = None
            for _ in range(3):
                if last.get("ptr"):
                    lib.gdx_free_hits(last.pop("ptr"))
                t0 = time.perf_counter()
                locate_pu()
                dt = time.perf_counter() - t0
                t_l = dt if t_l is None or dt < t_l else t_l
            same_t = total.value == total_hits and int(offs[-1]) == total_hits
            # the narrow form (gdx_locate_many_alloc_layout32): u32 offsets + 8-byte hits in pinned memory of the library's, written
            # by the device; from the pageable array and from a pinned copy of it (no staging copy on the way in)
            def locate32(qptr):
                res = _lib.Hits32()
                _lib.check(lib.gdx_locate_many_alloc_layout32(index._h, qptr, None, nq, C.byref(lay), C.byref(res),
                                                              status.ctypes.data_as(u8p)))
                return res

            def time32(qptr):
                best, res = None, None
                for _ in range(3):
                    if res is not None:
                        lib.gdx_free_hits32(C.byref(res))
                    t0 = time.perf_counter()
                    res = locate32(qptr)
                    dt = time.perf_counter() - t0
                    best = dt if best is None or dt < best else best
                return best, res

            t_l32, res32 = time32(packed.ctypes.data_as(u8p))
            same_32 = res32.total_hits == total_hits
            if same_32 and last.get("ptr") and total_hits:  # the same offsets and hits as the wide call
                o32 = np.ctypeslib.as_array(res32.hit_offsets, shape=(nq + 1,))
                same_32 = bool(np.array_equal(o32, offs.astype(np.uint32)))
                n_cmp = min(total_hits, 4_000_000)
                h32 = np.ctypeslib.as_array(res32.hits, shape=(2 * n_cmp,)).reshape(n_cmp, 2)
                h64 = np.ctypeslib.as_array(C.cast(last["ptr"], _lib.u64p), shape=(2 * n_cmp,)).reshape(n_cmp, 2)
                same_32 = same_32 and bool(np.array_equal(h32, h64.astype(np.uint32)))
                tail32 = np.ctypeslib.as_array(res32.hits, shape=(2 * total_hits,))[-2 * n_cmp:]
                tail64 = np.ctypeslib.as_array(C.cast(last["ptr"], _lib.u64p), shape=(2 * total_hits,))[-2 * n_cmp:]
                same_32 = same_32 and bool(np.array_equal(tail32, tail64.astype(np.uint32)))
            lib.gdx_free_hits32(C.byref(res32))
            pinned_in = torch.from_numpy(packed).pin_memory()
            t_l32p, res32p = time32(C.cast(C.c_void_p(pinned_in.data_ptr()), u8p))
            same_32 = same_32 and res32p.total_hits == total_hits
            lib.gdx_free_hits32(C.byref(res32p))
            del pinned_in
            lib.gdx_release_cached_hits()
            if last.get("ptr"):
                lib.gdx_free_hits(last.pop("ptr"))
            if not same_c or not same_t or not same_32:
                raise SystemExit("PARITY FAILURE: the packed + uniform host calls disagree with the device-resident path")
            in_pu = (nq * ulen + 3) // 4
            # what the narrow call's results cross the link as (host_api.hip: the found-bitmap wire, expanded by host threads): a bit
            # per read, 4 bytes (+ a text id byte) per read with one hit, {read, count} + 8 bytes per hit for the others with hits
            cnts = np.diff(offs.astype(np.int64))
            n_one, n_more = int((cnts == 1).sum()), int((cnts > 1).sum())
            id_bytes = 1 if int(index.info.num_texts) > 1 else 0
            out_locate32_bytes = nq // 8 + 8 * (nq // 2048 + 2) + (4 + id_bytes) * n_one + 8 * n_more + 8 * int(cnts[cnts > 1].sum())
            del cnts
            res["packed_uniform"] = {
                "count_qps": nq / t_c, "count_seconds": t_c, "locate_qps": nq / t_l, "locate_seconds": t_l, "h2d_bytes": in_pu,
                "count_over_bound": t_c / bound(in_pu, out_count_bytes, search_ms),
                "locate_over_bound": t_l / bound(in_pu, out_locate_bytes, step_ms),
                "locate32_qps": nq / t_l32, "locate32_seconds": t_l32,
                "locate32_over_bound": t_l32 / bound(in_pu, out_locate32_bytes, step_ms),
                "locate32_pinned_input_qps": nq / t_l32p, "locate32_pinned_input_seconds": t_l32p,
                "locate32_pinned_input_over_bound": t_l32p / bound(in_pu, out_locate32_bytes, step_ms),
                "d2h_bytes_locate32": out_locate32_bytes,
                "calls": "gdx_count_many_layout / gdx_locate_many_alloc_layout, layout = {packed, uniform_len}: 2-bit codes, no "
                         "offsets; locate32 = gdx_locate_many_alloc_layout32 (u32 offsets + 8-byte hits in pinned memory of the "
                         "library's; the results cross PCIe as the found-bitmap wire -- d2h_bytes_locate32 -- and host threads expand "
                         "them; pinned_input: the 2-bit codes lie in pinned memory too, no staging copy)",
                "results_identical_to_device_path": {"counts": same_c, "hits_total": same_t, "narrow_equals_wide": same_32}}
    try:
        res["fastq_to_hits"] = fastq_to_hits(np, index, qbuf, qoff, nq, offs)
    except OSError as e:  # (no room for the file)
        res["fastq_to_hits"] = {"error": repr(e)}
    log(f"[bench] end to end: {res}")
    return res


def fastq_to_hits(np, index, qbuf, qoff, nq, offs, n_reads=24_000_000, batch_reads=8_000_000):
    """A FASTQ file of the batch's first reads -> gdx_fastx_next_batch_ex (the library's reader: the file memory-mapped, a batch
    parsed by all host threads the process may use) -> gdx_pack_queries_table (2-bit codes, host threads) ->
    gdx_locate_many_alloc_layout32, reader and packer one batch ahead of the GPU calls in a thread of their own.  What the
    reference's ROADMAP.md:35-37 worries about: reading the queries can cost more than searching them -- it still does (the
    kernels take 25 G reads a second), but by one order of magnitude less than with round 5's single parsing thread."""
    import ctypes as C
    import queue
    import tempfile
    import threading

    from genedex_amd import _lib, alphabet, fastx

    lib = _lib.load()
    n = int(min(n_reads, nq))
    lens = np.diff(qoff[: n + 1].astype(np.int64))
    if n == 0 or not bool((lens == lens[0]).all()):
        return None
    ln = int(lens[0])
    rec = np.empty((n, ln * 2 + 7), dtype=np.uint8)  # "@r\n" + read + "\n+\n" + quality + "\n"
    rec[:, 0], rec[:, 1], rec[:, 2] = ord("@"), ord("r"), 10
    rec[:, 3: 3 + ln] = qbuf[: n * ln].reshape(n, ln)
    rec[:, 3 + ln], rec[:, 4 + ln], rec[:, 5 + ln] = 10, ord("+"), 10
    rec[:, 6 + ln: 6 + 2 * ln] = ord("I")
    rec[:, 6 + 2 * ln] = 10
    with tempfile.NamedTemporaryFile(prefix="gdx_bench_", suffix=".fq", dir="/tmp", delete=False) as f:
        path = f.name
    try:
        rec.tofile(path)
        file_bytes = os.path.getsize(path)
        del rec
        alpha = alphabet.ascii_dna_with_n()
        t0 = time.perf_counter()
        n_read = sum(qo.size - 1 for _, qo in fastx.read_batches(path, max_records=batch_reads, buffer_bytes=batch_reads * ln))
        t_reader = time.perf_counter() - t0
        os.environ["GDX_FASTX_THREADS"] = "0"  # (round 5's reader, for the record: one thread, a streaming read of the file)
        t0 = time.perf_counter()
        n_read1 = sum(qo.size - 1 for _, qo in fastx.read_batches(path, max_records=batch_reads, buffer_bytes=batch_reads * ln))
        t_reader1 = time.perf_counter() - t0
        del os.environ["GDX_FASTX_THREADS"]
        q = queue.Queue(maxsize=1)

        def producer():
            # (three buffer sets: one being filled, one in the queue, one in the GPU call -- no copy of a batch)
            for b in fastx.read_packed_batches(path, alpha, max_records=batch_reads, buffer_bytes=batch_reads * ln, n_buffers=3):
                q.put((b["packed"], b["nq"], b["uniform_len"], b["exceptions"].size))
            q.put(None)

        lay = _lib.QueryLayout()
        lib.gdx_query_layout_init(C.byref(lay))
        status = np.empty(batch_reads, dtype=np.uint8)
        t0 = time.perf_counter()
        th = threading.Thread(target=producer)
        th.start()
        hits, reads, n_exc = 0, 0, 0
        while True:
            item = q.get()
            if item is None:
                break
            packed, bn, ul, ne = item
            lay.packed, lay.uniform_len = 1, ul
            r32 = _lib.Hits32()
            _lib.check(lib.gdx_locate_many_alloc_layout32(index._h, packed.ctypes.data_as(_lib.u8p), None, bn, C.byref(lay),
                                                          C.byref(r32), status.ctypes.data_as(_lib.u8p)))
            hits += r32.total_hits
            reads += bn
            n_exc += ne
            lib.gdx_free_hits32(C.byref(r32))
        th.join()
        dt = time.perf_counter() - t0
        same = reads == n and n_read == n and n_read1 == n and n_exc == 0 and hits == int(offs[n])
        if not same:
            raise SystemExit(f"PARITY FAILURE: FASTQ -> hits gave {reads} reads / {hits} hits, the device path {n} / {int(offs[n])}")
        return {"reads": n, "file_bytes": file_bytes, "fastq_to_hits_qps": n / dt, "seconds": dt, "file_GBps": file_bytes / dt / 1e9,
                "reader_alone_qps": n / t_reader, "reader_alone_file_GBps": file_bytes / t_reader / 1e9,
                "reader_alone_one_thread_qps": n / t_reader1, "batch_reads": batch_reads,
                "hits": hits, "hits_identical_to_device_path": same,
                "what": "FASTQ file -> gdx_fastx_next_batch_ex (mapped file, blocks parsed in parallel) -> gdx_pack_queries_table -> "
                        "gdx_locate_many_alloc_layout32, reader and packer one batch ahead in a thread of their own"}
    finally:
        os.remove(path)


def packed_end_to_end(np, torch, index, queries, nq, dev_counts, h2d, d2h, search_ms):
    """The count call on 2-bit packed queries (include/gdx.h "packed queries"; pair-line kernels): a quarter of the query
    bytes over PCIe; packing is done once by gdx_pack_queries (host threads) and timed separately -- a caller that stores
    its reads packed never pays it."""
    import ctypes as C

    from genedex_amd import _lib

    lib = _lib.load()
    dev = queries.qbuf.device
    nbytes = queries.total_bytes
    qbuf = queries.qbuf[:nbytes].cpu().numpy()
    qoff = queries.qoff.cpu().numpy().astype(np.uint64)
    counts = dev_counts.cpu().numpy().astype(np.uint64) & np.uint64(0xFFFFFFFF)
    status = np.empty(nq, dtype=np.uint8)
    u8p, u64p = _lib.u8p, _lib.u64p

    def best_of(fn, reps=2):
        fn()
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return best

    packed = np.empty(int(lib.gdx_packed_bytes(nbytes)), dtype=np.uint8)
    exc = np.empty(1 << 20, dtype=np.uint64)
    n_exc = C.c_uint64(0)
    t_pack = None
    for _ in range(2):  # (the first call also touches the pages of `packed` for the first time)
        t0 = time.perf_counter()
        _lib.check(lib.gdx_pack_queries(index._h, qbuf.ctypes.data_as(u8p), qoff.ctypes.data_as(u64p), nq,
                                        packed.ctypes.data_as(u8p), exc.ctypes.data_as(u64p), exc.size, C.byref(n_exc)))
        dt = time.perf_counter() - t0
        t_pack = dt if t_pack is None or dt < t_pack else t_pack
    counts_p = np.empty(nq, dtype=np.uint64)

    def count_packed_call():
        _lib.check(lib.gdx_count_many_packed(index._h, packed.ctypes.data_as(u8p), qoff.ctypes.data_as(u64p), nq,
                                             counts_p.ctypes.data_as(u64p), status.ctypes.data_as(u8p)))

    t_count_packed = best_of(count_packed_call)
    keep = np.ones(nq, dtype=bool)
    keep[exc[: n_exc.value].astype(np.int64)] = False
    same_packed = bool(np.array_equal(counts_p[keep], counts[keep]))
    if not same_packed:
        raise SystemExit("PARITY FAILURE: packed queries give other counts than ASCII queries")
    # device-resident: the search kernel on packed input (records mode), packed on the device from the ASCII batch
    d_packed = torch.zeros(packed.size, dtype=torch.uint8, device=dev)
    d_bad = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.gdx_pack_queries_dev(index._h, C.c_void_p(queries.qbuf.data_ptr()), nbytes, C.c_void_p(d_packed.data_ptr()),
                                        None, C.c_void_p(d_bad.data_ptr()), stream))
    rec = torch.empty((nq, 4), dtype=torch.int32, device=dev)

    def packed_search():
        _lib.check(lib.gdx_locate_many_search_packed_dev(index._h, C.c_void_p(d_packed.data_ptr()),
                                                         C.c_void_p(queries.qoff.data_ptr()), nq, C.c_void_p(rec.data_ptr()),
                                                         stream))

    packed_search()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        packed_search()
    ev[1].record()
    torch.cuda.synchronize()
    packed_search_ms = ev[0].elapsed_time(ev[1]) / 3
    same_dev = bool(torch.equal((rec[:, 1] - rec[:, 0])[torch.from_numpy(keep).to(dev)],
                                dev_counts[torch.from_numpy(keep).to(dev)]))
    if not same_dev:
        raise SystemExit("PARITY FAILURE: the packed device search gives other counts")
    del d_packed, rec
    packed_in_bytes = nbytes // 4 + 8 * (nq + 1)
    out_count_bytes = 5 * nq
    return {"count_qps": nq / t_count_packed, "count_seconds": t_count_packed, "h2d_bytes": packed_in_bytes,
            "count_over_bound": t_count_packed / max(packed_in_bytes / (h2d * 1e9), out_count_bytes / (d2h * 1e9), search_ms / 1e3),
            "host_packing_seconds_not_included": t_pack, "host_packing_GBps_of_ascii": nbytes / t_pack / 1e9,
            "host_packing_reads_per_s": nq / t_pack, "exception_queries": int(n_exc.value),
            "device_search_ms_on_packed_input": packed_search_ms,
            "counts_identical_outside_the_exceptions": same_packed and same_dev,
            "pcie_h2d_GBps": h2d, "pcie_d2h_GBps": d2h}


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


def host_threads():
    """CPUs this process may actually use: the affinity mask, cut by the cgroup CPU quota if there is one
    (os.cpu_count() reports the machine, not the container)."""
    n = len(os.sched_getaffinity(0))
    note = f"affinity {n} of {os.cpu_count()} CPUs"
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else int(t.split()[0]) / int(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / 100000.0)):
        try:
            q = parse(open(path).read())
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
        if q is not None:
            note += f", cgroup quota {q:.1f} CPUs"
            n = max(1, min(n, int(q)))
        break
    return n, note


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


if __name__ == "__main__":
    main()
