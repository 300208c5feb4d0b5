"""benchlib.pmc -- roofline inputs: rocprofv3 child passes of the same workload (PMC counters, kernel durations) and what is read out of them (split out of bench.py in round 6; bench.py re-exports everything)."""
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

__all__ = ['PMC_PASSES', 'KERNEL_REGEX', 'pmc_child', 'read_kernel_stats', 'rocprof_ms_of', 'run_live_pmc', 'short_kernel_name', 'traffic_of', 'traffic_requests_of', 'committed_traffic']

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


def committed_traffic(args, nq, aux, why):
    """Fallback when the live PMC passes are unavailable: the committed summary of the same configuration."""
    path = os.path.join(ROOT, "profiles", "r06", "search_pmc_final.json")
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
