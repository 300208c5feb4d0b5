#!/usr/bin/env python3
"""Condenses a tools/pmc_passes.sh summary into the committed per-kernel PMC files.

usage: tools/make_pmc_final.py <gpurun_out/<dir>/summary.json> <queries per launch> <op> [out_dir]

Writes <out_dir>/search_pmc_final.json (read by bench.py pmc_traffic), locate_pmc_final.json and
bench_hg38_final_pmc_all.json (every counter of both kernels, per launch).
"""
import json
import os
import sys

summary, nq, op = sys.argv[1], int(sys.argv[2]), sys.argv[3]
out_dir = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r01")
d = json.load(open(summary))


def kernel_counters(pattern, exclude="stats"):
    names = {k for v in d.values() for k in v if pattern in k and exclude not in k}
    if len(names) != 1:
        raise SystemExit(f"expected one kernel matching {pattern!r}, found {sorted(names)}")
    name = names.pop()
    return name, {c: v[name]["per_launch"] for c, v in d.items() if name in v}


search_name, search = kernel_counters("search_pair_kernel")
short = search_name.replace("gdx::", "")
cmd = (f"tools/pmc_passes.sh <dir> --op {op} --nq {nq}  (rocprofv3 --pmc <one group per pass> -- python3 bench.py --steps 1 "
       f"--warmup 0 --no-cpu-baseline --no-bandwidth --secondary-depth 0 --verify-hits 0 --op {op} --nq {nq})")
json.dump({
    "workload": "hg38", "lookup_depth": 0, "kernel": short, "queries_per_launch": nq,
    "FETCH_SIZE_KB_per_launch": search["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": search["WRITE_SIZE"],
    "TCC_EA0_RDREQ_per_launch": search["TCC_EA0_RDREQ_sum"], "TCC_REQ_per_launch": search["TCC_REQ_sum"],
    "TCC_HIT_per_launch": search["TCC_HIT_sum"], "TCC_MISS_per_launch": search["TCC_MISS_sum"],
    "command": cmd,
    "note": "hg38-scale index (n = 3.1e9), a fraction of the 100 M reads per launch; per-query traffic is independent of the "
            "batch size. FETCH_SIZE tallies 64 B per request while every request is 128 B (tools/calibrate_fetch_size.sh, "
            "profiles/r01/fetch_size_calibration.json), hence read bytes = 2 * FETCH_SIZE * 1024.",
}, open(os.path.join(out_dir, "search_pmc_final.json"), "w"), indent=1)
both = {"search": {"kernel": search_name, "queries_per_launch": nq, "per_launch": search}, "command": cmd}
if op != "count":
    locate_name, locate = kernel_counters("locate_queue_kernel")
    json.dump({"kernel": locate_name, "queries_per_launch": nq, "per_launch": locate, "command": cmd},
              open(os.path.join(out_dir, "locate_pmc_final.json"), "w"), indent=1)
    both["locate"] = {"kernel": locate_name, "per_launch": locate}
json.dump(both, open(os.path.join(out_dir, "bench_hg38_final_pmc_all.json"), "w"), indent=1)
print(f"{short}: {search['TCC_EA0_RDREQ_sum'] / nq:.2f} DRAM read requests per query, "
      f"{2 * search['FETCH_SIZE'] * 1024 / nq:.0f} B fetched per query")
