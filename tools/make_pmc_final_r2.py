#!/usr/bin/env python3
"""profiles/rNN/search_pmc_final.json from a bench.py result whose roofline was measured live (since round 4: the side
file, gpurun_out/bench_secondary.json -- the stdout line carries the short form only): the fallback bench.py uses (and
labels as such) when rocprofv3 is not available in a later run.
usage: tools/make_pmc_final_r2.py <bench_secondary.json> [out]"""
import json
import os
import sys

d = json.load(open(sys.argv[1]))
r, c = d["roofline"], d["config"]
assert r["traffic_source"].startswith("live"), "the bench line has no live PMC traffic"
out = {"workload": c["name"], "lookup_depth": c["lookup_depth"], "path": c["path"],
       "jump_entry_bytes": c["aux_structures"]["jump_entry_bytes"], "top_table_depth": c["aux_structures"]["top_table_depth"],
       "seed_k": c["aux_structures"].get("seed", {}).get("k", 0), "input": c.get("input", "ascii"),
       "kernel": r["kernel"], "queries_per_launch": c["queries_per_gpu"],
       "read_bytes_per_launch": r["traffic_read_bytes"], "write_bytes_per_launch": r["traffic_write_bytes"],
       "read_requests_per_launch": r["dram_read_requests_per_launch"],
       "write_requests_per_launch": r["dram_write_requests_per_launch"],
       "l2_requests_per_launch": r["l2_requests_per_launch"], "l2_hits_per_launch": r["l2_hits_per_launch"],
       "avg_launch_ms_in_that_run": r["avg_launch_ms"],
       "source": "bench.py live PMC passes (rocprofv3 --pmc, FETCH_SIZE x 2 + WRITE_SIZE, separate passes, full 100 M batch)"}
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r02", "search_pmc_final.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out))
