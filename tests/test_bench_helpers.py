"""CPU checks of bench.py's bookkeeping (no GPU): the roofline is computed from PMC counters the way
MI355X_MICROARCH.md prescribes, the committed fallback summary matches the format bench.py reads, and the JSON line of
the committed final run has the fields the contract asks for with a roofline fraction that is a fraction."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_traffic_from_pmc_counters():
    pmc = {"gdx::search_pair_kernel4<0, 32, 1>": {
        "FETCH_SIZE": {"per_launch": 1000.0, "launches": 3}, "WRITE_SIZE": {"per_launch": 100.0, "launches": 3},
        "TCC_EA0_RDREQ_sum": {"per_launch": 16000.0, "launches": 3}, "TCC_EA0_WRREQ_sum": {"per_launch": 1600.0, "launches": 3},
        "TCC_REQ_sum": {"per_launch": 20000.0, "launches": 3}, "TCC_HIT_sum": {"per_launch": 2000.0, "launches": 3}},
        "gdx::search_pair_stats_kernel4<0, 32, 0>": {"FETCH_SIZE": {"per_launch": 5.0, "launches": 1}},
        "gdx::locate_queue_kernel<gdx::LineTable, false, true>": {"FETCH_SIZE": {"per_launch": 10.0, "launches": 2}}}
    t = bench.traffic_of(pmc, "search_pair_kernel")
    # FETCH_SIZE is in KB and tallies 64 B per 128-byte request on gfx950: doubled; WRITE_SIZE as it is
    assert t["read_bytes"] == 2 * 1000.0 * 1024 and t["write_bytes"] == 100.0 * 1024
    assert t["bytes"] == t["read_bytes"] + t["write_bytes"] and t["read_requests"] == 16000.0
    assert bench.traffic_of(pmc, "locate_queue_kernel") is None  # no WRITE_SIZE pass for it: not half a number
    # two launches that make one step: counters add up
    pmc["gdx::search_fast_kernel4<32>"] = {"FETCH_SIZE": {"per_launch": 500.0, "launches": 3},
                                           "WRITE_SIZE": {"per_launch": 50.0, "launches": 3}}
    both = bench.traffic_of(pmc, "search_fast_kernel|search_pair_kernel")
    assert both["bytes"] == t["bytes"] + 2 * 500.0 * 1024 + 50.0 * 1024 and len(both["by_kernel"]) == 2
    assert bench.traffic_of(None, "x") is None and bench.traffic_of(pmc, "no_such_kernel") is None
    name = ("void gdx::(anonymous namespace)::search_pair_kernel4<0, 32, 1>(gdx::IndexView, unsigned char const*, "
            "unsigned long const*)")
    assert bench.short_kernel_name(name) == "gdx::search_pair_kernel4<0, 32, 1>"


def test_committed_fallback_summary_is_readable():
    args = argparse.Namespace(workload="hg38", lookup_depth=0, path="records")
    aux = {"jump_entry_bytes": 0, "top_table_depth": 0, "seed": {"k": 24}}  # the headline index (bench.py --index seed)
    t, source = bench.committed_traffic(args, 100_000_000, aux, "test")
    assert t is not None and "NOT measured in this run" in source
    assert 1.5e10 < t["bytes"] < 3.0e10 and 1.0 < t["read_requests"] / 1e8 < 2.0  # ~200 bytes, ~1.6 requests per read
    other, why = bench.committed_traffic(args, 100_000_000, {"jump_entry_bytes": 16, "top_table_depth": 16, "seed": {"k": 0}}, "test")
    assert other is None and "another configuration" in why


def test_committed_bench_line_keeps_the_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02", "bench_hg38_final.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u32"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic_source"].startswith("live")
    assert abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] / 1e3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["bit_exact_vs_gpu"] == {"intervals": True, "counts": True, "hits": True}
    assert abs(d["value"] - d["config"]["queries_per_gpu"] / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]
    names = [s["name"] for s in d["secondary"]]
    assert "reference_arrays_only" in names and any("genome_like" in n for n in names) and any("mixed" in n for n in names)
