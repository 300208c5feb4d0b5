"""benchlib.line -- the ONE stdout line (< 4 KB) and the side file with everything else (split out of bench.py in round 6; bench.py re-exports everything)."""
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

__all__ = ['LINE_LIMIT', 'write_side_file', '_pick', '_num', 'compact_line']

LINE_LIMIT = 4096  # bytes: the driver keeps a bounded tail of stdout; round 3's 25 KB line was cut and went unparsed


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
