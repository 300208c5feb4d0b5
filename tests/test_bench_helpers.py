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
    args = argparse.Namespace(workload="hg38", lookup_depth=0, path="records", input="ascii")
    aux = {"jump_entry_bytes": 0, "top_table_depth": 14, "seed": {"k": 24}}  # the headline index: the library's default shape
    t, source = bench.committed_traffic(args, 100_000_000, aux, "test")
    assert t is not None and "NOT measured in this run" in source
    assert 1.7e10 < t["bytes"] < 2.2e10 and 1.3 < t["read_requests"] / 1e8 < 1.7  # ~196 bytes, ~1.49 requests per ASCII read
    # (the summary is of the batch as IO symbols + offsets: another form of the batch is another configuration)
    other_form, why = bench.committed_traffic(argparse.Namespace(**{**vars(args), "input": "packed+uniform"}), 100_000_000, aux, "test")
    assert other_form is None and "another configuration" in why
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


def test_committed_round3_line_keeps_the_contract():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r03", "bench_hg38_final.json")).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic_source"].startswith("live")
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] / 1e3) / 1e9) < 1e-6 * r["achieved"]
    assert abs(d["value"] - d["config"]["queries_per_gpu"] / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["bit_exact_vs_gpu"] == {"intervals": True, "counts": True, "hits": True}
    # the few queries that travel beside the 4 bytes per query of an N > 1 gather
    assert d["config"]["compact_exceptions"]["queries"] < d["config"]["queries_per_gpu"] // 1000
    # the kernel's average under rocprofv3 and the HIP events of that same process agree (the bench line's own process is another)
    u = json.loads(open(os.path.join(ROOT, "profiles", "r03", "bench_hg38_final_under_rocprof.json")).read().strip().splitlines()[-1])
    stats = open(os.path.join(ROOT, "profiles", "r03", "bench_hg38_final_kernel_stats.md")).read()
    row = [ln for ln in stats.splitlines() if "search_seed_kernel4<1, false>" in ln][0].split("|")
    assert abs(float(row[4]) - u["kernel_ms"]["search"]) < 0.03 * u["kernel_ms"]["search"]


def test_committed_round4_line_is_what_the_driver_can_read():
    """profiles/r04/bench_hg38_final.json is the stdout line of the driver's command: one line below 4 KB with the contract's
    keys; its roofline fraction follows from its own traffic and time, the profiler's time beside it agrees with the kernel
    statistics kept under profiles/, and the side file holds what the line leaves out."""
    raw = open(os.path.join(ROOT, "profiles", "r04", "bench_hg38_final.json")).read()
    assert raw.count("\n") <= 1 and len(raw.encode()) < 4096
    d = json.loads(raw)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["higher_is_better"] is True and d["dtype"] == "u32"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["input"] == "packed+uniform"
    assert abs(d["value"] - d["config"]["queries_per_gpu"] / (d["ms_per_step"] / 1e3)) < 1e-4 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["traffic_source"].startswith("live")
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["traffic"] / (r["avg_launch_ms"] / 1e3) / 1e9 / r["peak"]) < 1e-4
    assert abs(r["frac_rocprof"] - r["traffic"] / (r["avg_launch_ms_rocprof"] / 1e3) / 1e9 / r["peak"]) < 1e-4
    # the two clocks are two processes: they agree within the process-to-process spread (3-8 %)
    assert abs(r["avg_launch_ms_rocprof"] - r["avg_launch_ms"]) < 0.08 * r["avg_launch_ms"]
    assert r["dram_read_requests_per_query"] < 1.25 and r["frac_algorithmic"] > 1.0
    assert 0.25 < r["reference_layout"]["frac_algorithmic"] < 0.40 and r["reference_layout"]["frac_traffic"] > 0.8
    # the kernel's average in the child pass's statistics (kept under profiles/) is what avg_launch_ms_rocprof is made of
    stats = open(os.path.join(ROOT, "profiles", "r04", "bench_child_kernel_stats.md")).read()
    row = [ln for ln in stats.splitlines() if "search_seed_lane_kernel<2, true>" in ln][0].split("|")
    assert 0.95 * r["avg_launch_ms_rocprof"] < float(row[4]) <= r["avg_launch_ms_rocprof"]  # (+ its list kernels)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["bit_exact_vs_gpu"] == {"intervals": True, "counts": True, "hits": True}
    assert d["ascii_input"]["offsets_and_hits_identical_to_headline"] is True and d["ascii_input"]["value"] < d["value"]
    side = json.load(open(os.path.join(ROOT, "profiles", "r04", "bench_secondary.json")))
    names = [s_["name"] for s_ in side["secondary"]]
    assert "reference_arrays_only" in names and any("genome_like" in n for n in names) and any("mixed" in n for n in names)
    assert abs(side["value"] - d["value"]) < 1e-5 * side["value"] and "measured_bandwidth" in side


class _FakeIndex:
    def __init__(self, starts):
        self.starts = starts

    def num_texts(self):
        return len(self.starts) - 1


class _FakeEngine:
    """the two kernels of the compact gather restated with numpy (gdx_compact_exceptions_dev, gdx_compact_split_hits_dev)"""

    def __init__(self, starts):
        import numpy as np

        self.np, self.starts, self.index = np, np.asarray(starts, dtype=np.int64), _FakeIndex(starts)

    def compact_exceptions(self, words, nq, queries, n):
        np = self.np
        idx = np.flatnonzero(words[:nq].numpy() == -2)[::-1].copy()  # (in no particular order)
        n[0] = len(idx)
        k = min(len(idx), queries.numel())
        queries[:k] = bench_torch().from_numpy(idx[:k].astype(np.int32))

    def compact_split_hits(self, words, nq, ids, pos):
        np = self.np
        w = words[:nq].numpy().astype(np.int64) & 0xFFFFFFFF
        hit = w < 0xFFFFFFFE
        tid = np.searchsorted(self.starts[1:] - 1, np.where(hit, w, 0), side="left")
        ids[:nq] = bench_torch().from_numpy(np.where(hit, tid, 0).astype(np.uint8))
        inside = np.where(hit, w - self.starts[np.minimum(tid, len(self.starts) - 2)], np.where(w == 0xFFFFFFFF, -1, -2))
        pos[:nq] = bench_torch().from_numpy(inside.astype(np.int32))


    # the found-bitmap wire (gdx_wire_pack_dev / gdx_wire_split_dev) through the library's own tensor restatements
    def wire_pack_workspace_bytes(self, nq):
        return 16

    def wire_pack(self, compact, hit_offsets, hits, nq, v, workspace):
        from genedex_amd import dist as gdist

        gdist.wire_pack_reference(compact, hit_offsets, hits, nq, v)

    def wire_split(self, v, nq, ids, pos):
        from genedex_amd import dist as gdist

        i, p = gdist.wire_split_reference(v, nq, bench_torch().from_numpy(self.starts))
        ids[:nq], pos[:nq] = i, p


def bench_torch():
    import torch

    return torch


def test_compact_gather_plumbing_with_a_restated_engine():
    """bench.make_gather -> make_compact_gather -> pack -> gathered_shards at world size 1 on CPU tensors: what travels
    expands to exactly the step's counts and hits, and the wire is chosen by its size."""
    import numpy as np
    import torch

    from genedex_amd import dist as gdist

    rng = np.random.default_rng(5)
    text_lens = [700, 50, 1200]
    starts = np.concatenate([[0], np.cumsum([n + 1 for n in text_lens])])
    nq = 4000
    counts = rng.choice([0, 1, 3], size=nq, p=[0.1, 0.88, 0.02])
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    tid = rng.integers(0, 3, int(off[-1]))
    pos = np.array([rng.integers(0, text_lens[t]) for t in tid], dtype=np.int64)
    words = np.full(nq, -2, dtype=np.int32)
    words[counts == 0] = -1
    single = np.flatnonzero((counts == 1) & (np.arange(nq) % 97 != 0))
    words[single] = (starts[tid[off[single]]] + pos[off[single]]).astype(np.int32)
    hits = torch.from_numpy(np.stack([tid, pos], axis=1).astype(np.int32))

    class Runner:
        pass

    r = Runner()
    r.nq, r.n_slots, r.use_compact, r.use_rec, r.total_hits = nq, 2, True, True, int(off[-1])
    r.eng = _FakeEngine(starts)
    r.outs = [{"compact": torch.from_numpy(words.copy()), "hit_offsets": torch.from_numpy(off), "rec": None} for _ in range(2)]
    r.hits = [hits.clone(), hits.clone()]
    r.counts = lambda o: torch.from_numpy(counts.astype(np.int32))
    n_exc = int((words == -2).sum())
    n_found = int((words >= 0).sum())
    os.environ.pop("GDX_BENCH_GATHER", None)
    # most reads are found: a bit per read + 4 bytes per found read is the smallest form
    gather, pack, nbytes = bench.make_gather(torch, gdist, r, torch.device("cpu"), True)
    n_exc_hits = int(counts[words == -2].sum())
    assert getattr(gather, "wire_name", "") == "bitmap" and nbytes < 4 * nq + 4 * n_exc + 5 * n_exc_hits + 8
    assert gather.exceptions == {"queries": n_exc, "hits": int(counts[words == -2].sum()), "found": n_found}
    assert len(gather.slots[0]) == 1 and gather.slots[0][0].numel() == nbytes  # ONE buffer per rank and step
    for slot in (0, 1):
        pack(slot)
        gather.submit(slot)
    gather.drain()
    cnt, hh = bench.gathered_shards(torch, gdist, gather, 1, [(0, nq)], [r.total_hits], True)
    assert cnt.tolist() == counts.tolist() and hh.tolist() == hits.tolist()
    # the compact words themselves (GDX_BENCH_GATHER=compact; what a batch with few found reads would choose)
    os.environ["GDX_BENCH_GATHER"] = "compact"
    try:
        gather, pack, nbytes = bench.make_gather(torch, gdist, r, torch.device("cpu"), True)
    finally:
        os.environ.pop("GDX_BENCH_GATHER", None)
    assert getattr(gather, "compact_wire", False) and getattr(gather, "wire_name", "compact") == "compact"
    assert nbytes == 4 * nq + 4 * n_exc + 5 * gather.exceptions["hits"] + 8
    assert r.outs[0]["compact"] is gather.slots[0][0]  # the search writes into the buffer that travels
    for slot in (0, 1):
        pack(slot)
        gather.submit(slot)
    gather.drain()
    cnt, hh = bench.gathered_shards(torch, gdist, gather, 1, [(0, nq)], [r.total_hits], True)
    assert cnt.tolist() == counts.tolist() and hh.tolist() == hits.tolist()
    # a batch of repeats (every query an exception) travels as arrays: fewer bytes
    r2 = Runner()
    r2.__dict__.update(r.__dict__)
    r2.outs = [{"compact": torch.full((nq,), -2, dtype=torch.int32), "hit_offsets": torch.from_numpy(off), "rec": None}
               for _ in range(2)]
    r2.hits = [hits.clone(), hits.clone()]
    r2.use_rec = False
    g2, count_of, _ = bench.make_gather(torch, gdist, r2, torch.device("cpu"), True)
    assert not getattr(g2, "compact_wire", False) and g2.hits_are_split


def _round3_result(n_gpus=1):
    d = json.loads(open(os.path.join(ROOT, "profiles", "r03", "bench_hg38_final.json")).read().strip().splitlines()[-1])
    d["n_gpus"] = n_gpus
    return d


def test_the_stdout_line_stays_below_4_kb_whatever_was_measured(tmp_path):
    """Round 3's line carried every secondary (25 KB) and the driver could not read it back: the line bench.py prints now
    holds the contract's keys, roofline and cpu_baseline only; everything else goes to the side file."""
    full = _round3_result()
    assert len(json.dumps(full)) > 20_000  # the round-3 line as it was
    side = bench.write_side_file(str(tmp_path / "side" / "bench_secondary.json"), full)
    assert side and json.load(open(side))["secondary"]  # nothing is lost: the side file has it all
    text = json.dumps(bench.compact_line(full, side))
    assert len(text.encode()) < bench.LINE_LIMIT == 4096 and "\n" not in text
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert "secondary" not in line and "measured_bandwidth" not in line and line["side_file"] == side
    assert "workload" in line["config"] and "model" not in line["config"] and "aux_structures" not in line["config"]
    assert abs(line["value"] - full["value"]) < 1e-5 * full["value"]
    r = line["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "frac_algorithmic",
                "wasted_traffic_ratio"):
        assert key in r, key
    assert abs(r["frac"] - r["traffic"] / (r["avg_launch_ms"] / 1e3) / 1e9 / r["peak"]) < 1e-4
    assert set(r["reference_layout"]) >= {"frac_traffic", "frac_algorithmic", "value"}
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and "thread_sweep_qps" not in c
    # a line that nothing could shorten enough still fits: strings are cut, optional parts go
    bloated = _round3_result()
    bloated["config"]["workload"] = "w" * 50_000
    bloated["roofline"]["traffic_source"] = "t" * 50_000
    bloated["roofline"]["kernel"] = "k" * 300
    bloated["cpu_baseline"]["sample"] = "s" * 50_000
    bloated["parity"] = {k: 1 for k in ("queries_with_status", "queries_found", "sum_of_counts_equals_hits", "hits_checked")}
    assert len(json.dumps(bench.compact_line(bloated, "x" * 100)).encode()) < bench.LINE_LIMIT
    # nothing measured (no PMC, no CPU baseline): the contract's keys are still there, as nulls
    bare = _round3_result()
    bare["roofline"] = {"bound": "hbm", "kernel": "k", "unit": "GB/s", "peak": 8000.0, "avg_launch_ms": 3.0, "traffic": None,
                        "achieved": None, "frac": None}
    bare["cpu_baseline"] = None
    line = bench.compact_line(bare, None)
    assert line["roofline"]["traffic"] is None and line["roofline"]["frac"] is None and line["cpu_baseline"] is None


def test_at_n_gpus_above_one_the_line_reports_the_sharded_batch():
    """BASELINE.json configs[3] is ONE 100 M batch sharded over the ranks: at N > 1 `value` is that measurement, the
    every-rank-its-own-batch number moves to `weak_scaling`."""
    full = _round3_result(n_gpus=8)
    weak_value, weak_ms = full["value"], full["ms_per_step"]
    full["config"]["gathered_bytes_per_rank_and_step"] = 400_000_000
    full["strong_scaling"] = {"scaling": "strong", "value": 1.2e11, "unit": "queries/s", "ms_per_step": 0.83, "queries_total": 100_000_000,
                              "queries_this_rank": 12_500_000, "steps": 20, "gathered_bytes_per_rank_and_step": 50_000_008,
                              "gather_wire": "compact", "shards_equal_single_rank_output": {"counts": True, "hits": True}}
    bench.report_strong_scaling(full, bench.WORKLOADS["hg38"])
    line = json.loads(json.dumps(bench.compact_line(full, None)))
    assert len(json.dumps(line).encode()) < bench.LINE_LIMIT
    assert line["scaling"] == "strong" and line["value"] == 1.2e11 and line["ms_per_step"] == 0.83 and line["n_gpus"] == 8
    assert "ONE batch of 100000000 reads sharded over 8 GPUs" in line["config"]["workload"]
    assert line["config"]["queries_total"] == 100_000_000 and line["config"]["queries_per_gpu"] == 12_500_000
    assert line["config"]["gather_wire"] == "compact" and line["config"]["gathered_bytes_per_rank_and_step"] == 50_000_008
    assert abs(line["weak_scaling"]["value"] - weak_value) < 1e-5 * weak_value and line["weak_scaling"]["ms_per_step"] == float(f"{weak_ms:.6g}")
    assert line["parity"]["shards_equal_single_rank_output"] == {"counts": True, "hits": True}


def test_kernel_stats_of_the_trace_child_pass(tmp_path):
    p = tmp_path / "k_kernel_stats.csv"
    p.write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                 '"void gdx::(anonymous namespace)::search_seed_kernel4<1, false>(gdx::IndexView)",13,52000000,4000000,50,1,2,3\n'
                 '"void gdx::(anonymous namespace)::search_verify_kernel4<1, true>(gdx::IndexView)",13,1300000,100000,5,1,2,3\n'
                 '"void rocprim::detail::something()",5,100,20,0,1,2,3\n')
    out = {"gdx::search_seed_kernel4<1, false>": {"FETCH_SIZE": {"per_launch": 1.0, "launches": 3}}}
    bench.read_kernel_stats(str(p), out, keep=str(tmp_path / "kept" / "stats.csv"))
    assert out["gdx::search_seed_kernel4<1, false>"]["rocprof"] == {"avg_ms": 4.0, "launches": 13}
    assert "FETCH_SIZE" in out["gdx::search_seed_kernel4<1, false>"] and len(out) == 2
    assert (tmp_path / "kept" / "stats.csv").exists()
    assert abs(bench.rocprof_ms_of(out, "search_seed_kernel|search_verify_kernel|search_kernel") - 4.1) < 1e-9
    assert bench.rocprof_ms_of(out, "locate_queue_kernel") is None and bench.rocprof_ms_of(None, "x") is None


def test_committed_round5_line_is_what_the_driver_can_read():
    """profiles/r05/bench_hg38_final.json: the stdout line of the driver's command on round 5's final tree -- below 4 KB, the
    contract's keys, a roofline fraction that follows from its own traffic and time, the profiler's time beside it in
    agreement with the kernel statistics kept under profiles/, the shard steps, the input form by name, and the side file
    with what the line leaves out (incl. the narrow host call checked against the wide one)."""
    raw = open(os.path.join(ROOT, "profiles", "r05", "bench_hg38_final.json")).read()
    assert raw.count("\n") <= 1 and len(raw.encode()) < 4096
    d = json.loads(raw)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "input_form", "shard_step_ms"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "u32" and d["config"]["input"] == "packed+uniform"
    assert abs(d["value"] - d["config"]["queries_per_gpu"] / (d["ms_per_step"] / 1e3)) < 1e-4 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["traffic_source"].startswith("live")
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["traffic"] / (r["avg_launch_ms"] / 1e3) / 1e9 / r["peak"]) < 1e-4
    assert abs(r["avg_launch_ms_rocprof"] - r["avg_launch_ms"]) < 0.08 * r["avg_launch_ms"]
    assert r["dram_read_requests_per_query"] < 1.2 and "search_seed_lane_kernel<2, true>" in r["kernel"]
    for rung in ("reference_layout", "reference_layout_d10", "reference_layout_d13"):
        assert 0.25 < r[rung]["frac_algorithmic"] < 0.40, rung
    stats = open(os.path.join(ROOT, "profiles", "r05", "bench_child_kernel_stats.md")).read()
    row = [ln for ln in stats.splitlines() if "search_seed_lane_kernel<2, true>" in ln][0].split("|")
    assert 0.95 * r["avg_launch_ms_rocprof"] < float(row[4]) <= r["avg_launch_ms_rocprof"]  # (+ its list kernels)
    # the same command under rocprofv3 --kernel-trace --stats: the kernel's average there agrees too
    stats = open(os.path.join(ROOT, "profiles", "r05", "bench_hg38_final_kernel_stats.md")).read()
    row = [ln for ln in stats.splitlines() if "search_seed_lane_kernel<2, true>" in ln][0].split("|")
    assert abs(float(row[4]) - r["avg_launch_ms"]) < 0.08 * r["avg_launch_ms"]
    assert float(d["shard_step_ms"]["12500000"]) <= 0.50  # what a rank of eight runs of the sharded batch
    assert d["locate_roofline"]["frac"] >= 0.55
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["bit_exact_vs_gpu"] == {"intervals": True, "counts": True, "hits": True}
    assert d["ascii_input"]["offsets_and_hits_identical_to_headline"] is True and d["ascii_input"]["value"] < d["value"]
    side = json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_secondary.json")))
    names = [s_["name"] for s_ in side["secondary"]]
    assert "reference_arrays_d10" in names and "reference_arrays_d13" in names and any("genome_like" in n for n in names)
    assert abs(side["value"] - d["value"]) < 1e-5 * side["value"]
    pu = side["end_to_end"]["packed_uniform"]
    assert pu["results_identical_to_device_path"] == {"counts": True, "hits_total": True, "narrow_equals_wide": True}
    assert side["end_to_end"]["fastq_to_hits"]["hits_identical_to_device_path"] is True


def test_committed_round6_line_is_what_the_driver_can_read():
    """profiles/r06/bench_hg38_final.json: the stdout line of the driver's command on round 6's tree.  `value` is timed on the
    reference's own input form (IO symbols + u64 offsets: the alphabet translation inside the timed region) on the index a
    caller of gdx_index_build gets with every option at its default, the pre-translated form is beside it, the 8(d) number of
    the headline carries its label, and the cursor / exact-interval numbers of the SAME index are in the line."""
    raw = open(os.path.join(ROOT, "profiles", "r06", "bench_hg38_final.json")).read()
    assert raw.count("\n") <= 1 and len(raw.encode()) < 4096
    d = json.loads(raw)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "input_form", "packed_input"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "u32"
    assert d["config"]["input"] == "ascii" and d["input_form"] == "IO symbols + u64 offsets"
    assert d["config"]["index_is_library_default"] is True and d["config"]["index_gb_per_replica"] < 135
    assert abs(d["value"] - d["config"]["queries_per_gpu"] / (d["ms_per_step"] / 1e3)) < 1e-4 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["traffic_source"].startswith("live")
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["traffic"] / (r["avg_launch_ms"] / 1e3) / 1e9 / r["peak"]) < 1e-4
    assert abs(r["avg_launch_ms_rocprof"] - r["avg_launch_ms"]) < 0.08 * r["avg_launch_ms"]
    assert r["frac_section8d_headline"] > 1.0 and "algorithm substituted" in r["frac_section8d_headline_label"]
    assert "search_seed_lane_kernel<1, false>" in r["kernel"]
    assert 0.25 < r["reference_layout"]["frac_algorithmic"] < 0.40
    stats = open(os.path.join(ROOT, "profiles", "r06", "bench_hg38_final_kernel_stats.md")).read()
    row = [ln for ln in stats.splitlines() if "search_seed_lane_kernel<1, false>" in ln][0].split("|")
    assert abs(float(row[4]) - r["avg_launch_ms"]) < 0.08 * r["avg_launch_ms"]
    p = d["packed_input"]
    assert p["offsets_and_hits_identical_to_headline"] is True and p["value"] > d["value"] and p["input"] == "packed+uniform"
    cur = d["cursor_api_50M_len20_150"]
    assert cur["headline_index"]["index_gb"] == round(d["config"]["index_gb_per_replica"])
    assert cur["headline_index"]["cursor_api_ms"] < 16.0 and cur["headline_index"]["fused_ms"] < 10.0
    assert cur["exact_intervals_100M_len50_headline_index_ms"] < 10.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["bit_exact_vs_gpu"] == {"intervals": True, "counts": True, "hits": True}
    side = json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_secondary.json")))
    names = [s_["name"] for s_ in side["secondary"]]
    assert any(n.startswith("cfg2_256MB_10M_len50_count") for n in names) and any("genome_like" in n for n in names)
    assert any("on the HEADLINE index" in n and n.startswith("mixed_lengths") for n in names)
    g = [s_ for s_ in side["secondary"] if "genome_like" in s_["name"]][0]
    assert g["aux_structures"]["default_shape"] is True and g["oracle_gate"]["hits_identical"] is True
    assert side["config"]["aux_structures"]["default_shape"] is True
    assert side["end_to_end"]["fastq_to_hits"]["hits_identical_to_device_path"] is True
