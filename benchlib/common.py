"""benchlib.common -- what every part of bench.py shares: the workloads, the index shapes, the step runner (one pass of the hot path over one batch) (split out of bench.py in round 6; bench.py re-exports everything)."""
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


__all__ = ['WORKLOADS', 'INPUT_FORMS', 'HBM_PEAK_GBPS', 'log', 'SEED_INDEX', 'REFERENCE_ARRAYS', 'LOOKUP_RUNGS', 'LOOKUP_PMC_READS', 'FULL_INDEX', 'workload_of', 'input_form', 'build_options_of', 'apply_query_options', '_null', 'StepRunner', 'timed_steps', 'time_config', 'host_threads']

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


# (seed_load_percent: slots of the seed table filled on average.  The library's default of 70 leaves 15 % of the buckets overflowing
# into their neighbours -- a second 128-byte fetch for the reads that land there; at 60 it is 6.6 %: 9 GB more of the 288, the
# step 4.5 % shorter on 100 M reads and 8 % on the 12.5 M a rank of eight runs (profiles/r05/seed_load_sweep.txt))
SEED_INDEX = dict(pair_lines=False, jump_entry_bytes=0, top_table_depth=0, full_suffix_array=True, seed_symbols=True, seed_load_percent=60)


REFERENCE_ARRAYS = dict(pair_lines=False, jump_entry_bytes=0, top_table_depth=0)  # the reference's information content, nothing else


LOOKUP_RUNGS = (10, 13)        # lookup-table depths of the `reference_arrays_dD` secondaries (lookup_table.rs:51-161)


LOOKUP_PMC_READS = 20_000_000  # reads of their PMC child passes


FULL_INDEX = dict(seed_symbols=True, inverse_suffix_array=True, aux_budget_bytes=250_000_000_000)


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
        # --op count: FmIndex::count_many (lib.rs:155-161) = gdx_count_many[_layout]_dev -- counts, no intervals, no hits (until
        # round 6 the count-only step called the exact-interval search, which packed reads do not reach the slim kernels of)
        self.count_only = not do_locate
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
        if self.count_only:
            o["counts"] = self.torch.zeros(max(self.nq, 1), dtype=self.torch.int32, device=o["status"].device)
        if self.use_rec:
            o["rec"] = self.eng.alloc_records(self.nq)
            o["compact"] = self.eng.alloc_compact(self.nq) if self.use_compact else None
        return o

    def search(self, o):
        if self.count_only:
            self.eng.count(self.q, o["counts"], o["status"])
        elif self.use_rec:
            self.eng.locate_search(self.q, o["rec"], compact=o["compact"])
        else:
            self.eng.search(self.q, o)

    def offsets(self, o):
        if self.count_only:
            return
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
        if self.count_only:
            return o["counts"][: self.nq]
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
        if self.count_only:
            self.total_hits = int(o["counts"][: self.nq].to(torch.int64).sum().item())
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
        self.fused_slots = getattr(self, "fused_slots", set()) | {slot}
        if after is not None:
            after(slot)

    def check_totals(self):
        """after the timed steps of the fused form: every slot's hit total must have fitted the buffer it was offered"""
        if not getattr(self, "fused_steps", 0):
            return
        self.torch.cuda.synchronize()
        for slot, (t, h) in enumerate(zip(self.totals, self.hits)):
            if slot not in getattr(self, "fused_slots", set()):  # (a run of one step leaves the second slot's totals untouched)
                continue
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
