#!/usr/bin/env python3
"""The headline step (count + locate on the 74 GB seed index, reads as 2-bit codes of uniform length) on the batch sizes a
rank of 1 / 2 / 4 / 8 GPUs gets of ONE 100 M-read batch: 100 M, 50 M, 25 M, 12.5 M reads.  Wall-clock and HIP-event times per
step; under `rocprofv3 --kernel-trace` the trace shows the launches of the small step and the gaps between them
(tools/trace_timeline.py).  usage: python tools/exp_shard_step.py [steps] [sizes, comma separated]  -> one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import (DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths,  # noqa: E402
                                synth_text)
from genedex_amd.index import build_options  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sizes = [int(s) for s in sys.argv[2].split(",")] if len(sys.argv) > 2 else [100_000_000, 50_000_000, 25_000_000, 12_500_000]
total = int(os.environ.get("GDX_EXP_TOTAL", 3_100_000_000))
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32",
                                     options=build_options(**dict(bench.SEED_INDEX, **({"seed_load_percent": int(os.environ["GDX_EXP_SEED_LOAD"])}
                                                                                        if os.environ.get("GDX_EXP_SEED_LOAD") else {}))))
eng = DeviceEngine(index)
full = DeviceQueries.synth(io_text, lengths, max(sizes), 50, 50, 900_000, seed=43)
res = {"steps": steps, "index_gb": index.info.device_bytes / 1e9, "seed": index.seed_info(), "shard_step": {}}
for nq in sizes:
    q = full.copy_slice(0, nq).as_packed(index).as_uniform(50)
    per = {}
    ref = None
    for mode in ("split", "fused"):
        runner = bench.StepRunner(torch, eng, q, nq, True, "records")
        runner.step_mode = mode
        runner.size()
        for _ in range(3):
            runner.step(0, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.step(0, True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        runner.check_totals()
        runner.widen_offsets()
        got = (runner.outs[0]["hit_offsets"].clone(), runner.hits[0][: runner.total_hits].clone())
        if ref is None:
            ref = got
        elif not (torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])):
            raise SystemExit(f"PARITY FAILURE: the {mode} step gives other offsets or hits than the split step ({nq} reads)")
        per[mode] = {"ms_per_step": ms, "search_ms": runner.mean_ms(runner.ev_search), "locate_ms": runner.mean_ms(runner.ev_locate),
                     "hits": runner.total_hits, "Gq_per_s": nq / ms / 1e6}
        if mode == "fused":
            # the step once more without events (an event record between two kernels costs ~5 us of idle queue)
            t0 = time.perf_counter()
            for _ in range(steps):
                runner.step(0, False)
            torch.cuda.synchronize()
            per[mode]["ms_per_step_no_events"] = (time.perf_counter() - t0) / steps * 1e3
            # what a rank adds for the gather (gdx_wire_pack_dev) and what the root does per received shard (gdx_wire_split_dev),
            # beside the round-4 wire (compact words: gdx_compact_exceptions_dev on the rank, gdx_compact_split_hits_dev on the root)
            from genedex_amd import dist as gdist
            o = runner.outs[0]
            n_exc, n_exc_hits = gdist.exception_sizes(o["compact"], o["hit_offsets"], nq)
            n_found = int((o["compact"][:nq] >= 0).sum().item()) + int((o["compact"][:nq] < -2).sum().item())
            layout = gdist.WireLayout(nq, n_found, max(n_exc, 1), max(n_exc_hits, 1))
            buf = torch.zeros(layout.nbytes, dtype=torch.uint8, device=dev)
            v = layout.views(buf)
            wws = torch.empty(max(eng.wire_pack_workspace_bytes(nq), 16), dtype=torch.uint8, device=dev)
            ids = torch.empty(nq, dtype=torch.uint8, device=dev)
            pos = torch.empty(nq, dtype=torch.int32, device=dev)
            ids2, pos2 = torch.empty_like(ids), torch.empty_like(pos)
            listed, n_listed = torch.zeros(max(n_exc, 1), dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)

            def timed(fn, reps=10):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps * 1e3

            wire = {"bytes": layout.nbytes, "bytes_per_read": layout.nbytes / nq, "payload_bytes_per_read":
                    layout.payload_bytes(nq, n_found, n_exc, n_exc_hits) / nq, "found": n_found, "exceptions": n_exc,
                    "exception_hits": n_exc_hits,
                    "pack_ms": timed(lambda: eng.wire_pack(o["compact"], o["hit_offsets"], runner.hits[0], nq, v, wws)),
                    "split_ms": timed(lambda: eng.wire_split(v, nq, ids, pos)),
                    "compact_wire_bytes_per_read": (4 * nq + 4 * n_exc + 5 * n_exc_hits + 8) / nq,
                    "compact_exceptions_ms": timed(lambda: eng.compact_exceptions(o["compact"], nq, listed, n_listed)),
                    "compact_split_ms": timed(lambda: eng.compact_split_hits(o["compact"], nq, ids2, pos2))}
            if not (torch.equal(ids, ids2) and torch.equal(pos, pos2)):
                raise SystemExit("PARITY FAILURE: the bitmap wire and the compact words split into different results")
            per["wire"] = wire
            del buf, v, wws, ids, pos, ids2, pos2
        del runner
        torch.cuda.empty_cache()
    res["shard_step"][str(nq)] = per
    del q, ref, got
print(json.dumps(res))
