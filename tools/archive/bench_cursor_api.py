#!/usr/bin/env python3
"""BASELINE workload 5 through the batched cursor API: cursor_empty + repeated gdx_cursor_extend_front_many_dev
(one launch per query position, finished cursors compacted away between launches) versus the fused
cursors_for_many_queries kernel.  Both must give identical intervals.  Prints one JSON object."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from genedex_amd import _lib, alphabet  # noqa: E402
from genedex_amd.device import (DeviceEngine, DeviceQueries, _ptr, _stream, build_index_from_device_text,  # noqa: E402
                                hg38_text_lengths, synth_text)

ap = argparse.ArgumentParser()
ap.add_argument("--total", type=int, default=3_100_000_000)
ap.add_argument("--nq", type=int, default=50_000_000)
ap.add_argument("--len-min", type=int, default=20)
ap.add_argument("--len-max", type=int, default=150)
ap.add_argument("--sampled-ppm", type=int, default=700_000)
args = ap.parse_args()

dev = torch.device("cuda", 0)
lib = _lib.load()
io_text = synth_text(args.total, device=dev)
lengths = hg38_text_lengths(args.total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32")
q = DeviceQueries.synth(io_text, lengths, args.nq, args.len_min, args.len_max, args.sampled_ppm, seed=43)
eng = DeviceEngine(index)
out = eng.alloc_outputs(args.nq)
n = index.total_text_len()

# fused path
eng.search(q, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.search(q, out)
torch.cuda.synchronize()
t_fused = time.perf_counter() - t0

# cursor API path
qlen = (q.qoff[1:] - q.qoff[:-1])
qend = q.qoff[1:]
final_s = torch.zeros(args.nq, dtype=torch.int32, device=dev)
final_e = torch.full((args.nq,), n if n < 2**31 else n - 2**32, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
qid = torch.arange(args.nq, device=dev)
keep = qlen > 0
qid = qid[keep]
cs = final_s[qid].clone()
ce = final_e[qid].clone()
launches, lanes = 0, 0
step = 0
while qid.numel() > 0:
    syms = q.qbuf[qend[qid] - 1 - step]
    status = torch.empty(qid.numel(), dtype=torch.uint8, device=dev)
    _lib.check(lib.gdx_cursor_extend_front_many_dev(eng.h, _ptr(cs), _ptr(ce), _ptr(syms), qid.numel(), _ptr(status),
                                                    _stream()))
    launches += 1
    lanes += qid.numel()
    step += 1
    alive = (cs != ce) & (qlen[qid] > step)
    done = ~alive
    final_s[qid[done]] = cs[done]
    final_e[qid[done]] = ce[done]
    qid, cs, ce = qid[alive], cs[alive], ce[alive]
torch.cuda.synchronize()
t_cursor = time.perf_counter() - t0
same = bool(torch.equal(final_s, out["start"]) and torch.equal(final_e, out["end"]))
print(json.dumps({"workload": f"{args.nq} reads of length {args.len_min}..{args.len_max}, {args.sampled_ppm / 1e4:.0f}% sampled",
                  "fused_cursors_for_many_queries_ms": t_fused * 1e3, "fused_queries_per_s": args.nq / t_fused,
                  "cursor_api_extend_front_many_ms": t_cursor * 1e3, "cursor_api_queries_per_s": args.nq / t_cursor,
                  "extend_launches": launches, "cursor_steps": lanes, "intervals_identical": same}))
if not same:
    sys.exit(1)
