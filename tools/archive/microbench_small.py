#!/usr/bin/env python3
"""Random entry gathers by single lanes (8 / 16 / 32 bytes) against group reads, over arrays of several sizes.
usage: python tools/microbench_small.py [GiB ...]   -> one JSON line"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from genedex_amd import _lib  # noqa: E402
from genedex_amd.device import _ptr, _stream  # noqa: E402

lib = _lib.load()
sizes = [float(x) for x in sys.argv[1:]] or [4.0, 32.0, 96.0]
res = {}
sink = torch.zeros(1, dtype=torch.int32, device="cuda")
n_acc = 1 << 28
for gib in sizes:
    nbytes = int(gib * (1 << 30)) // 4096 * 4096
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    src[: 1 << 30].random_(0, 255)

    def timed(line, mode):
        n_lines = nbytes // line
        if n_lines >= 1 << 32:
            return None
        best = None
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(lib.gdx_bench_random_gather(_ptr(src), n_lines, line, n_acc, 7, mode, _ptr(sink), _stream()))
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None or ms < best else best
        return n_acc / (best / 1e3) / 1e9

    r = {}
    for line in (8, 16, 32):
        r[f"lane{line}_G_per_s"] = timed(line, 3)
    for line in (64, 128):
        r[f"group{line}_G_per_s"] = timed(line, 1)
        r[f"lane{line}_G_per_s"] = timed(line, 0)
    res[f"{gib:g}GiB"] = r
    del src
    torch.cuda.empty_cache()
print(json.dumps(res))
