#!/usr/bin/env python3
"""What the result stores of the search kernel cost: the same kernel with all / some / no output pointers.
Plain run: prints ms per variant.  Under `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -- python3 tools/exp_store_cost.py 10000000 1`
the dispatches of the search kernel appear in the order of VARIANTS (one launch each)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from genedex_amd import _lib, alphabet  # noqa: E402
from genedex_amd.device import (DeviceEngine, DeviceQueries, _ptr, _stream, build_index_from_device_text,  # noqa: E402
                                hg38_text_lengths, synth_text)

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
total = 3_100_000_000
text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
ix = build_index_from_device_text(text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32")
q = DeviceQueries.synth(text, lengths, nq, 50, 50, 900_000, seed=43)
eng = DeviceEngine(ix)
out = eng.alloc_outputs(nq, hint=True)
lib = eng.lib
P = _ptr


def call(start, end, status, hint):
    if hint is not None:
        return lambda: _lib.check(lib.gdx_cursors_for_many_queries_hint_dev(eng.h, P(q.qbuf), P(q.qoff), q.nq, start, end, status,
                                                                            hint, _stream()))
    return lambda: _lib.check(lib.gdx_cursors_for_many_queries_dev(eng.h, P(q.qbuf), P(q.qoff), q.nq, start, end, status,
                                                                   _stream()))


VARIANTS = [("no outputs", call(None, None, None, None)),
            ("start", call(P(out["start"]), None, None, None)),
            ("start+end", call(P(out["start"]), P(out["end"]), None, None)),
            ("start+end+status", call(P(out["start"]), P(out["end"]), P(out["status"]), None)),
            ("start+end+status+hint", call(P(out["start"]), P(out["end"]), P(out["status"]), P(out["hint"]))),
            ("hint only", call(None, None, None, P(out["hint"]))),
            ("status only", call(None, None, P(out["status"]), None))]
for name, fn in VARIANTS:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if reps > 1:
        fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / reps:.2f} ms", flush=True)
