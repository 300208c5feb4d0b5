"""bench.py's N > 1 path (size exchange, narrow count buffers, double-buffered asynchronous gather, max-over-ranks
timing) on ONE GPU: two ranks share device 0 and talk over gloo (tools/dryrun_two_ranks.sh).  RCCL refuses two ranks on
one device, so the collective library itself is not what this covers; everything around it is."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, GDX_BENCH_ONE_GPU="1", GDX_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "small",
           "--steps", "3", "--no-bandwidth", "--nq", "1000000"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["queries_per_gpu"] == 1_000_000 and d["value"] > 0
    assert d["config"]["gathered_bytes_per_rank_and_step"] > 1_000_000  # 1-byte counts + 8 bytes per hit
    assert d["parity"]["hits_checked"] == d["parity"]["hits_matching_text"] > 0
    assert d["cpu_baseline"] is None  # N = 1 only
    # BASELINE configs[3]: ONE batch sharded over the ranks; rank 0 reran it alone and compared bit for bit
    st = d["strong_scaling"]
    assert st["scaling"] == "strong" and st["queries_total"] == 1_000_000 and st["queries_this_rank"] == 500_000
    assert st["shards_equal_single_rank_output"] == {"counts": True, "hits": True} and st["value"] > 0
