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
@pytest.mark.parametrize("wire", ["bitmap", "compact", "arrays"])
def test_bench_two_ranks_on_one_gpu(wire):
    """wire: what travels to rank 0 -- a bit per read + the positions of the found reads + the exceptions (gdx_wire_pack_dev /
    gdx_wire_split_dev; what an index with a seed table chooses when most reads are found), the search's compact results +
    the exceptions' counts and hits (rank 0 splits them into text id + position with gdx_compact_split_hits_dev), or count
    bytes + text id bytes + int32 positions per hit (GDX_BENCH_GATHER=arrays; any index)"""
    env = dict(os.environ, GDX_BENCH_ONE_GPU="1", GDX_BENCH_BACKEND="gloo", GDX_BENCH_GATHER=wire)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", {"compact": "29547", "arrays": "29549", "bitmap": "29551"}[wire], os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--workload", "small", "--steps", "3", "--no-bandwidth", "--nq", "1000000"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 prints the one JSON line
    assert len(lines[0].encode()) < 4096  # the driver keeps a bounded tail of stdout
    d = json.loads(lines[0])
    # BASELINE configs[3]: the line's value is ONE batch sharded over the ranks (strong scaling); rank 0 reran the batch
    # alone and compared the concatenated shards bit for bit; the every-rank-its-own-batch number sits beside it
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3
    assert "ONE batch of 1000000 reads sharded over 2 GPUs" in d["config"]["workload"]
    # (rank 0's shard is the smaller one: it also splits the shard it receives -- dist.root_weight_for)
    sys.path.insert(0, ROOT)
    from genedex_amd import dist as gdist

    full = json.load(open(os.path.join(ROOT, d["side_file"]) if not os.path.isabs(d["side_file"]) else d["side_file"]))
    st = full["strong_scaling"]
    lo0, hi0 = gdist.shard_range(1_000_000, 0, 2, st["root_weight"])
    assert 0.25 <= st["root_weight"] <= 2.0 and st["gather_probe_GBps_per_link"] > 0
    assert d["config"]["queries_total"] == 1_000_000 and d["config"]["queries_per_gpu"] == hi0 - lo0 and d["value"] > 0
    assert abs(d["value"] - 1_000_000 / (d["ms_per_step"] / 1e3)) < 1e-4 * d["value"]
    # arrays: 1-byte counts + 5 bytes per hit (text id byte + int32 position); compact: 4 bytes per query + the exceptions
    assert d["config"]["gathered_bytes_per_rank_and_step"] > 500_000
    assert d["config"]["gather_wire"] == wire
    assert d["parity"]["hits_checked"] == d["parity"]["hits_matching_text"] > 0
    assert d["parity"]["shards_equal_single_rank_output"] == {"counts": True, "hits": True}
    assert d["cpu_baseline"] is None  # N = 1 only
    # the preflight (before anything was timed): a small gathered step equal to rank 0's own output, the backend's rank count, the
    # probe's link rate -- in the line
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["gather_backend"] == "gloo" and d["config"]["gather_link_GBps"] > 0
    assert full["config"]["preflight"]["shards_equal_single_rank_output"] == {"counts": True, "hits": True}
    w = d["weak_scaling"]
    assert w["queries_per_gpu"] == 1_000_000 and w["value"] > 0 and w["gathered_bytes_per_rank_and_step"] > 1_000_000
    assert st["scaling"] == "strong" and st["queries_total"] == 1_000_000 and st["queries_this_rank"] == hi0 - lo0
    assert d["results_sharded"]["value"] > 0 and st["results_sharded"]["ms_per_step"] > 0  # the same step without the gather
    assert st["shards_equal_single_rank_output"] == {"counts": True, "hits": True} and st["gather_wire"] == wire


def _device_count():
    import torch

    return torch.cuda.device_count()


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl():
    """The same run on the real backend -- torch's "nccl" process group IS RCCL on ROCm -- one rank per GPU: weak and
    strong scaling, the shard == single-rank check included.  Needs two GPUs; the single-GPU boxes skip it."""
    if _device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GDX_BENCH_ONE_GPU", None)
    env.pop("GDX_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29548", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "small",
           "--steps", "3", "--no-bandwidth", "--nq", "1000000"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["parity"]["shards_equal_single_rank_output"] == {"counts": True, "hits": True}


def test_gather_count_dtype_is_one_rccl_maps():
    """torch's NCCL/RCCL process group has no 16-bit integer type (ProcessGroupNCCL: int8, uint8, int32, int64, floats):
    the gathered per-query counts must travel as uint8 or int32 whatever the largest count is."""
    import re

    src = open(os.path.join(ROOT, "benchlib", "multi.py")).read()
    m = re.search(r"count_dtype = (.*)", src)
    assert m and "int16" not in m.group(1) and "uint8" in m.group(1) and "int32" in m.group(1)
