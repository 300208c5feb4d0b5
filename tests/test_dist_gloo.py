"""World-size-2 gloo test of the multi-GPU plumbing (genedex_amd/dist.py): contiguous query shards, every rank
answers its shard against its own index replica (the CPU oracle stands in for the GPU here), results are
gathered to rank 0 and must equal the single-process answer."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genedex_amd import alphabet, synth
    from genedex_amd import dist as gdist
    from oracle.oracle import OracleIndex

    a = alphabet.ascii_dna_with_n()
    texts = synth.host_texts(total=60_000, n_texts=3, seed=42)
    replica = OracleIndex.build(texts, a.io_to_dense_table, 6, 4, sa_rate=4, lookup_depth=3)
    nq = 5001  # not divisible by the world size
    qbuf, qoff = synth.host_queries(texts, nq=nq, len_min=10, len_max=40, sampled_fraction=0.7, seed=43)
    lo, hi = gdist.shard_range(nq, rank, world)
    sub_off = qoff[lo:hi + 1] - qoff[lo]
    sub_buf = qbuf[int(qoff[lo]):int(qoff[hi])] if qoff[hi] > qoff[lo] else np.zeros(1, dtype=np.uint8)
    s, e = replica.cursors_for_many(np.ascontiguousarray(sub_buf), np.ascontiguousarray(sub_off))
    off, t, p = replica.locate_intervals(s, e)
    counts = torch.from_numpy((e - s).astype(np.int64))
    pad = torch.zeros(nq // world + 1, dtype=torch.int64)
    pad[: counts.numel()] = counts
    got_counts = gdist.gather_variable(pad, counts.numel(), dst=0)
    hits = torch.from_numpy(np.stack([t.astype(np.int64), p.astype(np.int64)], axis=1))
    got_hits = gdist.gather_variable(hits, hits.shape[0], dst=0)
    # pipelined (asynchronous, double-buffered) gather, the form bench.py uses between steps
    max_hits = gdist.max_int_over_ranks(hits.shape[0], torch.device("cpu"))
    slots = [[torch.zeros(nq // world + 1, dtype=torch.int64), torch.zeros((max_hits, 2), dtype=torch.int64),
              torch.zeros(nq // world + 1, dtype=torch.uint8)]  # bench.py ships counts as the narrowest type
             for _ in range(2)]
    pg = gdist.PipelinedGather(slots, dst=0)
    pipelined_ok = True
    for step in range(5):
        slot = step % 2
        pg.acquire(slot)
        slots[slot][0].zero_()
        slots[slot][0][: counts.numel()] = counts + step
        slots[slot][1].zero_()
        slots[slot][1][: hits.shape[0]] = hits
        slots[slot][2].zero_()
        slots[slot][2][: counts.numel()].copy_(torch.clamp(counts, max=255))
        pg.submit(slot)
    pg.drain()
    if rank == 0:
        for slot, step in ((0, 4), (1, 3)):
            got = pg.gathered(slot)
            for r in range(world):
                rlo, rhi = gdist.shard_range(nq, r, world)
                pipelined_ok &= bool((got[0][r][: rhi - rlo] >= step).all())
            pipelined_ok &= torch.equal(got[1][0][: hits.shape[0]], hits)
            pipelined_ok &= torch.equal(got[2][0][: counts.numel()].to(torch.int64), torch.clamp(counts, max=255))
    fixed = gdist.gather_fixed(torch.tensor([rank, hi - lo]), dst=0)
    tmax = gdist.max_over_ranks(float(rank + 1), torch.device("cpu"))
    ints_ok = gdist.gather_ints(10 + rank, torch.device("cpu")) == [10, 11]
    gdist.barrier()
    if rank == 0:
        all_counts = torch.cat(got_counts).numpy()
        all_hits = torch.cat(got_hits).numpy()
        fs, fe = replica.cursors_for_many(qbuf, qoff)
        foff, ft, fp = replica.locate_intervals(fs, fe)
        ok = (np.array_equal(all_counts, (fe - fs).astype(np.int64))
              and np.array_equal(all_hits[:, 0], ft.astype(np.int64)) and np.array_equal(all_hits[:, 1], fp.astype(np.int64))
              and [x.tolist() for x in fixed] == [[0, nq // 2], [1, nq - nq // 2]] and tmax == 2.0 and pipelined_ok and ints_ok)
        open(result_path, "w").write("ok" if ok else "mismatch")
    dist.destroy_process_group()


def test_sharded_queries_gathered_equal_single_process(tmp_path):
    result = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), str(result)), nprocs=2, join=True)
    assert result.read_text() == "ok"


def test_shard_ranges_partition_the_batch():
    from genedex_amd.dist import shard_range

    for n in (0, 1, 7, 100, 100_000_001):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1
