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
    # the compact wire (bench.py make_compact_gather): one word per query -- the position of its only hit in the concatenated
    # texts, none, or "see the exceptions" -- plus counts and hits of the exceptions; the root splits the words into text id +
    # position (gdx_compact_split_hits_dev on the GPU; restated with numpy here) when the gather is acquired
    starts = np.concatenate([[0], np.cumsum([len(x) + 1 for x in texts])]).astype(np.int64)
    n_shard = hi - lo
    cnt_np = (e - s).astype(np.int64)
    words = np.full(n_shard, gdist.COMPACT_SEE, dtype=np.int32)
    words[cnt_np == 0] = gdist.COMPACT_NONE
    single = np.flatnonzero((cnt_np == 1) & (np.arange(n_shard) % 7 != 0))  # (some single hits travel as exceptions too)
    words[single] = (starts[t[off[single]].astype(np.int64)] + p[off[single]].astype(np.int64)).astype(np.int32)
    hits32 = torch.from_numpy(np.stack([t.astype(np.int32), p.astype(np.int32)], axis=1))
    n_exc, n_exc_hits = gdist.exception_sizes(torch.from_numpy(words), torch.from_numpy(off.astype(np.int64)), n_shard)
    cpu = torch.device("cpu")
    cap_q, cap_h = gdist.max_int_over_ranks(n_exc, cpu), gdist.max_int_over_ranks(n_exc_hits, cpu)
    n_max = nq // world + 1
    cslots = [[torch.full((n_max,), -1, dtype=torch.int32), torch.zeros(cap_q, dtype=torch.int32),
               torch.zeros(cap_h, dtype=torch.uint8), torch.zeros(cap_h, dtype=torch.int32), torch.zeros(2, dtype=torch.int32)]
              for _ in range(2)]
    split = [[None] * world for _ in range(2)]
    arrivals = []

    def on_gathered(slot):
        arrivals.append(slot)
        for r, w in enumerate(cg.gathered(slot)[0]):
            w = w.numpy().astype(np.int64)
            tid = np.searchsorted(starts[1:] - 1, np.maximum(w, 0), side="left")  # the sentinel at or after the position
            inside = np.where(w >= 0, w - starts[np.minimum(tid, len(texts) - 1)], w)
            split[slot][r] = (torch.from_numpy(np.where(w >= 0, tid, 0).astype(np.uint8)), torch.from_numpy(inside.astype(np.int32)))

    cg = gdist.PipelinedGather(cslots, dst=0, on_gathered=on_gathered)
    for step in range(3):
        slot = step % 2
        cg.acquire(slot)
        cslots[slot][0][:n_shard] = torch.from_numpy(words)
        gdist.pack_exceptions(cslots[slot][0], torch.from_numpy(off.astype(np.int64)), hits32, n_shard, *cslots[slot][1:])
        cg.submit(slot)
    cg.drain()
    compact_ok = True
    if rank == 0:
        compact_ok = sorted(arrivals) == [0, 0, 1]
        c_cnt, c_hits = [], []
        for r in range(world):
            rlo, rhi = gdist.shard_range(nq, r, world)
            got = cg.gathered(0)
            cc, hh = gdist.expand_split_results(split[0][r][0], split[0][r][1], got[1][r], got[2][r], got[3][r], got[4][r], rhi - rlo)
            c_cnt.append(cc)
            c_hits.append(hh)
        c_cnt, c_hits = torch.cat(c_cnt).numpy(), torch.cat(c_hits).numpy()
    else:
        compact_ok = arrivals == []
    # the found-bitmap wire (bench.py make_bitmap_gather): a bit per read + the positions of the found reads + the exceptions,
    # everything a rank sends in ONE byte buffer (dist.WireLayout); gdx_wire_pack_dev / gdx_wire_split_dev restated with tensor
    # operations (dist.wire_pack_reference / wire_split_reference: what the GPU tests compare the kernels with)
    n_found = int(((words >= 0) | (words < -2)).sum())
    layout = gdist.WireLayout(n_max, gdist.max_int_over_ranks(n_found, cpu), cap_q, cap_h)
    bbufs = [torch.zeros(layout.nbytes, dtype=torch.uint8) for _ in range(2)]
    bsplit = [[None] * world for _ in range(2)]
    tstarts = torch.from_numpy(starts)

    def on_bitmap(slot):
        for r, buf in enumerate(bg.gathered(slot)[0]):
            rlo, rhi = gdist.shard_range(nq, r, world)
            bsplit[slot][r] = gdist.wire_split_reference(layout.views(buf), rhi - rlo, tstarts)

    bg = gdist.PipelinedGather([[b] for b in bbufs], dst=0, on_gathered=on_bitmap)
    for step in range(3):
        slot = step % 2
        bg.acquire(slot)
        gdist.wire_pack_reference(torch.from_numpy(words), torch.from_numpy(off.astype(np.int64)), hits32, n_shard, layout.views(bbufs[slot]))
        bg.submit(slot)
    bg.drain()
    bitmap_ok = True
    if rank == 0:
        b_cnt, b_hits = [], []
        for r in range(world):
            rlo, rhi = gdist.shard_range(nq, r, world)
            v = layout.views(bg.gathered(0)[0][r])
            cc, hh = gdist.expand_split_results(bsplit[0][r][0], bsplit[0][r][1], v["exc_cnt"], v["exc_ids"], v["exc_pos"], v["meta"], rhi - rlo)
            b_cnt.append(cc)
            b_hits.append(hh)
            # the two wires must say the same about every read
            bitmap_ok &= torch.equal(bsplit[0][r][0][: rhi - rlo], split[0][r][0][: rhi - rlo])
            bitmap_ok &= torch.equal(bsplit[0][r][1][: rhi - rlo], split[0][r][1][: rhi - rlo])
        b_cnt, b_hits = torch.cat(b_cnt).numpy(), torch.cat(b_hits).numpy()
        bitmap_ok &= layout.nbytes < 4 * n_max + 4 * cap_q + 5 * cap_h + 8 or n_found > 0.9 * n_shard
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
              and [x.tolist() for x in fixed] == [[0, nq // 2], [1, nq - nq // 2]] and tmax == 2.0 and pipelined_ok and ints_ok
              and compact_ok and np.array_equal(c_cnt, (fe - fs).astype(np.int64))
              and np.array_equal(c_hits[:, 0], ft.astype(np.int32)) and np.array_equal(c_hits[:, 1], fp.astype(np.int32))
              and bitmap_ok and np.array_equal(b_cnt, (fe - fs).astype(np.int64))
              and np.array_equal(b_hits[:, 0], ft.astype(np.int32)) and np.array_equal(b_hits[:, 1], fp.astype(np.int32)))
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
            # a smaller shard for the root of the gather (it also splits what it receives)
            from genedex_amd.dist import root_weight_for

            w = root_weight_for(world)
            r = [shard_range(n, k, world, w) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [hi - lo for lo, hi in r]
            if world > 1 and n >= 100:
                assert max(sizes[1:]) - min(sizes[1:]) <= 1 and abs(sizes[0] - w * sizes[1]) <= 1 + w  # (both sizes are rounded)
    # many ranks: a smaller root shard; two ranks on nominal links: a larger one; slower links: larger still
    assert root_weight_for(1) == 1.0 and 0.6 < root_weight_for(8) < 0.8 and root_weight_for(100) == 0.25
    assert 1.2 < root_weight_for(2) < 1.6 and root_weight_for(8, link_GBps=50.0) > root_weight_for(8) and root_weight_for(2, link_GBps=1.0) == 2.0


def test_exceptions_beyond_their_buffers_are_noticed():
    """pack_exceptions never writes past a capacity; the true numbers travel in `meta` and the receiving side refuses"""
    import pytest

    from genedex_amd import dist as gdist

    words = torch.tensor([5, -2, -1, -2, -2], dtype=torch.int32)
    off = torch.tensor([0, 1, 3, 3, 6, 7], dtype=torch.int64)
    hits = torch.arange(14, dtype=torch.int32).reshape(7, 2)
    exc_cnt, exc_ids, exc_pos = torch.zeros(3, dtype=torch.int32), torch.zeros(4, dtype=torch.uint8), torch.zeros(4, dtype=torch.int32)
    meta = torch.zeros(2, dtype=torch.int32)
    gdist.pack_exceptions(words, off, hits, 5, exc_cnt, exc_ids, exc_pos, meta)
    assert meta.tolist() == [3, 6] and exc_cnt.tolist() == [2, 3, 1]
    assert exc_ids[:2].tolist() == [2, 4] and exc_pos[:2].tolist() == [3, 5]  # what fits whole is there
    ids, pos = torch.zeros(5, dtype=torch.uint8), torch.tensor([1, -2, -1, -2, -2], dtype=torch.int32)
    with pytest.raises(ValueError):
        gdist.expand_split_results(ids, pos, exc_cnt, exc_ids, exc_pos, meta, 5)
    big_ids, big_pos = torch.zeros(6, dtype=torch.uint8), torch.zeros(6, dtype=torch.int32)
    gdist.pack_exceptions(words, off, hits, 5, exc_cnt, big_ids, big_pos, meta)
    cnt, hh = gdist.expand_split_results(ids, pos, exc_cnt, big_ids, big_pos, meta, 5)
    assert cnt.tolist() == [1, 2, 0, 3, 1]
    assert hh.tolist() == [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10, 11], [12, 13]]
    # the same from a list of the exceptions in any order (what gdx_compact_exceptions_dev hands over), longer than needed
    exc_cnt4 = torch.zeros(4, dtype=torch.int32)
    listed = (torch.tensor([4, 1, 3, 99], dtype=torch.int32), torch.tensor([3], dtype=torch.int64))
    gdist.pack_exceptions(words, off, hits, 5, exc_cnt4, big_ids, big_pos, meta, listed)
    assert meta.tolist() == [3, 6] and exc_cnt4.tolist() == [2, 3, 1, 0]
    cnt, hh2 = gdist.expand_split_results(ids, pos, exc_cnt4, big_ids, big_pos, meta, 5)
    assert cnt.tolist() == [1, 2, 0, 3, 1] and hh2.tolist() == hh.tolist()
