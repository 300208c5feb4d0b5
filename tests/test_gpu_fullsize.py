"""Full-size parity tests of BASELINE.json configs[2] and configs[4] (hg38-scale text, 3.1 G symbols, 24 texts):
size-independent properties over the whole batch plus a prefix compared bit for bit with the CPU oracle running on
the very same index (BWT and samples exported from the GPU build).  One index serves both tests."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

TOTAL = 3_100_000_000
N_TEXTS = 24


@pytest.fixture(scope="module")
def hg38_scale():
    import torch

    from genedex_amd import alphabet
    from genedex_amd.device import DeviceEngine, build_index_from_device_text, hg38_text_lengths, synth_text
    from oracle.oracle import OracleIndex

    dev = torch.device("cuda", 0)
    a = alphabet.ascii_dna_with_n()
    io_text = synth_text(TOTAL, seed=42, n_per_million=10_000, device=dev)
    lengths = hg38_text_lengths(TOTAL, N_TEXTS)
    index = build_index_from_device_text(io_text, lengths, a, index_storage="u32")
    eng = DeviceEngine(index)
    aux = eng.aux_info()
    assert aux["pair_lines"] and aux["jump_entry_bytes"] == 32 and aux["top_table_depth"] == 16 and not aux["shrunk_by_budget"]
    threads = min(os.cpu_count() or 1, 64)
    cpu = OracleIndex.from_bwt(index.export_bwt(), index.export_sa_samples(), 4, *index.export_borders(),
                               index.export_sentinel_indices(), a.io_to_dense_table, 6, 4, width=32, n_threads=threads)
    yield {"torch": torch, "dev": dev, "io_text": io_text, "lengths": lengths, "index": index, "eng": eng, "cpu": cpu,
           "threads": threads}
    del eng, index
    torch.cuda.empty_cache()


def test_full_size_properties_workload3(hg38_scale):
    """configs[2]: 100 M len-50 reads, count + locate on one GPU."""
    import bench
    from genedex_amd.device import DeviceQueries

    h = hg38_scale
    torch, eng, dev = h["torch"], h["eng"], h["dev"]
    nq = 100_000_000
    q = DeviceQueries.synth(h["io_text"], h["lengths"], nq, 50, 50, 900_000, seed=43)
    # the timed path of bench.py: search records (lazy tails) -> offsets -> hits
    rec = eng.alloc_records(nq)
    off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
    eng.locate_search(q, rec)
    eng.locate_offsets(rec, nq, off)
    torch.cuda.synchronize()
    total = int(off[nq].item())
    counts = (rec[:nq, 1] - rec[:nq, 0]).to(torch.int64)
    assert int(((rec[:nq, 3] >> 24) & 0xff).ne(0).sum().item()) == 0  # no query with a status
    assert int(counts.sum().item()) == total
    found = int((counts > 0).sum().item())
    assert 0.8990 * nq < found < 0.9005 * nq  # the 90 % sampled reads are all found, a few random ones too
    hits = torch.empty((total, 2), dtype=torch.int32, device=dev)
    ws = torch.empty(eng.locate_workspace_bytes(total), dtype=torch.uint8, device=dev)
    eng.locate_hits(rec, nq, off, total, hits, ws)
    torch.cuda.synchronize()
    chk = bench.verify_hits(torch, h["io_text"], h["lengths"], q, {"hit_offsets": off}, hits, total, nq, 2_000_000)
    assert chk["hits_checked"] == chk["hits_matching_text"] == 2_000_000  # every checked hit spells its read
    # the interval path (exact intervals + hints, gdx_cursors_for_many_queries_hint_dev) gives the same hits, with and
    # without its hints, and with the walk on the rank lines alone
    out = eng.alloc_outputs(nq, hint=True)
    eng.search(q, out)
    eng.hit_offsets(out, nq)
    torch.cuda.synchronize()
    assert torch.equal(out["hit_offsets"], off) and not bool(out["status"].any().item())
    assert torch.equal((out["end"] - out["start"]).to(torch.int64), counts)
    hits2 = torch.empty_like(hits)
    eng.locate(out, nq, total, hits2, ws)
    torch.cuda.synchronize()
    assert torch.equal(hits, hits2)
    plain = {k: v for k, v in out.items() if k != "hint"}
    h["index"].set_query_options(locate_jump_walk=False)
    eng.locate(plain, nq, total, hits2, ws)
    torch.cuda.synchronize()
    h["index"].set_query_options()
    assert torch.equal(hits, hits2)
    del hits2
    # a prefix against the oracle on the same index: intervals, hit offsets, hits in the same order
    m = 1_500_000
    qbuf, qoff = q.host_slice(0, m)
    cs, ce = h["cpu"].cursors_for_many(qbuf, qoff, n_threads=h["threads"])
    assert np.array_equal(out["start"][:m].cpu().numpy().astype(np.uint32), cs.astype(np.uint32))
    assert np.array_equal(out["end"][:m].cpu().numpy().astype(np.uint32), ce.astype(np.uint32))
    co, ct, cp = h["cpu"].locate_intervals(cs, ce, n_threads=h["threads"])
    assert np.array_equal(off[:m + 1].cpu().numpy().astype(np.uint64), co)
    gh = hits[: int(co[-1])].cpu().numpy().astype(np.uint32)
    assert np.array_equal(gh[:, 0], ct.astype(np.uint32)) and np.array_equal(gh[:, 1], cp.astype(np.uint32))


def test_full_size_properties_workload5(hg38_scale):
    """configs[4]: 50 M reads of length 20..150, 70 % sampled / 30 % random, through the fused call and through the
    batched cursor API (cursor_empty + gdx_cursor_extend_front_strings_dev, 32 symbols per call, device-side active
    lists): identical intervals, early termination visible in the shrinking active lists."""
    from genedex_amd.device import DeviceQueries

    h = hg38_scale
    torch, eng, dev = h["torch"], h["eng"], h["dev"]
    nq = 50_000_000
    q = DeviceQueries.synth(h["io_text"], h["lengths"], nq, 20, 150, 700_000, seed=47)
    out = eng.alloc_outputs(nq)
    eng.search(q, out)
    torch.cuda.synchronize()
    assert not bool(out["status"].any().item())
    found = int((out["end"] != out["start"]).sum().item())
    # 70 % of the reads are drawn from the text, but a window with an N is redrawn at most 8 times and then replaced
    # by a random read: long reads (P(no N in 150 symbols) = 0.22) are found a little less often than 70 %
    assert 0.66 * nq < found < 0.70 * nq
    n = h["index"].total_text_len()
    beg, end = q.qoff[:-1], q.qoff[1:]
    lens = end - beg
    assert int(lens.min().item()) == 20 and int(lens.max().item()) == 150
    cur_s = torch.zeros(nq, dtype=torch.int32, device=dev)
    cur_e = torch.full((nq,), n - (1 << 32), dtype=torch.int32, device=dev)  # n as u32
    cur_st = torch.zeros(nq, dtype=torch.uint8, device=dev)
    act = [torch.empty(nq, dtype=torch.int32, device=dev) for _ in range(2)]
    n_act = [torch.empty(1, dtype=torch.int32, device=dev) for _ in range(2)]
    hi, a, na, live = end, None, None, []
    for r in range(5):  # 5 x 32 symbols >= 150
        lo = torch.maximum(hi - 32, beg)
        eng.cursor_extend_strings(cur_s, cur_e, q.qbuf, lo, hi, nq, cur_st, a, na, act[r % 2], n_act[r % 2])
        a, na, hi = act[r % 2], n_act[r % 2], lo
        live.append(int(na.item()))
    assert bool((hi == beg).all().item())  # every symbol was offered
    assert torch.equal(cur_s, out["start"]) and torch.equal(cur_e, out["end"]) and not bool(cur_st.any().item())
    # random reads die in the first call (top table), found reads stay alive to the end
    assert live == sorted(live, reverse=True) and live[0] < 0.70 * nq and live[-1] == found
    # the same cursors, one symbol per launch (Cursor::extend_query_front as the reference has it), on a prefix
    m = 200_000
    s1 = torch.zeros(m, dtype=torch.int32, device=dev)
    e1 = torch.full((m,), n - (1 << 32), dtype=torch.int32, device=dev)
    st1 = torch.zeros(m, dtype=torch.uint8, device=dev)
    for j in range(150):
        at = end[:m] - 1 - j
        alive = at >= beg[:m]
        sym = torch.where(alive, q.qbuf[at.clamp(min=0)], torch.full_like(at, ord("A"), dtype=torch.uint8))
        keep_s, keep_e = s1.clone(), e1.clone()
        eng.lib.gdx_cursor_extend_front_many_dev(eng.h, s1.data_ptr(), e1.data_ptr(), sym.data_ptr(), m, st1.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream)
        s1 = torch.where(alive, s1, keep_s)  # exhausted queries keep their cursor
        e1 = torch.where(alive, e1, keep_e)
    assert torch.equal(s1, out["start"][:m]) and torch.equal(e1, out["end"][:m])
    # a prefix against the oracle on the same index (the 64-wide batched path of the reference, restated)
    mo = 1_000_000
    qbuf, qoff = q.host_slice(0, mo)
    cs, ce = h["cpu"].cursors_for_many(qbuf, qoff, n_threads=h["threads"])
    assert np.array_equal(out["start"][:mo].cpu().numpy().astype(np.uint32), cs.astype(np.uint32))
    assert np.array_equal(out["end"][:mo].cpu().numpy().astype(np.uint32), ce.astype(np.uint32))
