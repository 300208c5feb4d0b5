#!/usr/bin/env python3
"""Index construction and queries on a genome-like text: segmental duplications with mutations, tandem repeats,
poly-A runs and long runs of N (assembly gaps) instead of the i.i.d. text of the headline benchmark.

usage: python tools/genome_like.py [total symbols, default 2^28] [n queries, default 2_000_000]
Prints one JSON line: build seconds (with the suffix sorter's own breakdown), search / locate times, and the
parity properties checked (sampled reads found, hits spell their query, a few counts against a direct scan).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genedex_amd import alphabet  # noqa: E402
from genedex_amd.device import DeviceEngine, DeviceQueries, build_index_from_device_text, genome_like_text  # noqa: E402


def count_by_scan(text: torch.Tensor, lengths, q: torch.Tensor) -> int:
    """Occurrences of q inside single texts, by comparing every alignment (no index involved)."""
    n, m = text.numel(), q.numel()
    bounds = torch.tensor(np.cumsum(lengths), device=text.device)
    found, chunk = 0, 1 << 29
    for base in range(0, n - m + 1, chunk):
        cnt = min(chunk, n - m + 1 - base)
        ok = torch.ones(cnt, dtype=torch.bool, device=text.device)
        for j in range(m):
            ok &= text[base + j:base + j + cnt] == q[j]
        starts = torch.nonzero(ok).flatten() + base
        tid = torch.searchsorted(bounds, starts, right=True)
        tid_end = torch.searchsorted(bounds, starts + m - 1, right=True)
        found += int((tid == tid_end).sum().item())
    return found


def run(total: int, nq: int) -> dict:
    dev = torch.device("cuda", 0)
    alpha = alphabet.ascii_dna_with_n()
    text = genome_like_text(total, dev)
    n_texts = 8
    lengths = [total // n_texts] * (n_texts - 1) + [total - total // n_texts * (n_texts - 1)]
    torch.cuda.synchronize()
    t0 = time.time()
    index = build_index_from_device_text(text, lengths, alpha, index_storage="u32")
    t_build = time.time() - t0
    stats = index.build_stats()
    q = DeviceQueries.synth(text, lengths, nq, 30, 100, 900_000, seed=11)
    eng = DeviceEngine(index)
    out = eng.alloc_outputs(nq)

    def timed(fn):
        torch.cuda.synchronize()
        t = time.time()
        fn()
        torch.cuda.synchronize()
        return time.time() - t

    timed(lambda: eng.search(q, out))
    t_search = timed(lambda: eng.search(q, out))
    counts = (out["end"] - out["start"]).to(torch.int64)
    all_hits = int(counts.sum().item())
    # locate only the queries with at most 10 000 occurrences (reads drawn from N gaps and poly-A have millions)
    out["end"] = torch.where(counts > 10_000, out["start"], out["end"])
    eng.hit_offsets(out, nq)
    torch.cuda.synchronize()
    total_hits = int(out["hit_offsets"][nq].item())
    res = {"total": total, "n_texts": n_texts, "build_seconds": t_build, "build_stats": stats, "queries": nq,
           "search_ms": t_search * 1e3, "hits_of_all_queries": all_hits, "hits_located": total_hits, "max_count": int(counts.max().item()),
           "queries_found": int((counts > 0).sum().item()), "queries_with_status": int((out["status"] != 0).sum().item()),
           "aux": eng.aux_info()}
    cap = 1_000_000_000
    if total_hits <= cap:
        hits = torch.empty((max(total_hits, 1), 2), dtype=torch.int32, device=dev)
        ws = torch.empty(max(eng.locate_workspace_bytes(total_hits), 16), dtype=torch.uint8, device=dev)
        res["locate_ms"] = timed(lambda: eng.locate(out, nq, total_hits, hits, ws)) * 1e3
        # every checked hit spells its query
        gen = torch.Generator(device=dev)
        gen.manual_seed(3)
        h = torch.randint(0, total_hits, (min(500_000, total_hits),), device=dev, generator=gen)
        off = out["hit_offsets"]
        qi = torch.searchsorted(off, h, right=True) - 1
        qb, qe = q.qoff[qi], q.qoff[qi + 1]
        toff = torch.zeros(n_texts + 1, dtype=torch.int64, device=dev)
        toff[1:] = torch.tensor(np.cumsum(lengths), device=dev)
        pos = toff[hits[h, 0].long()] + hits[h, 1].long()
        good = torch.ones_like(h, dtype=torch.bool)
        good &= pos + (qe - qb) <= toff[hits[h, 0].long() + 1]
        for j in range(100):
            live = (qe - qb) > j
            a = text[torch.clamp(pos + j, max=total - 1)]
            b = q.qbuf[torch.clamp(qb + j, max=q.qbuf.numel() - 1)]
            good &= (~live) | (a == b) | ((a | 0x20) == (b | 0x20))
        res["hits_checked"], res["hits_spelling_their_query"] = int(h.numel()), int(good.sum().item())
    # counts against a direct scan for a few queries, the most frequent ones included
    order = torch.argsort(counts, descending=True)
    picks = torch.cat([order[:3], order[order.numel() // 2: order.numel() // 2 + 3], order[-3:]]).tolist()
    scan_ok = 0
    for i in picks:
        qq = q.qbuf[int(q.qoff[i].item()):int(q.qoff[i + 1].item())]
        scan_ok += int(count_by_scan(text, lengths, qq) == int(counts[i].item()))
    res["counts_checked_by_scan"], res["counts_equal_scan"] = len(picks), scan_ok
    return res


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28,
                         int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000)))
