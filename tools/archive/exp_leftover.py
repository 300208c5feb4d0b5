import os, sys, torch
sys.path.insert(0, "/root/repo")
from genedex_amd import alphabet
from genedex_amd.device import DeviceEngine, DeviceQueries, build_index_from_device_text, hg38_text_lengths, synth_text
total = 3_100_000_000
dev = torch.device("cuda", 0)
io_text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
lengths = hg38_text_lengths(total, 24)
index = build_index_from_device_text(io_text, lengths, alphabet.ascii_dna_with_n(), index_storage="u32")
eng = DeviceEngine(index)
for L in (50, 40, 58, 64, 100):
    n = 10_000_000
    q = DeviceQueries.synth(io_text, lengths, n, L, L, 900_000, seed=43)
    rec = eng.alloc_records(n)
    print("len", L, file=sys.stderr); sys.stderr.flush()
    eng.locate_search(q, rec)
    torch.cuda.synchronize()
    cnt = rec[:n,1]-rec[:n,0]
    print("  found", int((cnt>0).sum()), "multi", int((cnt>1).sum()), file=sys.stderr)
