import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from genedex_amd import alphabet
from genedex_amd.device import DeviceEngine, DeviceQueries, build_index_from_device_text, synth_text
dev = torch.device("cuda", 0)
total, nq = 1 << 28, 10_000_000
text = synth_text(total, seed=42, n_per_million=10_000, device=dev)
index = build_index_from_device_text(text, [total], alphabet.ascii_dna_with_n(), index_storage="i32")
eng = DeviceEngine(index)
q = DeviceQueries.synth(text, [total], nq, 50, 50, 900_000, seed=43)
outs = {}
for name, qq in (("ascii", q), ("packed", q.as_packed(index)), ("packed+uniform", q.as_packed(index).as_uniform(50))):
    o = eng.alloc_outputs(nq)
    eng.search(qq, o); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): eng.search(qq, o)
    torch.cuda.synchronize()
    outs[name] = (o["start"].clone(), o["end"].clone())
    print(name, (time.perf_counter() - t0) / 5 * 1e3, "ms per 10 M exact intervals")
assert all(torch.equal(outs["ascii"][0], v[0]) and torch.equal(outs["ascii"][1], v[1]) for v in outs.values())
print("identical")
