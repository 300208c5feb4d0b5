#!/bin/bash
# search time against occupancy: dynamic LDS per block caps the resident blocks per CU
mkdir -p gpurun_out/r2f
for pad in 0 14000 23000 31000 44000; do
  GDX_SEARCH_PAD=$pad python bench.py --steps 6 --warmup 2 --no-cpu-baseline --secondary-depth 0 --no-bandwidth --no-live-pmc --verify-hits 0 > gpurun_out/r2f/b_$pad.json 2> gpurun_out/r2f/b_$pad.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2f/b_$pad.json"))
print("pad $pad", d["ms_per_step"], d["kernel_ms"])
PY
done
