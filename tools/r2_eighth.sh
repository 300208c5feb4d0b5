#!/bin/bash
mkdir -p gpurun_out/r2k
python -m pytest tests/test_gpu_parity.py tests/test_gpu_api_contract.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r2k/pytest.log
cat gpurun_out/r2k/pytest.log
for d in 3 0; do
GDX_SEARCH_DEFER=$d python bench.py --no-live-pmc --no-cpu-baseline > gpurun_out/r2k/bench_d$d.json 2> gpurun_out/r2k/bench_d$d.err; echo rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/r2k/bench_d$d.json"))
print("defer $d", d["value"], d["ms_per_step"], d["kernel_ms"])
e=d["end_to_end"]; print({k:e[k] for k in ("count_seconds","locate_seconds","count_over_bound","locate_over_bound")}, e["packed_queries"]["count_seconds"])
for s in d["secondary"]:
    if "genome" in s["name"] or "mixed" in s["name"]: print({k:s.get(k) for k in ("name","value","ms_per_step","search_ms","scan_and_locate_ms","fused_ms","cursor_api_ms")})
PY
done
