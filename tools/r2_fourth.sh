#!/bin/bash
mkdir -p gpurun_out/r2e
python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=25 2>&1 | tail -40 > gpurun_out/r2e/pytest.log
cat gpurun_out/r2e/pytest.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --secondary-depth 0 --no-bandwidth --no-live-pmc > gpurun_out/r2e/bench_records.json 2> gpurun_out/r2e/bench_records.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2e/bench_records.json"))
print("records", d["value"], d["ms_per_step"], d["kernel_ms"], d["parity"])
PY
