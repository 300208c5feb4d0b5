#!/bin/bash
mkdir -p gpurun_out/r2c
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r2c/pytest.log
cat gpurun_out/r2c/pytest.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --secondary-depth 0 --no-bandwidth --no-live-pmc > gpurun_out/r2c/bench_records.json 2> gpurun_out/r2c/bench_records.err
tail -2 gpurun_out/r2c/bench_records.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2c/bench_records.json"))
print("records", d["value"], d["ms_per_step"], d["kernel_ms"], d["parity"])
PY
python tools/microbench_small.py 4 32 96 > gpurun_out/r2c/microbench_small.json 2> gpurun_out/r2c/microbench_small.err
cat gpurun_out/r2c/microbench_small.json
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/r2c/pmc_sq -- python3 $R/bench.py --pmc-child > $R/gpurun_out/r2c/pmc_sq.log 2>&1
python3 $R/tools/summarize_rocprof.py pmc $R/gpurun_out/r2c/pmc_sq > $R/gpurun_out/r2c/pmc_sq.json
find $R/gpurun_out/r2c/pmc_sq -name '*.csv' -delete
python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/r2c/pmc_sq.json"))
for c,v in d.items():
    for k,x in v.items():
        if "search_pair" in k or "locate_queue" in k: print(c,k[:50],x["launches"],x["per_launch"])
PY
