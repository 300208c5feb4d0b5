#!/bin/bash
mkdir -p gpurun_out/r2h
python -m pytest tests/test_gpu_api_contract.py tests/test_gpu_fullsize.py -m gpu -x -q --durations=8 2>&1 | tail -25 > gpurun_out/r2h/pytest.log
cat gpurun_out/r2h/pytest.log
python bench.py --no-live-pmc --no-cpu-baseline > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; echo rc=$?
grep -E "end to end|mixed_lengths|lookup_depth|Error|error" gpurun_out/r2h/bench.err | cut -c1-1500
