#!/bin/bash
mkdir -p gpurun_out/r2j
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed or fused or cursor_strings" 2>&1 | tail -8 > gpurun_out/r2j/pytest.log
cat gpurun_out/r2j/pytest.log
python bench.py --no-live-pmc --no-cpu-baseline --secondary-depth 0 > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err; echo rc=$?
grep -E "end to end|Error|error|PARITY" gpurun_out/r2j/bench.err | cut -c1-3000
