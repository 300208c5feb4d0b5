#!/bin/bash
mkdir -p gpurun_out/r2i
python -m pytest tests -m gpu -x -q --durations=10 2>&1 | tail -30 > gpurun_out/r2i/pytest.log
cat gpurun_out/r2i/pytest.log
python bench.py > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/bench.err; echo rc=$?
grep -E "end to end|genome_like|Error|error|PARITY" gpurun_out/r2i/bench.err | cut -c1-2500
