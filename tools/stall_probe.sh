#!/bin/bash
# Round 4, VERDICT item 2: tests/test_gpu_seed.py alone, as the first file of a fresh process, WITHOUT the conftest fixture
# that brings torch's runtime up first -- three times, every test named with its duration, a traceback dump of every thread
# after 150 s in one test, a hard stop after 900 s per run.  Output: gpurun_out/stall/run{1,2,3}.log
mkdir -p gpurun_out/stall
# (until round 4 a session fixture brought torch up first; the probe ran with it disabled: GDX_TEST_NO_RUNTIME_FIRST=1)
for i in 1 2 3; do
  timeout 900 python -X faulthandler -m pytest tests/test_gpu_seed.py -m gpu -v --durations=0 \
      -o faulthandler_timeout=150 -p no:cacheprovider > gpurun_out/stall/run$i.log 2>&1
  echo "run $i rc=$?" | tee -a gpurun_out/stall/summary.txt
  tail -3 gpurun_out/stall/run$i.log | tee -a gpurun_out/stall/summary.txt
done
# the same with the fixture, once

timeout 900 python -X faulthandler -m pytest tests/test_gpu_seed.py -m gpu -v --durations=0 -o faulthandler_timeout=150 \
    -p no:cacheprovider > gpurun_out/stall/run_fixture.log 2>&1
echo "with fixture rc=$?" | tee -a gpurun_out/stall/summary.txt
tail -3 gpurun_out/stall/run_fixture.log | tee -a gpurun_out/stall/summary.txt
