#!/bin/bash
# Calibrates FETCH_SIZE / TCC_EA0_RDREQ on known byte counts: the gather and stream micro-benchmarks of
# include/gdx_bench.h move exactly n_accesses * line_bytes (gathers) or `bytes` (streams) per launch.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/calib
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/tools/microbench.py 4 > $OUT/p$i.log 2>&1
done
python3 $R/tools/summarize_rocprof.py pmc $OUT > $OUT/summary.json
