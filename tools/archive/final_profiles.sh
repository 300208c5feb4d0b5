#!/bin/bash
# Produces everything profiles/r01/*final* is made of, on the GPU box:
#   1. the default bench line                       -> gpurun_out/<tag>/bench.json
#   2. the same command under rocprofv3 --kernel-trace --stats -> kernel_stats.md (tools/summarize_rocprof.py stats)
#   3. PMC passes (count+locate, 10 M reads)        -> pmc/summary.json
# usage (from the repo root, via gpurun): tools/final_profiles.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline \
  --no-bandwidth --secondary-depth 0 > $OUT/trace_bench.json 2> $OUT/trace.err
python3 $R/tools/summarize_rocprof.py stats $OUT/trace > $OUT/kernel_stats.md
cd $R
tools/pmc_passes.sh $TAG/pmc --nq 10000000 > $OUT/pmc.log 2>&1
# keep the merge small: the raw traces are tens of MB
find $OUT/trace -name '*kernel_trace.csv' -delete
