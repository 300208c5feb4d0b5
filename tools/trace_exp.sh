#!/bin/bash
# kernel-trace statistics of one of the tools/exp_*.py scripts: tools/trace_exp.sh <out_dir_under_gpurun_out> <script> [args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
SCRIPT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/$SCRIPT "$@" > $OUT/trace_out.json 2> $OUT/trace.err
python3 $R/tools/summarize_rocprof.py stats $OUT/trace 80 > $OUT/kernel_stats.md
find $OUT/trace -name '*.csv' -delete
head -14 $OUT/kernel_stats.md
