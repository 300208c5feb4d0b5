#!/bin/bash
# per-dispatch timeline of one of the tools/exp_*.py scripts (the last N launches with the gaps between them):
# tools/trace_timeline.sh <out_dir_under_gpurun_out> <n> <script> [args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
N=$1; shift
SCRIPT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/$SCRIPT "$@" > $OUT/trace_out.json 2> $OUT/trace.err
python3 $R/tools/summarize_rocprof.py stats $OUT/trace 40 > $OUT/kernel_stats.md
python3 $R/tools/summarize_rocprof.py timeline $OUT/trace $N > $OUT/timeline.md
find $OUT/trace -name '*.csv' -delete
tail -5 $OUT/timeline.md
