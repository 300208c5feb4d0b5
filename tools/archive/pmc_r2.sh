#!/bin/bash
# PMC passes of bench.py at the FULL batch (one counter group per rocprofv3 run, never mixed with tracing).
# usage: tools/pmc_r2.sh <out_dir_under_gpurun_out> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--no-live-pmc --no-strong --no-extras --steps 1 --warmup 0 --no-cpu-baseline --no-bandwidth --secondary-depth 0 --verify-hits 0 $*"
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_HIT_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1
done
python3 $R/tools/summarize_rocprof.py pmc $OUT > $OUT/summary.json
find $OUT -name '*_counter_collection.csv' -size +2M -delete
find $OUT -name '*agent_info.csv' -delete
