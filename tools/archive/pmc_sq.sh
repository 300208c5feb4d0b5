#!/bin/bash
# instruction-issue counters of the search kernels at the full batch (PMC only, one group per run)
# usage: tools/pmc_sq.sh <out_dir_under_gpurun_out> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-bandwidth --no-live-pmc --no-strong --no-extras --secondary-depth 0 --verify-hits 0 $*"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --kernel-include-regex "search_fast|search_pair|locate_queue" --output-format csv -d $OUT/p$i -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1
done
python3 $R/tools/summarize_rocprof.py pmc $OUT > $OUT/summary.json
find $OUT -name '*_counter_collection.csv' -size +2M -delete
find $OUT -name '*agent_info.csv' -delete
