#!/bin/bash
# request and instruction counters of the kernels matching a regex in one of the tools/exp_*.py scripts: rocprofv3 --pmc, one
# counter group per run, never mixed with tracing.  usage: tools/pmc_exp.sh <out_dir_under_gpurun_out> <kernel regex> <script> [args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
REGEX=$1; shift
SCRIPT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_HIT_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-include-regex "$REGEX" --output-format csv -d $OUT/p$i -- python3 $R/$SCRIPT "$@" > $OUT/p$i.log 2>&1
done
python3 $R/tools/summarize_rocprof.py pmc $OUT > $OUT/summary.json
find $OUT -name '*_counter_collection.csv' -size +2M -delete
find $OUT -name '*agent_info.csv' -delete
