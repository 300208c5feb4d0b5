#!/bin/bash
# Runs bench.py under rocprofv3 once per counter group (PMC passes must not be mixed with tracing).
# PMC passes serialise and replay kernels: keep the workload small (--nq 10000000).
# TA_* counters abort rocprofv3 on this pool (signal 6, then a hang) -- never list them.
# usage: tools/pmc_passes.sh <out_dir_under_gpurun_out> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-bandwidth --secondary-depth 0 --verify-hits 0 $*"
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INST_LEVEL_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1
done
python3 $R/tools/summarize_rocprof.py pmc $OUT > $OUT/summary.json
