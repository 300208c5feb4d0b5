#!/bin/bash
# Everything profiles/r02/*final* is made of, on the GPU box (via gpurun, from the repo root): tools/final_profiles_r2.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_hg38_final.json 2> $OUT/bench.err; echo bench rc=$?
python3 tools/make_pmc_final_r2.py $OUT/bench_hg38_final.json $OUT/search_pmc_final.json > /dev/null
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 \
  --no-live-pmc --no-cpu-baseline --no-bandwidth --no-extras --secondary-depth 0 > $OUT/trace_bench.json 2> $OUT/trace.err
python3 $R/tools/summarize_rocprof.py stats $OUT/trace > $OUT/bench_hg38_final_kernel_stats.md
find $OUT/trace -name '*.csv' -delete
cd $R
python3 tests/parity_sweep.py 250 1 > $OUT/parity_sweep_seed1.json 2> $OUT/parity1.err
python3 tests/parity_sweep.py 250 2 > $OUT/parity_sweep_seed2.json 2> $OUT/parity2.err
python3 tools/microbench_small.py 4 32 96 > $OUT/microbench_small.json 2> /dev/null
tail -c 600 $OUT/parity_sweep_seed1.json; echo; cat $OUT/bench_hg38_final_kernel_stats.md | head -12
