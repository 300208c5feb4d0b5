#!/bin/bash
# Everything profiles/r03/*final* is made of, on the GPU box (via gpurun, from the repo root): tools/final_profiles_r3.sh <tag> [bench]
# (second argument "bench": only the bench line, its PMC fallback and the kernel stats)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
MODE=${2:-all}
[ "$MODE" = all ] && python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/gpu_tests_final.log
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_hg38_final.json 2> $OUT/bench.err; echo bench rc=$?
python3 tools/make_pmc_final_r2.py $OUT/bench_hg38_final.json $OUT/search_pmc_final.json > /dev/null
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 \
  --no-live-pmc --no-cpu-baseline --no-bandwidth --no-extras --secondary-depth 0 > $OUT/trace_bench.json 2> $OUT/trace.err
python3 $R/tools/summarize_rocprof.py stats $OUT/trace > $OUT/bench_hg38_final_kernel_stats.md
find $OUT/trace -name '*.csv' -delete
cd $R
[ "$MODE" = all ] || { head -12 $OUT/bench_hg38_final_kernel_stats.md; exit 0; }
# exact intervals and the cursor API on the index with every structure (seed table + inverse suffix array + tables)
python3 tools/exp_general.py 3 seed_symbols=1 inverse_suffix_array=1 aux_budget_bytes=250000000000 > $OUT/exp_general_final.json 2> /dev/null
python3 tools/exp_seed.py 3 > $OUT/exp_seed_final.json 2> /dev/null
python3 tests/parity_sweep.py 300 21 > $OUT/parity_sweep_seed21.json 2> $OUT/parity21.err
bash tools/pmc_seed.sh $1/pmc_seed_kernel > /dev/null 2>&1
python3 tools/exp_gather_pack.py 5 > $OUT/exp_gather_pack.json 2> /dev/null
bash tools/trace_exp.sh $1/genome_like_trace tools/exp_genome_like.py seed_symbols=1 aux_budget_bytes=250000000000 full_sa=1 > /dev/null 2>&1
tail -3 $OUT/gpu_tests_final.log; tail -c 400 $OUT/parity_sweep_seed21.json; echo; cat $OUT/bench_hg38_final_kernel_stats.md | head -12
