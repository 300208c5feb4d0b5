#!/bin/bash
# Everything profiles/r0N/*final* is made of, on the GPU box (via gpurun, from the repo root; rounds 4 and 5):
#   tools/final_profiles.sh <tag> [bench|all]
# bench: the bench line + its side file + the PMC fallback summary + rocprofv3 kernel stats of the same command
# all  : + the full GPU test suite, the four input forms, the C host, a parity sweep
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
MODE=${2:-all}
[ "$MODE" = all ] && python3 -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/gpu_tests_final.log
# the driver's command; the one stdout line is the contract's (< 4 KB), everything else goes to the side file
python3 bench.py --steps 20 --warmup 5 --side-file $OUT/bench_secondary.json > $OUT/bench_hg38_final.json 2> $OUT/bench.err; echo bench rc=$?
wc -c $OUT/bench_hg38_final.json
python3 tools/make_pmc_final_r2.py $OUT/bench_secondary.json $OUT/search_pmc_final.json > /dev/null
# (the kernel stats of the bench's own kernel-trace child pass: what roofline.avg_launch_ms_rocprof was read from)
python3 tools/summarize_rocprof.py stats $OUT > $OUT/bench_child_kernel_stats.md 2> /dev/null
export TMPDIR=/tmp
cd /tmp
# the same command under rocprofv3 --kernel-trace --stats (no nested PMC children: the profiler's preload holds the GPU)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 \
  --no-live-pmc --no-cpu-baseline --no-bandwidth --no-extras --secondary-depth 0 --side-file $OUT/trace_secondary.json \
  > $OUT/bench_hg38_final_under_rocprof.json 2> $OUT/trace.err
python3 $R/tools/summarize_rocprof.py stats $OUT/trace > $OUT/bench_hg38_final_kernel_stats.md
find $OUT/trace -name '*.csv' -delete
cd $R
[ "$MODE" = all ] || { head -12 $OUT/bench_hg38_final_kernel_stats.md; exit 0; }
# the step on every form of the batch (gdx_query_layout_t), each with its own live PMC passes
for f in ascii uniform packed; do
  python3 bench.py --steps 20 --warmup 5 --input $f --no-extras --no-cpu-baseline --no-bandwidth --secondary-depth 0 \
    --side-file $OUT/side_$f.json > $OUT/bench_input_$f.json 2> $OUT/bench_input_$f.err
done
python3 tests/parity_sweep.py 300 21 > $OUT/parity_sweep_seed21.json 2> $OUT/parity21.err
tail -3 $OUT/gpu_tests_final.log; tail -c 300 $OUT/parity_sweep_seed21.json; echo; head -12 $OUT/bench_hg38_final_kernel_stats.md
# round 5: the step on 100 / 50 / 25 / 12.5 M reads (split and fused, the gather wires' pack / split kernels), the launches of
# the last steps with the gaps between them, and the host-pointer pipeline's threads with the results crossing PCIe as the
# found-bitmap wire (default) and written by the device
python3 tools/exp_shard_step.py 20 > $OUT/shard_step.json 2> $OUT/shard_step.err
bash tools/trace_timeline.sh $1/timeline 140 tools/exp_shard_step.py 3 > /dev/null
python3 tools/host_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/host_timing_wire.log
GDX_HOST_RESULTS=dma python3 tools/host_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/host_timing_dma.log
grep "rep 2" $OUT/host_timing_wire.log $OUT/host_timing_dma.log
