#!/bin/bash
# round 2, first GPU pass: the GPU test suite with the new jump format / lazy tails / cursor strings, then A/B of the
# record path against the round-1 array path
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a/pytest.log
cat gpurun_out/r2a/pytest.log
for path in records arrays; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --secondary-depth 0 --no-bandwidth --path $path > gpurun_out/r2a/bench_$path.json 2> gpurun_out/r2a/bench_$path.err
  tail -3 gpurun_out/r2a/bench_$path.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2a/bench_$path.json"))
print("$path", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["locate_roofline"]["avg_launch_ms"], d["roofline"]["line_fetches_per_query"], d["parity"])
PY
done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --secondary-depth 0 --no-bandwidth --op count > gpurun_out/r2a/bench_count.json 2> gpurun_out/r2a/bench_count.err
python -c "
import json
d=json.load(open('gpurun_out/r2a/bench_count.json'))
print('count', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])
"
