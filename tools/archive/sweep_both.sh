#!/bin/bash
# usage: tools/sweep_both.sh "<ENV assignments>" ... ; count+locate and count-only hg38 runs per setting
for cfg in "$@"; do
  env $cfg python bench.py --steps 3 --no-cpu-baseline --no-bandwidth --secondary-depth 0 ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', 'value', round(d['value']/1e6), 'step', round(d['ms_per_step'],1), 'search', round(d['roofline']['avg_launch_ms'],1), 'locate', round(d['locate_roofline']['avg_launch_ms'],2), 'iters', round(d['roofline']['line_fetches_per_query'],2), 'index GB', round(d['index_bytes']/1e9,1), 'build s', round(d['index_build_seconds'],1))"
done
