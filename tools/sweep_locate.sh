#!/bin/bash
for cfg in "$@"; do
  env $cfg python bench.py --steps 3 --no-cpu-baseline --no-bandwidth --secondary-depth 0 ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', 'value', round(d['value']/1e6), 'step', round(d['ms_per_step'],1), 'search', round(d['roofline']['avg_launch_ms'],1), 'locate', round(d['locate_roofline']['avg_launch_ms'],2), d['parity']['hits_matching_text'])"
done
