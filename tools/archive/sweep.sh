#!/bin/bash
# usage: tools/sweep.sh "<ENV assignments>" ... ; runs the count-only hg38 bench once per setting
# prints: setting, M queries/s, search ms, found fraction, line fetches per query, active-lane fraction
for cfg in "$@"; do
  env $cfg python bench.py --op count --steps 3 --no-cpu-baseline --no-bandwidth --secondary-depth 0 ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg', round(d['value']/1e6), round(r['avg_launch_ms'],1), d['parity']['found_fraction'], round(r['line_fetches_per_query'] or 0,2), round(r['active_lane_fraction'] or 0,3))"
done
