#!/bin/bash
# usage: tools/sweep.sh "<ENV assignments>" ... ; runs the count-only hg38 bench once per setting
for cfg in "$@"; do
  env $cfg python bench.py --op count --steps 3 --no-cpu-baseline --no-bandwidth --secondary-depth 0 ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value']/1e6), round(d['roofline']['avg_launch_ms'],1), d['parity']['found_fraction'])"
done
