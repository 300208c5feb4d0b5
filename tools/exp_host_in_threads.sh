#!/bin/bash
# the packing feeder's pool size (GDX_HOST_IN_THREADS) against the ASCII host calls' rates: tools/exp_host_in_threads.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
export GDX_EXP_BOTH=0
for t in default 8 10 12 14 16; do
  if [ $t = default ]; then unset GDX_HOST_IN_THREADS; else export GDX_HOST_IN_THREADS=$t; fi
  echo "== feeder pool: $t"
  python3 $R/tools/exp_host_ascii.py 2>&1 | grep "GDX_HOST_PACK=1" | grep -v "rep 0"
done
