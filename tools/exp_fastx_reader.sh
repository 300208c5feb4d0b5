#!/bin/bash
# The mapped FASTQ reader alone on this box's CPUs (no GPU work): thread counts, tile sizes, the newline index on and off.
#   tools/exp_fastx_reader.sh [reads = 24000000]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-24000000}
F=/tmp/gdx_reader_bench.fq
python3 - "$N" "$F" <<'PY'
import sys, numpy as np
n, path, ln = int(sys.argv[1]), sys.argv[2], 50
rng = np.random.default_rng(1)
rec = np.empty((n, ln * 2 + 7), dtype=np.uint8)
rec[:, 0], rec[:, 1], rec[:, 2] = ord("@"), ord("r"), 10
rec[:, 3:3 + ln] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, ln), dtype=np.uint8)]
rec[:, 3 + ln], rec[:, 4 + ln], rec[:, 5 + ln] = 10, ord("+"), 10
rec[:, 6 + ln:6 + 2 * ln] = ord("I")
rec[:, 6 + 2 * ln] = 10
rec.tofile(path)
PY
g++ -std=c++17 -O3 -pthread -I $R/genedex_amd/csrc -I $R/include -o /tmp/fastx_reader_bench $R/tools/fastx_reader_bench.cpp || exit 1
nproc
for t in 8 12 16 24 32; do
  echo "== $t threads"; /tmp/fastx_reader_bench $F 8000000 $t | tail -1
  echo "== $t threads, no newline index"; GDX_FASTX_NEWLINE_INDEX=0 /tmp/fastx_reader_bench $F 8000000 $t | tail -1
done
for b in 524288 1048576 4194304 8388608; do
  echo "== 16 threads, tiles of $b bytes"; GDX_FASTX_BLOCK_BYTES=$b /tmp/fastx_reader_bench $F 8000000 16 | tail -1
done
echo "== 16 threads, no MADV_POPULATE_READ"; GDX_FASTX_POPULATE=0 /tmp/fastx_reader_bench $F 8000000 16 | tail -1
echo "== stages, 16 threads"; GDX_FASTX_TIMING=1 /tmp/fastx_reader_bench $F 8000000 16 1>/dev/null 2>/tmp/t.err; tail -3 /tmp/t.err
echo "== stages, 16 threads, no newline index"; GDX_FASTX_NEWLINE_INDEX=0 GDX_FASTX_TIMING=1 /tmp/fastx_reader_bench $F 8000000 16 1>/dev/null 2>/tmp/t.err; tail -3 /tmp/t.err
if [ -f $R/tools/tmp_old/fastx_old.hpp ]; then
  g++ -std=c++17 -O3 -pthread -DFASTX_HEADER='"'$R/tools/tmp_old/fastx_old.hpp'"' -I $R/genedex_amd/csrc -I $R/include -o /tmp/fastx_reader_bench_old $R/tools/fastx_reader_bench.cpp
  for t in 8 16 32; do echo "== round-6 first reader (one block per thread), $t threads"; /tmp/fastx_reader_bench_old $F 8000000 $t | tail -1; done
fi
rm -f $F
