#!/bin/bash
# per-launch durations (in launch order) of the kernels of one of the tools/exp_*.py scripts whose name matches a pattern:
# tools/trace_launches.sh <out_dir_under_gpurun_out> <pattern> <script> [args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
PAT=$1; shift
SCRIPT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/$SCRIPT "$@" > $OUT/trace_out.json 2> $OUT/trace.err
python3 - "$OUT" "$PAT" <<'PY' > $OUT/launches.txt
import csv, glob, re, sys
out, pat = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r["Kernel_Name"]):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
for s, e, n in rows:
    # the kernel's own name with its template arguments: what stands in front of the parameter list, namespaces dropped
    # (the first "gdx::" of a kernel in the anonymous namespace is in its parameters)
    head = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?(?:gdx::)?([A-Za-z0-9_]+(?:<[^()]*>)?)\(", head)
    print(f"{(e - s) / 1e6:9.3f} ms  {m.group(1) if m else n[:80]}")
PY
find $OUT/trace -name '*.csv' -delete
tail -40 $OUT/launches.txt
