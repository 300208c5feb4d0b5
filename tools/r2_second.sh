#!/bin/bash
mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2b/pytest.log
cat gpurun_out/r2b/pytest.log
tools/pmc_r2.sh r2b/pmc_records --path records
tools/pmc_r2.sh r2b/pmc_arrays --path arrays
python - <<'PY'
import json
for p in ("records","arrays"):
    d=json.load(open(f"gpurun_out/r2b/pmc_{p}/summary.json"))
    for c,v in d.items():
        for k,x in v.items():
            if "search_pair" in k or "locate_queue" in k:
                print(p,c,k[:40],x["launches"],x["per_launch"])
PY
