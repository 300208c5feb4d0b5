#!/usr/bin/env python3
import json, sys
d = json.load(open(sys.argv[1]))
pat = sys.argv[2] if len(sys.argv) > 2 else "search_kernel"
for c, v in d.items():
    for k, x in v.items():
        if pat in k:
            print(f"{c:34s} {k[:44]:44s} per_launch {x['per_launch']:.5g}  launches {x['launches']}")
