#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output into the small summaries kept under profiles/.

    tools/summarize_rocprof.py stats  <dir with *_kernel_stats.csv>         -> markdown table on stdout
    tools/summarize_rocprof.py pmc    <dir with *_counter_collection.csv>   -> json on stdout
    tools/summarize_rocprof.py timeline <dir with *_kernel_trace.csv> [n]   -> the last n dispatches: start, duration, gap before
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"(gdx::[A-Za-z0-9_]+(?:<[^>(]*>)?)", name)
    if m:
        return m.group(1)
    m = re.search(r"rocprim::.*?wrapped_([a-z_]+)_config", name)
    if m:
        return "rocprim::" + m.group(1)
    return name.split("(")[0][-60:]


def stats(d):
    f = glob.glob(f"{d}/**/*_kernel_stats.csv", recursive=True)[0]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0])
        a[0] += int(r["Calls"])
        a[1] += int(r["TotalDurationNs"])
    total = sum(v[1] for v in agg.values())
    print("| kernel | calls | total ms | avg ms | % |")
    print("|---|---|---|---|---|")
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25  # rows shown (by total time)
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"| {k} | {c} | {t / 1e6:.3f} | {t / c / 1e6:.4f} | {100 * t / total:.2f} |")


def pmc(d):
    out = {}
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            out.setdefault(c, {})[k] = {"launches": n, "sum": s, "per_launch": s / n}
    print(json.dumps(out, indent=1))


def timeline(d):
    f = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))))
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = rows[-n:]
    t0, prev_end = rows[0][0], None
    print("| start us | dur us | gap before us | kernel |")
    print("|---|---|---|---|")
    for s, e, k in rows:
        gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:.1f}"
        print(f"| {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap} | {k} |")
        prev_end = e if prev_end is None else max(prev_end, e)


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "timeline": timeline}[sys.argv[1]](sys.argv[2])
