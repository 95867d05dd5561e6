"""Summarise a rocprofv3 --pmc --kernel-trace CSV directory: per kernel, the per-launch mean of every
counter and the median duration.  Usage: python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import os
import statistics
import sys
from collections import defaultdict

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if pat in k:
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if pat in k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(cnt):
    med = statistics.median(dur[k]) if dur[k] else float("nan")
    print("## %s  (%d launches, median %.1f us under the profiler)" % (k, len(dur[k]), med))
    print("| counter | per launch |\n|---|---|")
    for c in sorted(cnt[k]):
        v = cnt[k][c]
        print("| %s | %.4g |" % (c, sum(v) / len(v)))
    print()
