"""Prints the top kernels of the newest *kernel_stats.csv under a rocprofv3 output directory.
Usage: python3 tools/kstats.py <dir> [rows]"""
import csv
import glob
import os
import sys

ks = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)
if not ks:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
for r in list(csv.DictReader(open(max(ks, key=os.path.getmtime))))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-90s calls %5s  avg %10.1f us  min %10.1f  max %10.1f  %6s %%" % (
        r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
