#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSVs (one row per dispatch and counter) into per-kernel means per dispatch."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"\(.*", "", row.get("Kernel_Name", "")).replace("pwnhip::", "")
            c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0)
            a = acc[name][c]; a[0] += v; a[1] += 1
for name in sorted(acc):
    print(name)
    for c, (s, n) in sorted(acc[name].items()):
        print(f"   {c:24s} mean/dispatch {s / n:16.1f}   dispatches {n}")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("\nkernel stats:", f)
    print(open(f).read())
