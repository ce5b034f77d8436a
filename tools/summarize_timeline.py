#!/usr/bin/env python3
"""Device timeline of the LAST repetition of tools/exp_single_pair_timeline.py from rocprofv3's kernel and memory-copy traces:
python tools/summarize_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [kernels per repetition]"""
import csv, glob, os, sys

d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))[:40]))
ev.sort()
# the last repetition starts at the last k_row_count (first kernel of the 2-frame convert)
starts = [i for i, e in enumerate(ev) if "k_row_count" in e[2] or "k_strip_count" in e[2]]
i0 = starts[-1]
rep = ev[i0:]
t0 = rep[0][0]
busy = 0; prev_end = t0
print(f"{'start us':>9} {'dur us':>8} {'gap us':>7}  what")
for s, e, name in rep:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {name}")
    busy += e - s; prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, device busy {busy / 1e3:.1f} us, {len(rep)} events")
