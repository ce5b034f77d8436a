#!/usr/bin/env python3
"""Per-step device timeline of `bench.py --mode partition` from a rocprofv3 kernel trace (+ memory-copy trace when present):
python tools/partition_timeline.py <dir>  -- for the last steps: when the step's first / last match kernel ran, idle gaps, and every kernel or copy
that is not one of the match call's own (collectives, flat-cloud copies, the look-ahead conversion) with its start offset inside the step."""
import csv, glob, os, sys

d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-48:], r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))[:40], r.get("Stream_Id", "?"), "-"))
ev.sort()
own = ("k_project", "k_corr_linearize", "k_solve_update", "k_match_score", "k_pack_records", "k_resolve_cur")
packs = [i for i, e in enumerate(ev) if "k_pack_records" in e[2]]
print(f"{len(ev)} events, {len(packs)} steps")
for si in range(max(1, len(packs) - 3), len(packs)):
    a, b = packs[si - 1], packs[si]
    t0 = ev[a][1]                                 # end of the previous step's last kernel
    step = ev[a + 1:b + 1]
    mine = [e for e in step if any(k in e[2] for k in own)]
    other = [e for e in step if not any(k in e[2] for k in own)]
    first = min(e[0] for e in mine); last = max(e[1] for e in mine)
    # idle: union of all events
    cur = t0; idle = 0
    for s, e, *_ in sorted(step):
        if s > cur: idle += s - cur
        cur = max(cur, e)
    print(f"step {si}: {(last - t0) / 1e3:.1f} us from the previous pack's end to this pack's end; first own kernel at +{(first - t0) / 1e3:.1f} us; idle {idle / 1e3:.1f} us; own kernel time {sum(e[1] - e[0] for e in mine) / 1e3:.1f} us")
    for s, e, n, st, q in other:
        print(f"    +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:8.1f} us  stream {st} queue {q}  {n}")
