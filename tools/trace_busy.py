"""GPU busy fraction and per-kernel overlap from a rocprofv3 kernel trace (csv): python tools/trace_busy.py <kernel_trace.csv> [skip_first_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in rows)
t0 = ev[0][0]
# steady state: the last 60 % of the trace
lo = t0 + int((ev[-1][1] - t0) * 0.4)
ev = [e for e in ev if e[0] >= lo]
span = ev[-1][1] - ev[0][0]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
gaps = []
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e - t0)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = {}
for s, e, n in ev:
    tot[n] = tot.get(n, 0) + (e - s)
print(f"span {span/1e6:.2f} ms, some kernel running {busy/span*100:.1f} %, sum of kernel durations / span = {sum(tot.values())/span:.2f}")
gaps.sort(reverse=True)
print("largest idle gaps (us):", [round(g[0] / 1e3, 1) for g in gaps[:12]], "count", len(gaps), "total idle ms", round(sum(g[0] for g in gaps) / 1e6, 3))
for n, v in sorted(tot.items(), key=lambda x: -x[1])[:8]:
    print(f"  {n:42s} {v/1e6:8.2f} ms")
