#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSVs (one row per dispatch and counter) into per-kernel means per dispatch, and writes the per-kernel HBM
traffic table bench.py reads (profiles/traffic.json).

  tools/summarize_pmc.py <pmc output dir>                                   -> the summary text on stdout (tools/profile_pmc.sh)
  tools/summarize_pmc.py <dir or committed summary .txt> --traffic-json profiles/traffic.json --version r03_v13 [--launch-items 64]
  ... --size-key 960x1280 --pixels 1228800 --launch-items 32 --frame-items 64     -> the table of another frame size, stored under "sizes"
  ... --mode-key partition --launch-items 64                                      -> the table of another workload (bench.py --mode partition), stored under "modes"

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read
(MI355X_MICROARCH.md, HBM section; re-checked in every set of passes on k_probe_read, which streams exactly 2 GiB per launch)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

KERNELS = ("k_corr_linearize", "k_stats", "k_unproject_integral", "k_project", "k_strip_count", "k_row_offsets", "k_solve_update",
           "k_match_score", "k_pack_records", "k_probe_read", "k_probe_copy")


def short(name):
    return re.sub(r"^void ", "", re.sub(r"[<(].*", "", name)).replace("pwnhip::", "").strip()


def collect_csv(out):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = short(row.get("Kernel_Name", ""))
                c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0)
                a = acc[name][c]; a[0] += v; a[1] += 1
    means = {k: {c: (s / n, n) for c, (s, n) in v.items()} for k, v in acc.items()}
    stats = {}
    stats_text = []
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        stats_text.append((f, open(f).read()))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                stats[short(row["Name"])] = (float(row["AverageNs"]), int(row["Calls"]))
    return means, stats, stats_text


def collect_txt(path):
    """the same tables from a committed summary text (profiles/*_pmc_summary_*.txt)"""
    means, stats = defaultdict(dict), {}
    cur = None
    in_stats = False
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith("kernel stats:"):
            in_stats = True; continue
        if in_stats:
            if line.startswith('"Name"') or not line.strip():
                continue
            row = next(csv.reader([line]))
            if len(row) >= 4:
                try:
                    stats[short(row[0])] = (float(row[3]), int(row[1]))
                except ValueError:
                    pass
            continue
        m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([0-9.eE+-]+)\s+dispatches\s+(\d+)", line)
        if m and cur is not None:
            means[cur][m.group(1)] = (float(m.group(2)), int(m.group(3)))
        elif line and not line.startswith(" "):
            cur = short(line)
    return means, stats, []


FRAME_KERNELS = ("k_stats", "k_unproject_integral", "k_strip_count", "k_row_offsets")      # launched per sub-batch of FRAMES, the others per sub-batch of pairs


def traffic_table(means, stats, version, source, launch_items, pixels, frame_items=None):
    kernels = {}
    for k in KERNELS:
        m = means.get(k)
        if not m or "FETCH_SIZE" not in m or "WRITE_SIZE" not in m:
            continue
        fetch_kb, write_kb = m["FETCH_SIZE"][0], m["WRITE_SIZE"][0]
        e = {"fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
             "dispatches": m["FETCH_SIZE"][1], "items_per_launch": (frame_items or launch_items) if k in FRAME_KERNELS else launch_items}
        if k in stats:
            e["avg_ns_rocprof"] = stats[k][0]
        for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_BUSY_CU_CYCLES", "SQ_INST_CYCLES_VMEM", "SQ_WAVE_CYCLES"):
            if c in m:
                e[c] = m[c][0]
        kernels[k] = e
    t = {"version": version, "source": source,
         "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE reads 1/2 on gfx950 (MI355X_MICROARCH.md, HBM section); confirmed in the same "
                       "passes on k_probe_read, which streams exactly 2 GiB per launch",
         "items_per_launch": launch_items, "pixels_per_frame": pixels, "kernels": kernels}
    if "k_corr_linearize" in kernels:
        t["k_corr_linearize_bytes_per_launch"] = kernels["k_corr_linearize"]["bytes_per_launch"]
        t["k_corr_linearize_bytes_per_pair_iteration"] = kernels["k_corr_linearize"]["bytes_per_launch"] / launch_items
    if "k_probe_read" in kernels:
        t["k_probe_read_check"] = {"bytes_streamed": 2147483648, "bytes_from_counters": kernels["k_probe_read"]["bytes_per_launch"]}
    return t


def main(argv):
    src = argv[1]
    opts = {argv[i]: argv[i + 1] for i in range(2, len(argv) - 1) if argv[i].startswith("--")}
    if os.path.isdir(src):
        means, stats, stats_text = collect_csv(src)
    else:
        means, stats, stats_text = collect_txt(src)
    if "--traffic-json" in opts:
        t = traffic_table(means, stats, opts.get("--version", "unversioned"), opts.get("--source", src),
                          int(opts.get("--launch-items", 64)), int(opts.get("--pixels", 307200)),
                          int(opts["--frame-items"]) if "--frame-items" in opts else None)
        old = {}
        if os.path.exists(opts["--traffic-json"]):
            try:
                old = json.load(open(opts["--traffic-json"]))
            except Exception:
                old = {}
        if "--mode-key" in opts:
            # the same kernels under another workload (e.g. --mode partition: every pair of a launch gathers from the SAME reference cloud, so the
            # caches serve most of what the algorithmic count charges): beside the pairs table, which stays the top level
            old.setdefault("modes", {})[opts["--mode-key"]] = t
            with open(opts["--traffic-json"], "w") as f:
                json.dump(old, f, indent=1)
            print("wrote", opts["--traffic-json"], "modes[%s]" % opts["--mode-key"], "version", t["version"], "kernels", sorted(t["kernels"]))
            return
        if "--size-key" in opts:
            # a table measured at another frame size (e.g. 960x1280: BASELINE configs[4]) goes beside the VGA table, which stays the top level
            old.setdefault("sizes", {})[opts["--size-key"]] = t
            with open(opts["--traffic-json"], "w") as f:
                json.dump(old, f, indent=1)
            print("wrote", opts["--traffic-json"], "sizes[%s]" % opts["--size-key"], "version", t["version"], "kernels", sorted(t["kernels"]))
            return
        if "sizes" in old:
            t["sizes"] = old["sizes"]
        if "modes" in old:
            t["modes"] = old["modes"]
        hist = dict(old.get("history", {}))
        if old.get("k_corr_linearize_bytes_per_pair_iteration") and old.get("version", "r02_v11") != t["version"]:
            hist[old.get("version", "r02_v11") + "_k_corr_linearize_bytes_per_pair_iteration"] = old["k_corr_linearize_bytes_per_pair_iteration"]
        t["history"] = hist
        with open(opts["--traffic-json"], "w") as f:
            json.dump(t, f, indent=1)
        print("wrote", opts["--traffic-json"], "version", t["version"], "kernels", sorted(t["kernels"]))
        return
    for name in sorted(means):
        print(name)
        for c, (m, n) in sorted(means[name].items()):
            print(f"   {c:24s} mean/dispatch {m:16.1f}   dispatches {n}")
    for f, txt in stats_text:
        print("\nkernel stats:", f)
        print(txt)


if __name__ == "__main__":
    main(sys.argv)
