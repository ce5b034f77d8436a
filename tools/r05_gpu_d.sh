#!/bin/bash
# A/B of k_stats' panel-major block order at 1280x960 (same box, alternating), + one FETCH_SIZE pass of the new order
set -u
O=gpurun_out/r05d; mkdir -p $O
A="--rows 960 --cols 1280 --pairs 32 --sub-pairs 32 --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-extras"
for rep in 1 2; do
  PWN_HIP_LIB=$PWD/build/variants/r05_rowmajor_stats.so timeout -k 10 200 python bench.py $A > $O/rowmajor_$rep.json 2> $O/rowmajor_$rep.err
  timeout -k 10 200 python bench.py $A > $O/panel_$rep.json 2> $O/panel_$rep.err
done
python - <<'PY'
import json
for n in ("rowmajor_1","panel_1","rowmajor_2","panel_2"):
    l=json.loads(open(f"gpurun_out/r05d/{n}.json").read().strip().splitlines()[-1])
    print(n, "pairs/s %.1f" % l["value"], "ms/step %.3f" % l["ms_per_step"], "stats ms/step %.3f" % l["stage_ms_per_step"]["stats"], "path_frac %.3f" % l["roofline"]["path_frac"])
PY
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --rows 960 --cols 1280 --pairs 32 --sub-pairs 32 --steps 1 --warmup 1 --no-cpu-baseline --no-latency --no-profile --no-extras --render-workers 1 --streams 1 > $O/fetch.json 2> $O/fetch.err
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("gpurun_out/r05d/fetch/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"]=="FETCH_SIZE":
            k=row["Kernel_Name"].split("(")[0][-40:]; a=acc[k]; a[0]+=float(row["Counter_Value"]); a[1]+=1
for k,(s,n) in acc.items():
    if "k_stats" in k or "probe_read" in k: print(k, "FETCH_SIZE mean/dispatch %.1f KB" % (s/n), n)
PY
