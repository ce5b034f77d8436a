#!/bin/bash
set -u
O=gpurun_out/r05f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_partition.py tests/test_gpu_step.py tests/test_matcher.py tests/test_tracker.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; echo "rc $rc"; tail -6 $O/tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --mode partition --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_partition.json 2> $O/bench_partition.err; echo "rc $?"
PWN_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 300 python bench.py --mode partition --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_partition_forced.json 2> $O/bench_partition_forced.err; echo "rc $?"
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config5 --no-tracker > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?"
python - <<'PY'
import json
for f in ("bench_partition","bench_partition_forced","bench_default"):
    l=json.loads(open(f"gpurun_out/r05f/{f}.json").read().strip().splitlines()[-1])
    print(f, "%.0f/s %.3f ms" % (l["value"], l["ms_per_step"]), "path %.3f" % l["roofline"]["path_frac"], {k: round(v,3) for k,v in l["stage_ms_per_step"].items() if v})
    if "closure_match_batch" in l: print("   closure_match_batch", l["closure_match_batch"]["alignments_per_s"], l["closure_match_batch"]["frac_of_peak"], "align_only", l["align_only"]["alignments_per_s"])
    if l.get("multi_gpu"): print("   collectives", l["multi_gpu"]["collectives_alone"])
PY
