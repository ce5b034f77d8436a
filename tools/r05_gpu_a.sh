#!/bin/bash
# round 5, first GPU call: (1) the kernel header without its compile-time switches gives the records of round 4 (old CRC file, other digest),
# (2) new tests, (3) new digests for the current sources, (4) default bench line, (5) partition line
set -u
O=gpurun_out/r05a; mkdir -p $O
B="--no-cpu-baseline --no-latency --no-extras --no-profile"
echo "== 1. 1024 pairs against the committed (round-4) digests"; date
timeout -k 10 600 python bench.py --gpus 1 --total-pairs 1024 --steps 1 --warmup 0 $B > $O/crc_vs_r04.json 2> $O/crc_vs_r04.err || { echo "crc check failed"; tail -5 $O/crc_vs_r04.err; exit 1; }
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05a/crc_vs_r04.json").read().strip().splitlines()[-1])
print("records vs round-4 digests:", l["gather"]["records_vs_single_gpu_run"])
PY
echo "== 2. new tests"; date
timeout -k 10 900 python -m pytest tests/test_partition.py tests/test_gpu_step.py -x -q -m gpu > $O/new_tests.txt 2>&1; echo "rc $?"; tail -15 $O/new_tests.txt
echo "== 3. digests for the current sources"; date
timeout -k 10 400 python bench.py --gpus 1 --total-pairs 1024 --steps 1 --warmup 0 $B --write-records-crc > $O/crc_write_sym6.json 2> $O/crc_write_sym6.err && \
timeout -k 10 400 python bench.py --gpus 1 --total-pairs 1024 --steps 1 --warmup 0 $B --omega-storage exact9 --write-records-crc > $O/crc_write_exact9.json 2> $O/crc_write_exact9.err && \
timeout -k 10 400 python bench.py --gpus 1 --mode partition --total-pairs 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --write-records-crc > $O/crc_write_partition.json 2> $O/crc_write_partition.err
echo "rc $?"; cp profiles/records_crc*.json $O/ 2>/dev/null; ls -la $O
echo "== 4. default bench line"; date
timeout -k 10 500 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?"
echo "== 5. partition line"; date
timeout -k 10 300 python bench.py --mode partition --steps 20 --warmup 3 > $O/bench_partition.json 2> $O/bench_partition.err; echo "rc $?"; tail -3 $O/bench_partition.err
python - <<'PY'
import json
for f in ("bench_default", "bench_partition"):
    try:
        l = json.loads(open(f"gpurun_out/r05a/{f}.json").read().strip().splitlines()[-1])
        print(f, l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"].get("path_frac"), l["gather"]["records_vs_single_gpu_run"])
    except Exception as e:
        print(f, "unreadable", e)
PY
date
