#!/bin/bash
# round 5, validation of the tree: digests for the current sources, full GPU suite, final bench lines
set -u
O=gpurun_out/r05g; mkdir -p $O
B="--no-cpu-baseline --no-latency --no-extras --no-profile"
echo "== digests"; date
timeout -k 10 400 python bench.py --gpus 1 --total-pairs 1024 --steps 1 --warmup 0 $B > $O/crc_check_before.json 2> $O/crc_check_before.err; python -c "
import json; l=json.loads(open('$O/crc_check_before.json').read().strip().splitlines()[-1]); print('pairs sym6 vs committed digests (older sources):', l['gather']['records_vs_single_gpu_run'])"
timeout -k 10 400 python bench.py --gpus 1 --total-pairs 1024 --steps 1 --warmup 0 $B --write-records-crc > $O/crc_write_sym6.json 2> $O/crc_write_sym6.err && \
timeout -k 10 400 python bench.py --gpus 1 --total-pairs 1024 --steps 1 --warmup 0 $B --omega-storage exact9 --write-records-crc > $O/crc_write_exact9.json 2> $O/crc_write_exact9.err && \
timeout -k 10 400 python bench.py --gpus 1 --mode partition --total-pairs 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --write-records-crc > $O/crc_write_partition.json 2> $O/crc_write_partition.err
echo "rc $?"; cp profiles/records_crc*.json $O/
echo "== full GPU suite"; date
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; echo "rc $?"; tail -5 $O/gpu_suite.txt
echo "== bench lines"; date
timeout -k 10 500 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?"
timeout -k 10 300 python bench.py --mode partition --steps 20 --warmup 3 > $O/bench_partition.json 2> $O/bench_partition.err; echo "rc $?"
python - <<'PY'
import json
for f in ("bench_default", "bench_partition"):
    l = json.loads(open(f"gpurun_out/r05g/{f}.json").read().strip().splitlines()[-1])
    print(f, "%.0f/s %.3f ms kernel %.3f path %.3f" % (l["value"], l["ms_per_step"], l["roofline"]["frac"], l["roofline"]["path_frac"]), l["gather"]["records_vs_single_gpu_run"]["equal"], l["gather"]["records_vs_single_gpu_run"]["file_is_for_these_kernels"])
PY
date
