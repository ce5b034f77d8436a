#!/bin/bash
# record digests for the sources in the tree (three files); first the check against the committed ones (older sources: must still be equal)
set -u
O=gpurun_out/r05crc; mkdir -p $O
B="--no-cpu-baseline --no-latency --no-extras --no-profile"
timeout -k 10 400 python bench.py --gpus 1 --total-pairs 1024 --steps 1 --warmup 0 $B > $O/check.json 2> $O/check.err; python -c "
import json; l=json.loads(open('$O/check.json').read().strip().splitlines()[-1]); print('pairs sym6 vs committed digests:', l['gather']['records_vs_single_gpu_run']['equal'], l['gather']['records_vs_single_gpu_run']['file_is_for_these_kernels'])"
timeout -k 10 400 python bench.py --gpus 1 --total-pairs 1024 --steps 1 --warmup 0 $B --write-records-crc > $O/w1.json 2> $O/w1.err && \
timeout -k 10 400 python bench.py --gpus 1 --total-pairs 1024 --steps 1 --warmup 0 $B --omega-storage exact9 --write-records-crc > $O/w2.json 2> $O/w2.err && \
timeout -k 10 400 python bench.py --gpus 1 --mode partition --total-pairs 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --write-records-crc > $O/w3.json 2> $O/w3.err
echo "rc $?"; cp profiles/records_crc*.json $O/
