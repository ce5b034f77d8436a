#!/bin/bash
# On the GPU box: rewrite the three record-digest files for the kernel sources in the tree (profiles/README.md) and hand them back through gpurun_out/crc/
set -u
O=gpurun_out/crc; mkdir -p $O
for args in "" "--omega-storage exact9" "--mode partition"; do
  timeout -k 10 400 python bench.py --gpus 1 $args --total-pairs 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-extras --no-profile --write-records-crc > $O/line_$(echo "$args" | tr -d ' -').json 2> $O/err.txt || { echo "failed: $args"; tail -5 $O/err.txt; exit 1; }
done
cp profiles/records_crc*.json $O/
