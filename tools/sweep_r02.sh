#!/bin/bash
# sub-batch / stream sweep of the headline step (same box, one process per setting)
for cfg in "64 64 2" "64 64 3" "32 32 2" "32 32 4" "128 64 2" "64 32 2" "64 32 3" "128 128 2" "64 64 1"; do
  set -- $cfg
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-extras --no-profile --sub-frames $1 --sub-pairs $2 --streams $3 > gpurun_out/s.json 2>/dev/null || { echo "$cfg FAILED"; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/s.json')); print('sub_frames $1 sub_pairs $2 streams $3:', round(d['value']), 'alignments/s', round(d['ms_per_step'],2), 'ms')"
done
