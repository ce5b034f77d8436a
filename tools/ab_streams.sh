#!/bin/bash
# streams x sub-batch sizes on the full bench step
for rep in 1 2; do
for cfg in "2 64 64" "3 64 64" "4 64 64" "3 32 64" "4 32 64" "4 32 32" "3 43 64" "2 32 64"; do
  set -- $cfg
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-profile --streams $1 --sub-pairs $2 --sub-frames $3 > gpurun_out/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('streams $1 sub_pairs $2 sub_frames $3: %.0f alignments/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
done
done
