#!/bin/bash
# band height of k_unproject_integral (PWN_IR_ROWS variants under build/variants) x converter sub-batch size
for rep in 1 2; do
for lib in build/variants/*.so; do
for sf in 64 128; do
  PWN_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --sub-frames $sf > gpurun_out/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/b.json')); s=d['stage_ms_per_step']; print('$lib sub_frames $sf', round(d['value']), 'unproject', round(s['unproject'],2), 'integral', round(s['integral'],2), 'stats', round(s['stats'],2))"
done
done
done
