#!/bin/bash
# sweep of the converter sub-batch size on the full bench step
for rep in 1 2; do
for sf in 32 64 128 256; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --sub-frames $sf > gpurun_out/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/b.json')); s=d['stage_ms_per_step']; print('sub_frames $sf', round(d['value']), {k: round(v,2) for k,v in s.items() if v})"
done
done
