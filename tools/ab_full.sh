#!/bin/bash
# A/B of prebuilt library variants (build/variants/*.so) on the full bench step (convert + align), one process per variant
for rep in 1 2; do
for lib in build/variants/*.so; do
  PWN_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency > gpurun_out/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/b.json')); s=d['stage_ms_per_step']; print('$lib', round(d['value']), {k: round(v,2) for k,v in s.items()})"
done
done
