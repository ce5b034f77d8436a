#!/bin/bash
# A/B of prebuilt library variants (build/variants/*.so) with bench.py, one process per variant
for lib in build/variants/*.so; do
  PWN_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --align-only > gpurun_out/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/b.json')); s=d['stage_ms_per_step']; print('$lib', round(d['value']), round(d['align_only_alignments_per_s']), 'project', round(s['project'],2), 'corr_lin', round(s['corr_linearize'],2), 'chi2', round(d['counters_mean']['chi2_final'],3))"
done
