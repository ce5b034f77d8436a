#!/bin/bash
# on the GPU box: same-box A/B of the bench step with exact9 and sym6 storage of the point information matrices (two rounds each)
mkdir -p gpurun_out
for rep in 1 2; do
for mode in exact9 sym6; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-extras --omega-storage $mode > gpurun_out/ab_omega_${mode}_$rep.json 2>gpurun_out/b.err || { echo "$mode FAILED"; tail -3 gpurun_out/b.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/ab_omega_${mode}_$rep.json')); s=d['stage_ms_per_step']; print('$mode', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k: round(v,3) for k,v in s.items() if v}, 'path_frac', round(d['path_roofline']['frac'],4), 'dom frac', round(d['roofline']['frac'],4), 'chi2', d['counters_mean']['chi2_final'])"
done
done
