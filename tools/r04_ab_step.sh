#!/bin/bash
# on the GPU box: same-box A/B of the bench step as one submission (fused) and as two calls with a host wait between them (split); two rounds
mkdir -p gpurun_out
for rep in 1 2 3; do
for mode in split fused; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-extras --no-profile --step-mode $mode $AB_ARGS > gpurun_out/ab_step_${mode}_$rep.json 2>gpurun_out/b.err || { echo "$mode FAILED"; tail -3 gpurun_out/b.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/ab_step_${mode}_$rep.json')); print('$mode', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'path_frac', round(d['path_roofline']['frac_of_peak'],4), 'gather', d['gather']['records_equal_local'], d['gather']['records_vs_single_gpu_run'].get('note','')[:40])"
done
done
