#!/bin/bash
# on the GPU box: single-pair latency and tracker rate once per library variant in build/variants, two rounds
mkdir -p gpurun_out
for rep in 1 2; do
for lib in build/variants/*.so; do
  PWN_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --tracker-frames 120 > gpurun_out/b.json 2>gpurun_out/b.err || { echo "$lib FAILED"; tail -3 gpurun_out/b.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$lib', 'single pair ms', round(d['single_pair_latency_ms'],4), 'tracker f/s', round(d['tracker_config2']['frames_per_s'],1), 'solve ms/step', round(d['stage_ms_per_step']['solve'],3))"
done
done
