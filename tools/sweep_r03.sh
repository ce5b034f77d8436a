#!/bin/bash
# on the GPU box: headline step under a few batching configurations (each line: sub_frames sub_pairs streams -> alignments/s, ms/step)
mkdir -p gpurun_out
for cfg in "64 64 2" "128 64 2" "32 64 2" "64 128 2" "64 32 2" "64 64 3" "32 32 4" "128 128 1" "64 64 2"; do
  set -- $cfg
  timeout 200 python bench.py --pairs 128 --steps 8 --warmup 2 --sub-frames $1 --sub-pairs $2 --streams $3 --no-cpu-baseline --no-latency --no-extras --no-profile > gpurun_out/sw.json 2>gpurun_out/sw.err || { echo "$cfg FAILED"; tail -2 gpurun_out/sw.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print('sub_frames $1 sub_pairs $2 streams $3 ->', round(d['value']), 'alignments/s', round(d['ms_per_step'],3), 'ms/step')"
done
