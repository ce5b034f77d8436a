#!/bin/bash
# Infinity-Cache residency of the aligner's sub-batches: split step (converter at 64 frames per launch), alignment cut into small sub-batches so that the
# clouds of the pairs in flight fit the 256 MiB last-level cache across the ten iterations.  Same box, one process per setting.
mkdir -p gpurun_out
out=gpurun_out/r05_subpairs_ic.txt
: > $out
for cfg in "64 4" "8 1" "8 2" "4 1" "4 2" "4 4" "2 4" "16 2" "64 4"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-extras --step-mode split --sub-pairs $1 --streams $2 > gpurun_out/b.json 2> gpurun_out/b.err || { echo "sub $1 streams $2 FAILED" >> $out; tail -3 gpurun_out/b.err >> $out; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/b.json')); s=d['stage_ms_per_step']; print('sub_pairs $1 streams $2:', round(d['value']), '/s  ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in s.items() if v}, 'chi2', round(d['counters_mean']['chi2_final'],3))" >> $out
done
cat $out
