#!/bin/bash
# Infinity-Cache residency experiment: per-stage kernel time of the align kernels as a function of the pairs per launch
# (one stream, serial profiled pass).  A sub-batch small enough to keep its clouds in the 256 MiB Infinity Cache re-reads them
# on-die in iterations 2..10.
mkdir -p gpurun_out
for sp in 2 4 6 8 12 16 32 64; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --streams 1 --sub-pairs $sp 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('sub_pairs', $sp, 'value %.0f'%d['value'], 'serial %.0f'%d['roofline']['serial_pass_alignments_per_s'], 'project %.2f corr %.2f solve %.2f'%(s['project'],s['corr_linearize'],s['solve']))" | tee -a gpurun_out/ab_subpairs_ic.log
done
