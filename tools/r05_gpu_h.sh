#!/bin/bash
# does the number of hardware queues the HIP runtime multiplexes its streams onto (GPU_MAX_HW_QUEUES, default 4) limit the overlap of the step's four streams?
set -u
O=gpurun_out/r05h; mkdir -p $O
A="--steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-extras --no-profile"
for rep in 1 2; do
for q in default 2 8 16; do
  if [ $q = default ]; then timeout -k 10 200 python bench.py $A > $O/q_${q}_$rep.json 2> $O/err.txt
  else GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py $A > $O/q_${q}_$rep.json 2> $O/err.txt; fi
  python -c "
import json; l=json.loads(open('$O/q_${q}_$rep.json').read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES=$q rep $rep: %.0f/s %.3f ms path %.3f' % (l['value'], l['ms_per_step'], l['roofline']['path_frac']))"
done
done
