#!/bin/bash
# upper bound of an asynchronous step (submit step k+1 before step k is waited for): a variant whose one-submission step does not wait at all
set -u
O=gpurun_out/r05k; mkdir -p $O
A="--steps 25 --warmup 3 --no-cpu-baseline --no-latency --no-extras --no-profile"
for rep in 1 2 3; do
for v in base ahead1; do
  PWN_HIP_LIB=$PWD/build/variants/$v.so timeout -k 10 200 python bench.py $A > $O/${v}_$rep.json 2> $O/err.txt
  python -c "
import json; l=json.loads(open('$O/${v}_$rep.json').read().strip().splitlines()[-1]); print('$v rep $rep: %.0f/s %.3f ms' % (l['value'], l['ms_per_step']))"
done
done
