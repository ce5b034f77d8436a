#!/bin/bash
set -u
O=gpurun_out/r05j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_step.py tests/test_handover.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/tests.txt
[ $rc -eq 0 ] || exit 1
A="--steps 25 --warmup 3 --no-cpu-baseline --no-latency --no-extras --no-profile"
for rep in 1 2 3; do
for v in base early; do
  PWN_HIP_LIB=$PWD/build/variants/$v.so timeout -k 10 200 python bench.py $A > $O/${v}_$rep.json 2> $O/err.txt
  python -c "
import json; l=json.loads(open('$O/${v}_$rep.json').read().strip().splitlines()[-1]); print('$v rep $rep: %.0f/s %.3f ms path %.4f' % (l['value'], l['ms_per_step'], l['roofline']['path_frac']), l['gather']['records_vs_single_gpu_run']['equal'])"
done
done
