#!/bin/bash
# round 5, second GPU call: new tests, full GPU suite, PMC passes at 1280x960 (BASELINE configs[4]; SURVEY.md 8(d) Config 5)
set -u
O=gpurun_out/r05b; mkdir -p $O
echo "== new tests"; date
timeout -k 10 900 python -m pytest tests/test_partition.py tests/test_gpu_step.py -x -q -m gpu -s > $O/new_tests.txt 2>&1; rc=$?; echo "rc $rc"; tail -8 $O/new_tests.txt
[ $rc -eq 0 ] || exit 1
echo "== PMC at 1280x960"; date
bash tools/profile_pmc.sh gpurun_out/r05_pmc_k2 --rows 960 --cols 1280 --pairs 32 --sub-pairs 32 --steps 1 --warmup 1 --no-cpu-baseline --no-latency --no-profile --no-extras --render-workers 1 --streams 1 > $O/pmc_k2.txt 2>&1; echo "rc $?"; tail -5 $O/pmc_k2.txt
echo "== full GPU suite"; date
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; echo "rc $?"; tail -5 $O/gpu_suite.txt
date
