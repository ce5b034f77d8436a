#!/bin/bash
# on the GPU box: converter wall time + whole step for every library variant in build/variants (tools/ab_build.py), fused converter on;
# first line = the two-kernel path of the first variant.  usage: bash tools/ab_fused.sh [reps]
mkdir -p gpurun_out
REPS=${1:-10}
first=1
for lib in build/variants/*.so; do
  if [ $first = 1 ]; then
    echo "== $lib two-kernel"; PWN_HIP_LIB=$PWD/$lib timeout 200 python tools/exp_convert_modes.py $REPS PWN_FUSED_CONVERT=0 2>&1 | grep mode | head -1; first=0
  fi
  echo "== $lib fused"; PWN_HIP_LIB=$PWD/$lib timeout 200 python tools/exp_convert_modes.py $REPS PWN_FUSED_CONVERT=1 2>&1 | grep "mode\|Error\|error" | head -2
done
