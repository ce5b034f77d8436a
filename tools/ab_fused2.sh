#!/bin/bash
# fused converter variants x frames per launch.  usage: bash tools/ab_fused2.sh [reps]
REPS=${1:-10}
for sf in 64 32; do
for lib in build/variants/*.so; do
  echo "== $lib sub_frames $sf"; PWN_SUB_FRAMES=$sf PWN_HIP_LIB=$PWD/$lib timeout 200 python tools/exp_convert_modes.py $REPS PWN_FUSED_CONVERT=1 2>&1 | grep "mode\|Error\|error" | head -1
done
done
echo "== two-kernel"; PWN_HIP_LIB=$PWD/build/variants/w6.so timeout 200 python tools/exp_convert_modes.py $REPS PWN_FUSED_CONVERT=0 2>&1 | grep mode | head -1
