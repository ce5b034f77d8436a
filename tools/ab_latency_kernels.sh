#!/bin/bash
# on the GPU box: tools/ab_latency.py (single pair: wall times + per-kernel device time) once per library variant in build/variants
for lib in build/variants/*.so; do
  echo "== $lib"; PWN_HIP_LIB=$PWD/$lib timeout 200 python tools/ab_latency.py 2>&1 | grep -v "^$" | tail -3
done
