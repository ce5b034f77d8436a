#!/bin/bash
# on the GPU box: single-pair latency (Python mirror, 30 alignments each, three rounds) per library variant in build/variants
for rep in 1 2 3; do
for lib in build/variants/*.so; do
  PWN_HIP_LIB=$PWD/$lib timeout 300 python tools/ab_latency.py 2>/dev/null | grep "profiling False" | sed "s|^|$lib |"
done
done
