#!/bin/bash
# round 6 soak: the wide parity net with the callers' decisions on both sides (tools/soak_parity.py); these are round 5's second-soak seeds (the case with the 2.6e-4 pose is among them)
set -u
O=gpurun_out/r06soak; mkdir -p $O
timeout -k 10 1000 python tools/soak_parity.py --omega-storage sym6 --seed0 5000 --small 240 --vga 64 > $O/soak_sym6_seed5000.txt 2>&1; echo "rc $?"; tail -1 $O/soak_sym6_seed5000.txt
