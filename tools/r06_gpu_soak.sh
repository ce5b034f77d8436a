#!/bin/bash
# round 6 soak: the wide parity net with the callers' decisions on both sides (tools/soak_parity.py) -> profiles/r06_soak_parity_*.txt
# (seed 5000 = round 5's second-soak seeds; ~3.5 + 3.5 + 7 minutes on a one-GPU box, most of it the oracle on the host)
set -u
O=gpurun_out/r06soak; mkdir -p $O
timeout -k 10 700 python tools/soak_parity.py --omega-storage sym6 --seed0 1000 --small 120 --vga 24 > $O/soak_sym6_seed1000.txt 2>&1; echo "rc $?"; tail -1 $O/soak_sym6_seed1000.txt
timeout -k 10 700 python tools/soak_parity.py --omega-storage exact9 --seed0 9000 --small 120 --vga 24 > $O/soak_exact9_seed9000.txt 2>&1; echo "rc $?"; tail -1 $O/soak_exact9_seed9000.txt
timeout -k 10 1000 python tools/soak_parity.py --omega-storage sym6 --seed0 5000 --small 240 --vga 64 > $O/soak_sym6_seed5000.txt 2>&1; echo "rc $?"; tail -1 $O/soak_sym6_seed5000.txt
timeout -k 10 900 python tools/soak_parity.py --omega-storage sym6 --seed0 3000 --small 100 --vga 20 --check-fallback > $O/soak_sym6_seed3000_fallback.txt 2>&1; echo "rc $?"; tail -1 $O/soak_sym6_seed3000_fallback.txt
timeout -k 10 300 python tools/soak_case.py --seed0 3000 --size vga --seed 3014 --small 100 --omega-storage sym6 > $O/soak_case_vga3014.txt 2>&1; echo "rc $?"
