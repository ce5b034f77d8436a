#!/bin/bash
# second soak of round 5: other seeds than every earlier run (scenes, sensor offsets, converter settings, noise), both omega storages
O=gpurun_out/r05_soak2; mkdir -p $O
(echo "== tools/soak_parity.py --seed0 5000 --small 240 --vga 64 --omega-storage sym6"; timeout -k 10 1000 python tools/soak_parity.py --seed0 5000 --small 240 --vga 64 --omega-storage sym6) > $O/soak_sym6.txt 2>&1; echo "rc $?"; tail -2 $O/soak_sym6.txt
(echo "== tools/soak_parity.py --seed0 7000 --small 240 --vga 64 (exact9)"; timeout -k 10 1000 python tools/soak_parity.py --seed0 7000 --small 240 --vga 64) > $O/soak_exact9.txt 2>&1; echo "rc $?"; tail -2 $O/soak_exact9.txt
