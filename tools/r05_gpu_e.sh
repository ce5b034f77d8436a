#!/bin/bash
# round 5: the profile set of the switch-free sources (v16): PMC passes + kernel trace of the default step, one process with both clocks, the
# partition step traced, the partition line with the collectives forced through RCCL at world size 1, the soak parity run in both omega storages
set -u
O=gpurun_out/r05e; mkdir -p $O
echo "== PMC + trace, default step (serial)"; date
bash tools/profile_pmc.sh gpurun_out/r05_pmc_vga > $O/pmc_vga.txt 2>&1; echo "rc $?"; tail -3 $O/pmc_vga.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== both clocks in one process"; date
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_same_run -- python3 bench.py --pairs 128 --steps 5 --warmup 1 --streams 1 --no-cpu-baseline --no-latency --no-extras --render-workers 1 > $O/bench_same_run.json 2> $O/bench_same_run.err; echo "rc $?"
echo "== partition step traced (serial)"; date
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_partition_trace -- python3 bench.py --mode partition --pairs 128 --steps 5 --warmup 1 --streams 1 --no-cpu-baseline --render-workers 1 > $O/bench_partition_traced.json 2> $O/bench_partition_traced.err; echo "rc $?"
echo "== partition line, collectives through RCCL at world size 1"; date
PWN_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 300 python bench.py --mode partition --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_partition_forced.json 2> $O/bench_partition_forced.err; echo "rc $?"
echo "== default line, gather through RCCL at world size 1 (multi_gpu keys)"; date
PWN_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_default_forced.json 2> $O/bench_default_forced.err; echo "rc $?"
echo "== soak parity"; date
(echo "== tools/soak_parity.py --small 150 --vga 40 --omega-storage sym6 (round 5, switch-free sources)"; timeout -k 10 700 python tools/soak_parity.py --small 150 --vga 40 --omega-storage sym6) > $O/soak_sym6.txt 2>&1; echo "rc $?"; tail -2 $O/soak_sym6.txt
(echo "== tools/soak_parity.py --small 150 --vga 40 (exact9; round 5, switch-free sources)"; timeout -k 10 700 python tools/soak_parity.py --small 150 --vga 40) > $O/soak_exact9.txt 2>&1; echo "rc $?"; tail -2 $O/soak_exact9.txt
date
