#!/bin/bash
# round 6: PMC + kernel-trace passes of the final kernels -- the headline step (pairs) and the partition step (VERDICT r5 missing #6)
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/profile_pmc.sh gpurun_out/pmc_v17 > gpurun_out/pmc_v17.log 2>&1; echo "pairs rc $?"; tail -3 gpurun_out/pmc_v17.log
bash tools/profile_pmc.sh gpurun_out/pmc_v17_partition --mode partition --pairs 128 --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-profile --no-extras --render-workers 1 --streams 1 > gpurun_out/pmc_v17_partition.log 2>&1; echo "partition rc $?"; tail -3 gpurun_out/pmc_v17_partition.log
find gpurun_out/pmc_v17 gpurun_out/pmc_v17_partition -name "*.csv" -size +8M -delete
