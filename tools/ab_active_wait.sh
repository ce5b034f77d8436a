#!/bin/bash
# on the GPU box: one VGA pair (convert 2 frames + align) and the tracker stream under the HIP runtime's host-wait settings
mkdir -p gpurun_out
for env in "" "ROC_ACTIVE_WAIT_TIMEOUT=200" "ROC_ACTIVE_WAIT_TIMEOUT=2000" "ROC_CPU_WAIT_FOR_SIGNAL=0" "ROC_ACTIVE_WAIT_TIMEOUT=2000 ROC_SYSTEM_SCOPE_SIGNAL=0"; do
  for rep in 1 2; do
    echo "== ${env:-default} (run $rep)"
    env $env timeout -k 10 100 python tools/exp_single_pair_timeline.py 80 2>&1 | tail -1
  done
  echo "== ${env:-default}: tracker"
  env $env timeout -k 10 150 python tools/exp_index_shortcut.py 2>&1 | tail -1 | cut -c1-60
done
