#!/bin/bash
# PMC passes for bench.py (each counter group in its own rocprofv3 run, no trace domains mixed in).
# usage (on the GPU box, from the repo root): bash tools/profile_pmc.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
# no child processes under rocprofv3 (its preloaded library has initialised the GPU): no CPU baseline child, frames rendered in-process
ARGS=${@:---pairs 128 --steps 1 --warmup 1 --no-cpu-baseline --no-latency --no-profile --no-extras --render-workers 1 --streams 1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift; echo "pass $name"; timeout -k 5 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "pass $name failed or timed out"; exit 1; }; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
echo "pass trace"; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/trace.json" 2> "$OUT/trace.err" || { echo "trace pass failed"; exit 1; }
python3 tools/summarize_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
