#!/bin/bash
# PMC exploration of the converter's kernels (each counter group its own rocprofv3 pass).  usage: bash tools/pmc_convert.sh <outdir>
OUT=${1:-gpurun_out/pmc_conv}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift; echo "pass $name: $@"; timeout -k 5 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/pmc_convert.py 3 > "$OUT/$name.log" 2>&1 || echo "pass $name failed (rc $?)"; }
# at most four counters of one hardware block per pass ("Request exceeds the capabilities of the hardware" otherwise)
run ea1  TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL
run ea2  TCC_TAG_STALL TCC_BUSY TCC_CYCLE TCC_REQ
run req  TCC_READ TCC_WRITE TCC_HIT TCC_MISS
run ta1  TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES
run sqm  SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM
run tcp1 TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_WRITE_TAGCONFLICT_STALL_CYCLES
run tlb  TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS
run sq   SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
python3 tools/summarize_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -A8 "^k_stats\|^k_unproject_integral$" "$OUT/summary.txt" | head -150
