#!/bin/bash
# PMC passes for the bench workload (run on the GPU box via gpurun).  usage: tools/pmc_passes.sh <outdir> [bench args...]
# (TA_*/TCP_* derived counters crash rocprofv3 7.2 on this pool: not collected.)
# Counters are collected in their own runs (no tracing flags), a few per pass (SQ 8 slots, TCC 4, FETCH_SIZE 3, WRITE_SIZE 2).
set -u
export TMPDIR=/tmp
out=$1; shift
mkdir -p "$out"
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -- python3 bench.py --steps 1 --warmup 0 --cpu-pairs 0 --no-configs --no-profile --no-verify --no-api-loop "$@" > "$out/p$i.log" 2>&1 || echo "pass $i failed: $set"
done <<'SETS'
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CU_CYCLES
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
FETCH_SIZE
WRITE_SIZE GRBM_GUI_ACTIVE
SETS
python3 tools/pmc_summary.py "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
