#!/bin/bash
# Counter passes over a small python workload, one rocprofv3 --pmc run per counter group (never combined with
# other trace domains than --kernel-trace), summarised per kernel by tools/pmc_kernels_summary.py.
#   bash tools/pmc_kernels.sh <tag> <kernel-substring> <python file> [env assignments...]
set -u
TAG=$1; KERNEL=$2; PY=$3; shift 3
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
GROUPS_=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
 "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_WRITE_sum TCC_READ_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_WRITEBACK_sum TCC_TAG_STALL_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum"
 "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TCC_BUSY_avr"
 "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM"
 "GRBM_GUI_ACTIVE TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum"
)
P=0
for G in "${GROUPS_[@]}"; do
  P=$((P+1))
  env "$@" timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$OUT/p$P" -- python "$PY" > "$OUT/p$P.log" 2>&1
  echo "pass $P ($G) exit $?"
done
python tools/pmc_kernels_summary.py "$OUT" "$KERNEL" | tee "$OUT/summary.txt"
find "$OUT" -name '*.db' -delete 2>/dev/null
