#!/bin/bash
# Counter passes over tools/probe_solvers_once.py (the kernels of configs 3 and 5), one rocprofv3 --pmc run per counter
# group (only beside --kernel-trace), summarised into profiles/<tag>_pmc_solvers.json by tools/pmc_solvers.py.
#   bash tools/pmc_solvers.sh r05
set -u
TAG=$1
OUT=gpurun_out/$TAG/pmc_solvers
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
GROUPS_=(
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY"
)
P=0
for G in "${GROUPS_[@]}"; do
  P=$((P+1))
  timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$OUT/p$P" -- python tools/probe_solvers_once.py > "$OUT/p$P.log" 2>&1
  echo "pmc solvers pass $P ($G) exit $?"
done
python tools/pmc_solvers.py "$OUT" "$TAG"
find "$OUT" -name '*.db' -delete 2>/dev/null
