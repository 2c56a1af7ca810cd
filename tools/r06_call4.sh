#!/bin/bash
# Round 6, GPU pass 4: (a) config 5 parity + counters with the tree-in-L2 default; (b) one-root search: volatile node
# accesses (default) against plain accesses + a compiler barrier after lane 0's store (tools/variants/libosg_noderef1.so).
set -u
OUT=gpurun_out/${1:-r06d}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
: > "$OUT/summary.txt"
echo "== config 5 parity, default form" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_timed_batch.py tests/test_gpu_cfr.py -q -m gpu -k "config5 or mccfr" > "$OUT/pytest_mccfr.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_mccfr.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
echo "== one-root search A/B" | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  echo "-- volatile (default) rep $rep" | tee -a "$OUT/summary.txt"
  timeout 200 python tools/probe_single_root.py 2>&1 | tail -3 | tee -a "$OUT/summary.txt"
  echo "-- plain + barrier (noderef1) rep $rep" | tee -a "$OUT/summary.txt"
  OSG_VARIANT_LIB=tools/variants/libosg_noderef1.so timeout 200 python tools/probe_single_root.py 2>&1 | tail -3 | tee -a "$OUT/summary.txt"
done
echo "== parity of the variant (C-ABI tree tests)" | tee -a "$OUT/summary.txt"
OSG_VARIANT_LIB=tools/variants/libosg_noderef1.so timeout 900 python -m pytest tests/test_z5_gpu_mcts_evaluator.py tests/test_gpu_mcts.py tests/test_z8_gpu_single_root_search.py -q -m gpu -x > "$OUT/pytest_noderef1.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_noderef1.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
echo "== counters of the solver kernels" | tee -a "$OUT/summary.txt"
bash tools/pmc_solvers.sh "${1:-r06d}" 2>&1 | tail -8 | cut -c1-600 | tee -a "$OUT/summary.txt"
cp profiles/${1:-r06d}_pmc_solvers.json "$OUT/" 2>/dev/null
