#!/bin/bash
# Round 6, GPU pass 21c: the one-root search — readlane for the chosen child, terminal-after-move, select9, hoisted rng bases (default) against the previous build (variant prev)
# (variant prev): tree parity (test_z8, test_z5, mirror tests), then the config-1 search rate, alternating.
set -u
OUT=gpurun_out/${1:-r06zy}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_z8_gpu_single_root_search.py tests/test_z5_gpu_mcts_evaluator.py tests/test_z6_gpu_reference_tests_on_mirror.py -q -m gpu -x > "$OUT/pytest_step.log" 2>&1
echo "pytest exit $?" | tee "$OUT/summary.txt"; tail -3 "$OUT/pytest_step.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2 3; do
  for v in prev default; do
    echo "-- $v (rep $rep)" | tee -a "$OUT/summary.txt"
    if [ $v = default ]; then timeout 300 python tools/probe_single_root.py 2>&1 | tail -4 | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_single_root.py 2>&1 | tail -4 | tee -a "$OUT/summary.txt"; fi
  done
done
for v in prev default; do
  echo "-- $v connect_four" | tee -a "$OUT/summary.txt"
  if [ $v = default ]; then PROBE_GAME=connect_four timeout 300 python tools/probe_single_root.py 2>&1 | tail -3 | tee -a "$OUT/summary.txt"
  else PROBE_GAME=connect_four OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_single_root.py 2>&1 | tail -3 | tee -a "$OUT/summary.txt"; fi
done
