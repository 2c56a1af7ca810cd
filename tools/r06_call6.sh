#!/bin/bash
# Round 6, GPU pass 6: ES-MCCFR flat kernel, tree in L2 with / without the top of the tree staged in the LDS that two
# workgroups per CU leave free; config 5 parity in the new default; the reference's mcts_example at seed 11 (the exact
# output for the test); wide-board rows of probe_kernels.
set -u
OUT=gpurun_out/${1:-r06g}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
: > "$OUT/summary.txt"
for rep in 1 2; do
  for top in 0 1; do
    echo "-- tree in L2, top of the tree in LDS: $top (rep $rep)" | tee -a "$OUT/summary.txt"
    OSG_MCCFR_TREE_LDS_TOP=$top timeout 300 python tools/probe_mccfr_bench16.py 2>&1 | tail -2 | tee -a "$OUT/summary.txt"
  done
done
echo "== config 5 parity (default)" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_timed_batch.py tests/test_gpu_cfr.py -q -m gpu -k "config5 or mccfr" > "$OUT/pytest_mccfr.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_mccfr.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
echo "== mcts_example --seed=11" | tee -a "$OUT/summary.txt"
timeout 300 tests/_refbuilt/reference_example_mcts_example --num_games=2 --max_simulations=1000 --quiet=true --seed=11 2>&1 | tail -12 | tee -a "$OUT/summary.txt"
echo "== wide boards" | tee -a "$OUT/summary.txt"
timeout 900 python tools/probe_kernels.py > "$OUT/probe_kernels.log" 2>&1
grep "wide" "$OUT/probe_kernels.log" | cut -c1-260 | tee -a "$OUT/summary.txt"
