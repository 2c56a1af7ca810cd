#!/bin/bash
# Round 6, GPU pass 13: k_cfr_sub descriptors fetched once per launch (default here) vs once per pass (variant subkeep0).
set -u
OUT=gpurun_out/${1:-r06p}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_cfr.py tests/test_gpu_timed_batch.py -q -m gpu -k "subtree or three_player or sub_ or persistent or grid_barrier or cfr_br or variants" > "$OUT/pytest_sub.log" 2>&1
echo "pytest exit $?" | tee "$OUT/summary.txt"; tail -3 "$OUT/pytest_sub.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2 3; do
  for v in subkeep0 default; do
    echo "-- $v (rep $rep)" | tee -a "$OUT/summary.txt"
    if [ $v = default ]; then timeout 300 python tools/probe_cfr_sub.py 2>&1 | grep -E "^sub:|kuhn_poker\(players=[56]\) \[sub|^leduc_poker \[sub" | cut -c1-200 | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_cfr_sub.py 2>&1 | grep -E "^sub:|kuhn_poker\(players=[56]\) \[sub|^leduc_poker \[sub" | cut -c1-200 | tee -a "$OUT/summary.txt"; fi
  done
done
