#!/bin/bash
# Round 5, GPU pass y: the hex playout's flood with both exits on the vector unit, tested once per two steps
# (OSG_FLOOD_MODE 2) against mode 1 (tools/variants/libosg_flood1.so): replay parity, then the search rate A/B.
set -u
OUT=gpurun_out/${1:-r05y}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== pytest (searches)" | tee "$OUT/summary.txt"
timeout 1200 python -m pytest tests -q -m gpu -k "mcts or search or wave or hex" --durations=5 > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -4 "$OUT/pytest.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  for v in flood1 default; do
    echo "-- $v $rep" | tee -a "$OUT/summary.txt"
    if [ $v = default ]; then timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep hex | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep hex | tee -a "$OUT/summary.txt"; fi
  done
done
