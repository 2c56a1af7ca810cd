#!/bin/bash
# Round 6, GPU pass 14: the wave-per-root search on hex boards above 128 cells (kS = 3 / 4 / 6 cell sets): replay parity,
# rates against the lane-per-root layout, and hex(9) / 11 x 11 against the previous wave kernel object (variant waveold).
set -u
OUT=gpurun_out/${1:-r06zn}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_mcts.py tests/test_gpu_parity.py -q -m gpu -x -k "mcts or hex_above or wave" > "$OUT/pytest_mcts.log" 2>&1
echo "pytest exit $?" | tee "$OUT/summary.txt"; tail -15 "$OUT/pytest_mcts.log" | cut -c1-400 | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  for v in waveold default; do
    echo "-- $v (rep $rep)" | tee -a "$OUT/summary.txt"
    if [ $v = default ]; then timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"; fi
  done
done
for spec in "hex 65536 512" "hex(board_size=12) 65536 256" "hex(board_size=13) 65536 256" "hex(board_size=15) 65536 256" "hex(board_size=16) 32768 256" "hex(board_size=19) 65536 128" "hex(board_size=19) 8192 512"; do
  set -- $spec
  for layout in 1 2; do
    timeout 600 python tools/probe_mcts.py "$1" $2 $3 0 $layout 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"
  done
done
