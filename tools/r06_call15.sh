#!/bin/bash
# Round 6, GPU pass 15: the wave-per-root search on the boards above 128 cells: phase shares (variant pt) and the
# wavefronts-per-SIMD choice (variants wpelo / wpehi against the default 5 / 4 / 3 for 3 / 4 / 6 cell sets).
set -u
OUT=gpurun_out/${1:-r06zp}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for spec in "16384 hex(board_size=13) 256 30" "16384 hex(board_size=16) 256 30" "16384 hex(board_size=19) 128 30" "8192 hex(board_size=19) 512 30"; do
  set -- $spec
  OSG_VARIANT_LIB=tools/variants/libosg_pt.so timeout 300 python tools/probe_mcts_phases.py $1 "$2" $3 $4 2>&1 | tail -9 | tee -a "$OUT/summary.txt"
done
for rep in 1 2; do
for v in wpelo default wpehi; do
  echo "-- $v (rep $rep)" | tee -a "$OUT/summary.txt"
  for spec in "hex(board_size=13) 65536 256" "hex(board_size=16) 32768 256" "hex(board_size=19) 65536 128" "hex(board_size=19) 8192 512"; do
    set -- $spec
    if [ $v = default ]; then timeout 300 python tools/probe_mcts.py "$1" $2 $3 0 2 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_mcts.py "$1" $2 $3 0 2 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"; fi
  done
done
done
