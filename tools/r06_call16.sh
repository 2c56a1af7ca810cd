#!/bin/bash
# Round 6, GPU pass 16: packed-flag floods on the boards above 128 cells (six cross-lane reads per step whatever kS is):
# parity, phase shares, rates at 5/4/3, 6/5/4 (default) and 7/6/5 wavefronts per SIMD, hex(9) against the previous object.
set -u
OUT=gpurun_out/${1:-r06zq}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_mcts.py tests/test_gpu_parity.py -q -m gpu -x -k "mcts or hex_above or wave" > "$OUT/pytest_mcts.log" 2>&1
echo "pytest exit $?" | tee "$OUT/summary.txt"; tail -5 "$OUT/pytest_mcts.log" | cut -c1-400 | tee -a "$OUT/summary.txt"
for spec in "16384 hex(board_size=13) 256 30" "16384 hex(board_size=16) 256 30" "16384 hex(board_size=19) 128 30"; do
  set -- $spec
  OSG_VARIANT_LIB=tools/variants/libosg_pt.so timeout 300 python tools/probe_mcts_phases.py $1 "$2" $3 $4 2>&1 | tail -9 | tee -a "$OUT/summary.txt"
done
for rep in 1 2; do
for v in wpe543 default wpe765; do
  echo "-- $v (rep $rep)" | tee -a "$OUT/summary.txt"
  for spec in "hex(board_size=12) 65536 256" "hex(board_size=13) 65536 256" "hex(board_size=16) 32768 256" "hex(board_size=19) 65536 128" "hex(board_size=19) 8192 512"; do
    set -- $spec
    if [ $v = default ]; then timeout 300 python tools/probe_mcts.py "$1" $2 $3 0 2 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_mcts.py "$1" $2 $3 0 2 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"; fi
  done
done
done
for rep in 1 2; do
  for v in waveold default; do
    echo "-- $v (rep $rep)" | tee -a "$OUT/summary.txt"
    if [ $v = default ]; then timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"; fi
  done
done
