#!/bin/bash
# Round 6, GPU pass 17: hex fill kernel — expansion from lane masks (expsets, now the default) against the 4-word mask
# (expmask), and the threshold search on the keys' 32 mixed bits first (thr3); parity of thr3, then rates, alternating.
set -u
OUT=gpurun_out/${1:-r06zr}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OSG_VARIANT_LIB=tools/variants/libosg_thr3.so timeout 1500 python -m pytest tests/test_gpu_mcts.py tests/test_gpu_fullsize.py -q -m gpu -x -k "mcts or wave or hex" > "$OUT/pytest_thr3.log" 2>&1
echo "pytest thr3 exit $?" | tee "$OUT/summary.txt"; tail -3 "$OUT/pytest_thr3.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
timeout 1500 python -m pytest tests/test_gpu_mcts.py -q -m gpu -x > "$OUT/pytest_default.log" 2>&1
echo "pytest default exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_default.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2 3; do
  for v in expmask default thr3; do
    echo "-- $v (rep $rep)" | tee -a "$OUT/summary.txt"
    if [ $v = default ]; then timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"; fi
  done
done
for v in default thr3; do
  echo "-- $v" | tee -a "$OUT/summary.txt"
  for spec in "hex 65536 512" "hex(board_size=13) 65536 256" "hex(board_size=16) 32768 256" "hex(board_size=19) 65536 128"; do
    set -- $spec
    if [ $v = default ]; then timeout 300 python tools/probe_mcts.py "$1" $2 $3 0 2 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_mcts.py "$1" $2 $3 0 2 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"; fi
  done
done
