#!/bin/bash
# Round 6, GPU pass 19: hex fill search — the first visit's IsTerminal() after the playouts (default) against before them
# (variant nodefer): search parity, then rates, alternating.
set -u
OUT=gpurun_out/${1:-r06zu}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_mcts.py tests/test_gpu_fullsize.py tests/test_gpu_timed_batch.py -q -m gpu -x -k "mcts or wave or hex or search or roots" > "$OUT/pytest_search.log" 2>&1
echo "pytest exit $?" | tee "$OUT/summary.txt"; tail -3 "$OUT/pytest_search.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2 3; do
  for v in nodefer default; do
    echo "-- $v (rep $rep)" | tee -a "$OUT/summary.txt"
    if [ $v = default ]; then timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"; fi
  done
done
for v in nodefer default; do
  echo "-- $v" | tee -a "$OUT/summary.txt"
  for spec in "hex 65536 512" "hex(board_size=13) 65536 256" "hex(board_size=16) 32768 256" "hex(board_size=19) 8192 512"; do
    set -- $spec
    if [ $v = default ]; then timeout 300 python tools/probe_mcts.py "$1" $2 $3 0 2 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_mcts.py "$1" $2 $3 0 2 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"; fi
  done
done
