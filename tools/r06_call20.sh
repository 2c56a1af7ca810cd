#!/bin/bash
# Round 6, GPU pass 20: the bench line with the refreshed counter profile of the search kernel; packed-flag floods for the
# boards of up to 128 cells (variants packed2 at 7, packed2w6 at 6 wavefronts per SIMD) against the default.
set -u
OUT=gpurun_out/${1:-r06zw}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python bench.py > "$OUT/bench_n1.log" 2> "$OUT/bench_n1.err"
echo "bench exit $?" | tee "$OUT/summary.txt"
tail -n 1 "$OUT/bench_n1.log" | wc -c | tee -a "$OUT/summary.txt"
tail -n 1 "$OUT/bench_n1.log" | tee -a "$OUT/summary.txt"
for v in packed2 packed2w6; do
  OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 900 python -m pytest tests/test_gpu_mcts.py -q -m gpu -x -k "replay_parity and hex" > "$OUT/pytest_$v.log" 2>&1
  echo "pytest $v exit $?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/pytest_$v.log" | cut -c1-200 | tee -a "$OUT/summary.txt"
done
for rep in 1 2 3; do
  for v in default packed2 packed2w6; do
    echo "-- $v (rep $rep)" | tee -a "$OUT/summary.txt"
    if [ $v = default ]; then timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"; fi
  done
done
for v in default packed2 packed2w6; do
  echo "-- $v" | tee -a "$OUT/summary.txt"
  if [ $v = default ]; then timeout 300 python tools/probe_mcts.py hex 65536 512 0 2 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"
  else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_mcts.py hex 65536 512 0 2 2>&1 | grep "sims/s" | tee -a "$OUT/summary.txt"; fi
done
