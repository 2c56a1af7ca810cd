#!/bin/bash
# Round 6, GPU pass 1: the bench exactly as the driver runs it (the last stdout line must be the < 4 KB record), then the
# whole -m gpu suite.
set -u
OUT=gpurun_out/${1:-r06a}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export OSG_BENCH_DETAIL_DIR="$PWD/$OUT"
( time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench_n1.log" 2> "$OUT/bench_n1.err"
echo "bench rc $? last line $(tail -1 "$OUT/bench_n1.log" | wc -c) chars" | tee "$OUT/summary.txt"
tail -1 "$OUT/bench_n1.log" | tee -a "$OUT/summary.txt"
unset OSG_BENCH_DETAIL_DIR
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -25 "$OUT/pytest_gpu.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
