#!/bin/bash
# Round 6, GPU pass 2: the new parity tests at size, then the bench with the 3-player-leduc parity leg.
set -u
OUT=gpurun_out/${1:-r06b}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_timed_batch.py tests/test_z12_gpu_struct_api.py -q -m gpu -s --durations=8 \
  -k "three_player or piece_form_poker or starting_position" > "$OUT/pytest_new.log" 2>&1
echo "pytest exit $?" | tee "$OUT/summary.txt"
grep -E "against the|passed|failed|Error|error" "$OUT/pytest_new.log" | cut -c1-400 | tail -30 | tee -a "$OUT/summary.txt"
export OSG_BENCH_DETAIL_DIR="$PWD/$OUT"
( time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench_n1.log" 2> "$OUT/bench_n1.err"
echo "bench rc $? last line $(tail -1 "$OUT/bench_n1.log" | wc -c) chars" | tee -a "$OUT/summary.txt"
tail -1 "$OUT/bench_n1.log" | tee -a "$OUT/summary.txt"
tail -4 "$OUT/bench_n1.err" | tee -a "$OUT/summary.txt"
