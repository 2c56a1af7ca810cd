#!/bin/bash
# Round 6, GPU pass 8: k_cfr_sub with a member's 64-byte record written by its quad in one contiguous piece: parity, rate, stamps.
set -u
OUT=gpurun_out/${1:-r06i}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_cfr.py tests/test_gpu_timed_batch.py -q -m gpu -k "subtree or three_player or sub_ or persistent or grid_barrier or cfr_br or variants" > "$OUT/pytest_sub.log" 2>&1
echo "pytest exit $?" | tee "$OUT/summary.txt"; tail -3 "$OUT/pytest_sub.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2; do timeout 300 python tools/probe_cfr_sub.py 2>&1 | grep -E "^sub|^auto|kuhn_poker\(players=[56]\)|^leduc" | cut -c1-200 | tee -a "$OUT/summary.txt"; done
for wg in 1 101 201; do
  OSG_CFR_SUB_STAMPS=$wg timeout 300 python tools/probe_cfr_sub_once.py 2>&1 | grep "pass 1" | sed "s/^/wg $((wg-1)): /" | cut -c1-220 | tee -a "$OUT/summary.txt"
done
