#!/bin/bash
# Round 6, GPU pass 11: k_cfr_sub, the two-member form per wavefront: parity, rate, barrier arrivals.
set -u
OUT=gpurun_out/${1:-r06m}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_cfr.py tests/test_gpu_timed_batch.py -q -m gpu -k "subtree or three_player or sub_ or persistent or grid_barrier or cfr_br or variants" > "$OUT/pytest_sub.log" 2>&1
echo "pytest exit $?" | tee "$OUT/summary.txt"; tail -3 "$OUT/pytest_sub.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2; do timeout 300 python tools/probe_cfr_sub.py 2>&1 | grep -E "^sub|^auto|kuhn_poker\(players=[56]\) \[sub|^leduc_poker \[sub" | cut -c1-200 | tee -a "$OUT/summary.txt"; done
OSG_CFR_SUB_STAMPS=1 timeout 300 python tools/probe_cfr_sub_once.py 2>&1 | grep "arrivals\|pass 1 (" | cut -c1-330 | tee -a "$OUT/summary.txt"
OSG_CFR_SUB_STAMPS=101 timeout 300 python tools/probe_cfr_sub_once.py 2>&1 | grep "pass 1 (" | cut -c1-330 | tee -a "$OUT/summary.txt"
