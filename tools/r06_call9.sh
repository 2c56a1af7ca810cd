#!/bin/bash
# Round 6, GPU pass 9: k_cfr_sub member-record prefetch before the sweep: 0 (none) / 1 (first member, default) / 2 (both).
set -u
OUT=gpurun_out/${1:-r06j}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_cfr.py tests/test_gpu_timed_batch.py -q -m gpu -k "subtree or three_player or sub_ or persistent or grid_barrier or cfr_br or variants" > "$OUT/pytest_sub.log" 2>&1
echo "pytest exit $?" | tee "$OUT/summary.txt"; tail -3 "$OUT/pytest_sub.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  for v in subpf0 default subpf2; do
    echo "-- prefetch $v (rep $rep)" | tee -a "$OUT/summary.txt"
    if [ $v = default ]; then timeout 300 python tools/probe_cfr_sub.py 2>&1 | grep -E "^sub:|kuhn_poker\(players=[56]\) \[sub|^leduc_poker \[sub" | cut -c1-200 | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_cfr_sub.py 2>&1 | grep -E "^sub:|kuhn_poker\(players=[56]\) \[sub|^leduc_poker \[sub" | cut -c1-200 | tee -a "$OUT/summary.txt"; fi
  done
done
for wg in 1 101; do
  OSG_CFR_SUB_STAMPS=$wg timeout 300 python tools/probe_cfr_sub_once.py 2>&1 | grep "pass 1" | sed "s/^/default wg $((wg-1)): /" | cut -c1-220 | tee -a "$OUT/summary.txt"
done
