#!/bin/bash
# Round 6, GPU pass 5: how even are k_cfr_sub's bins on 3-player leduc (histories / members per player), and the phase
# stamps of several workgroups.
set -u
OUT=gpurun_out/${1:-r06e}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OSG_CFR_SUB_STATS=1 timeout 300 python tools/probe_cfr_sub_once.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/summary.txt"
for wg in 1 41 101 161 201 251; do
  OSG_CFR_SUB_STAMPS=$wg timeout 300 python tools/probe_cfr_sub_once.py 2>&1 | grep "pass" | sed "s/^/wg $((wg-1)): /" | tee -a "$OUT/summary.txt"
done
