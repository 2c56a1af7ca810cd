#!/bin/bash
# Round 6, GPU pass 10: k_cfr_sub — when every workgroup reaches the two barriers of a pass (who do the others wait for?).
set -u
OUT=gpurun_out/${1:-r06l}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for rep in 1 2 3; do OSG_CFR_SUB_STAMPS=1 timeout 300 python tools/probe_cfr_sub_once.py 2>&1 | grep "arrivals\|pass 1 (" | cut -c1-330 | tee -a "$OUT/summary.txt"; done
