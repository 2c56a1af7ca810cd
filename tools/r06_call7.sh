#!/bin/bash
# Round 6, GPU pass 7: where k_cfr_sub's members phase spends its time — timing variants (results wrong by construction):
# 1 = no term stores, 2 = no LDS gathers, 4 = no static record fetch, 7 = none of the three.
set -u
OUT=gpurun_out/${1:-r06h}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
: > "$OUT/summary.txt"
for v in default subexp1 subexp2 subexp4 subexp7; do
  for wg in 101 201; do
    if [ $v = default ]; then OSG_CFR_SUB_STAMPS=$wg timeout 300 python tools/probe_cfr_sub_once.py 2>&1 | grep "pass 1" | sed "s/^/$v wg $((wg-1)): /" | cut -c1-220 | tee -a "$OUT/summary.txt"
    else OSG_VARIANT_LIB=tools/variants/libosg_$v.so OSG_CFR_SUB_STAMPS=$wg timeout 300 python tools/probe_cfr_sub_once.py 2>&1 | grep "pass 1" | sed "s/^/$v wg $((wg-1)): /" | cut -c1-220 | tee -a "$OUT/summary.txt"; fi
  done
done
for v in default subexp7; do
  echo "-- rate $v" | tee -a "$OUT/summary.txt"
  if [ $v = default ]; then timeout 300 python tools/probe_cfr_sub.py 2>&1 | grep -E "^sub:" | tee -a "$OUT/summary.txt"
  else OSG_VARIANT_LIB=tools/variants/libosg_$v.so timeout 300 python tools/probe_cfr_sub.py 2>&1 | grep -E "^sub:" | tee -a "$OUT/summary.txt"; fi
done
