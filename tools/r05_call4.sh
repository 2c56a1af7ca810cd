#!/bin/bash
set -u
OUT=gpurun_out/r05d
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cfr.py tests/test_z1_gpu_reference_vectors.py tests/test_z4_gpu_reference_vectors_r2.py -q -m gpu > "$OUT/pytest_d.log" 2>&1
echo "pytest exit $?" | tee "$OUT/summary.txt"; tail -3 "$OUT/pytest_d.log" | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  OSG_CFR_SPLIT_W0=1 timeout 300 python tools/probe_cfr.py > "$OUT/probe_cfr_w0_$rep.log" 2>&1
  timeout 300 python tools/probe_cfr.py > "$OUT/probe_cfr_w3_$rep.log" 2>&1
  echo "-- loop form $rep"; grep -E "split" "$OUT/probe_cfr_w0_$rep.log" | cut -c1-150
  echo "-- row width 3 $rep"; grep -E "split" "$OUT/probe_cfr_w3_$rep.log" | cut -c1-150
done 2>&1 | tee -a "$OUT/summary.txt"
