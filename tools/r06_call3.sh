#!/bin/bash
# Round 6, GPU pass 3: ES-MCCFR flat kernel with the tree in LDS (one 1024-lane workgroup per CU, 4 waves/SIMD) against
# the tree read from global memory / L2 (two workgroups per CU, 8 waves/SIMD): rate A/B at the bench's 16 x 2^20
# schedule, parity of a whole mini-batch in the new form, and the exchange A/B test added this round.
set -u
OUT=gpurun_out/${1:-r06c}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
: > "$OUT/summary.txt"
for rep in 1 2; do
  for where in lds global; do
    echo "-- tree in $where (rep $rep)" | tee -a "$OUT/summary.txt"
    OSG_MCCFR_TREE=$where timeout 300 python tools/probe_mccfr_bench16.py 2>&1 | tail -2 | tee -a "$OUT/summary.txt"
  done
done
echo "== parity of a whole mini-batch, tree in global" | tee -a "$OUT/summary.txt"
OSG_MCCFR_TREE=global timeout 900 python -m pytest tests/test_gpu_timed_batch.py tests/test_gpu_cfr.py -q -m gpu -k "config5 or mccfr" > "$OUT/pytest_mccfr_global.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_mccfr_global.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
echo "== exchange tests" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_z10_gpu_oneshot_allreduce.py tests/test_z7_gpu_exchange_steps.py -q -m gpu > "$OUT/pytest_exchange.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_exchange.log" | cut -c1-400 | tee -a "$OUT/summary.txt"
