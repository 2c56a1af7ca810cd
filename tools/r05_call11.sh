#!/bin/bash
# Round 5, GPU pass q: the flat ES-MCCFR kernel as ONE loop across a lane's trajectories, A/B against the previous build
# (tools/variants/libosg_prevmccfr.so), MCCFR tests incl. the 2^20-trajectory parity at size.
set -u
OUT=gpurun_out/${1:-r05q}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== pytest (mccfr)" | tee "$OUT/summary.txt"
timeout 1500 python -m pytest tests/test_gpu_cfr.py tests/test_gpu_timed_batch.py tests/test_z4_gpu_reference_vectors_r2.py tests/test_z7_gpu_exchange_steps.py -q -m gpu -k "mccfr or config5 or MCCFR or sharded or minibatch" --durations=5 > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -8 "$OUT/pytest.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  for v in prevmccfr now; do
    if [ $v = now ]; then unset OSG_VARIANT_LIB; else export OSG_VARIANT_LIB=tools/variants/libosg_$v.so; fi
    echo "-- $v $rep" | tee -a "$OUT/summary.txt"
    timeout 300 python tools/probe_cfr.py 2>&1 | grep -E "MCCFR" | cut -c1-150 | tee -a "$OUT/summary.txt"
    timeout 300 python tools/probe_mccfr_bench16.py 2>&1 | grep -E "mini-batches|per launch" | cut -c1-200 | tee -a "$OUT/summary.txt"
  done
done
unset OSG_VARIANT_LIB
du -sh "$OUT"
