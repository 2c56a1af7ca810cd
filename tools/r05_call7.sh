#!/bin/bash
# Round 5, GPU pass j: k_cfr_split with 16-byte term pieces and a release word, A/B against the previous build
# (tools/variants/libosg_prevsplit.so = osg_cfr.hip of the commit before), all CFR tests.
set -u
OUT=gpurun_out/${1:-r05j}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== pytest tests/test_gpu_cfr.py + reference vectors" | tee "$OUT/summary.txt"
timeout 1500 python -m pytest tests/test_gpu_cfr.py tests/test_z1_gpu_reference_vectors.py tests/test_z4_gpu_reference_vectors_r2.py -q -m gpu --durations=5 -x > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -10 "$OUT/pytest.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  for v in prevsplit now; do
    if [ $v = now ]; then unset OSG_VARIANT_LIB; else export OSG_VARIANT_LIB=tools/variants/libosg_$v.so; fi
    timeout 300 python tools/probe_cfr.py > "$OUT/probe_cfr_${v}_$rep.log" 2>&1
    echo "-- $v $rep"; grep -E "split|players=3|^leduc_poker CFR iters" "$OUT/probe_cfr_${v}_$rep.log" | cut -c1-150
  done
done 2>&1 | tee -a "$OUT/summary.txt"
unset OSG_VARIANT_LIB
timeout 300 python tools/probe_judge.py 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-200 | tee -a "$OUT/summary.txt"
du -sh "$OUT"
