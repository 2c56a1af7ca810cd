#!/bin/bash
# Round 5, second GPU pass: kuhn kernel diet, ES-MCCFR flat kernel variants, hex step without the mask row.
set -u
OUT=gpurun_out/r05b
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== pytest (touched)" | tee "$OUT/summary.txt"
timeout 1200 python -m pytest tests/test_gpu_timed_batch.py tests/test_gpu_cfr.py tests/test_gpu_parity.py tests/test_z1_gpu_reference_vectors.py tests/test_z4_gpu_reference_vectors_r2.py -q -m gpu --durations=8 > "$OUT/pytest_b.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -14 "$OUT/pytest_b.log" | tee -a "$OUT/summary.txt"
echo "== A/B solvers: r4regs (round-4 kernels) / peekonly / frames2only / now" | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  for v in r4regs peekonly frames2only now; do
    if [ $v = now ]; then unset OSG_VARIANT_LIB; else export OSG_VARIANT_LIB=tools/variants/libosg_$v.so; fi
    timeout 300 python tools/probe_cfr.py > "$OUT/probe_cfr_${v}_$rep.log" 2>&1
    echo "-- $v $rep"; grep -E "^kuhn_poker CFR|split \(auto\)|MCCFR" "$OUT/probe_cfr_${v}_$rep.log" | cut -c1-150
  done
done 2>&1 | tee -a "$OUT/summary.txt"
unset OSG_VARIANT_LIB
echo "== hex step" | tee -a "$OUT/summary.txt"
timeout 600 python tools/probe_hex_step.py > "$OUT/hex_step.log" 2>&1; grep -v amdgpu.ids "$OUT/hex_step.log" | tee -a "$OUT/summary.txt"
echo "== counters of the solver kernels (now)" | tee -a "$OUT/summary.txt"
bash tools/pmc_solvers.sh r05b 2>&1 | tee -a "$OUT/summary.txt"
du -sh "$OUT"
