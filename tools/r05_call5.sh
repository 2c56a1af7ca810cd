#!/bin/bash
# Round 5, GPU pass f: the forest form of k_cfr_sub (bins packed so that every workgroup sweeps one bin per pass) against
# a subtree per bin (OSG_CFR_SUB_PACK=0), tables bit for bit and iterations/s, with phase stamps of three workgroups.
set -u
OUT=gpurun_out/${1:-r05f}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== pytest (touched)" | tee "$OUT/summary.txt"
timeout 1200 python -m pytest tests/test_gpu_cfr.py "tests/test_z6_gpu_reference_tests_on_mirror.py::test_reference_example_program_runs_on_the_mirror" -q -m gpu --durations=8 -x > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -25 "$OUT/pytest.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  for pack in 0 1; do
    echo "-- OSG_CFR_SUB_PACK=$pack rep $rep" | tee -a "$OUT/summary.txt"
    OSG_CFR_SUB_PACK=$pack timeout 300 python tools/probe_cfr_sub.py > "$OUT/probe_cfr_sub_pack${pack}_$rep.log" 2>&1
    grep -E "identical|^grid|^sub|^auto" "$OUT/probe_cfr_sub_pack${pack}_$rep.log" | tee -a "$OUT/summary.txt"
  done
done
for wg in 1 101 251; do
  echo "-- stamps of workgroup $((wg-1)), pack 1" | tee -a "$OUT/summary.txt"
  OSG_CFR_SUB_STAMPS=$wg PROBE_ONLY_SUB=1 timeout 300 python tools/probe_cfr_sub.py 2>&1 | grep -E "k_cfr_sub" | tail -4 | tee -a "$OUT/summary.txt"
done
du -sh "$OUT"
