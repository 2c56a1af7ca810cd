#!/bin/bash
# Round 5, GPU pass s: the large-tree policy evaluation with every player's best response in one set of launches.
set -u
OUT=gpurun_out/${1:-r05s}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== pytest (evaluation, CFR-BR, reference vectors)" | tee "$OUT/summary.txt"
timeout 1500 python -m pytest tests/test_gpu_cfr.py tests/test_z1_gpu_reference_vectors.py tests/test_z4_gpu_reference_vectors_r2.py tests/test_z6_gpu_reference_tests_on_mirror.py -q -m gpu -k "eval or nash or best_response or exploit or cfr_br or judge or three_player or reference_vectors or tabular or best or large_tree" --durations=5 > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -8 "$OUT/pytest.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2; do timeout 300 python tools/probe_judge.py 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-200 | tee -a "$OUT/summary.txt"; done
du -sh "$OUT"
