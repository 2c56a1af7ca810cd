#!/bin/bash
# Round 5, GPU pass k: the observer types, the struct API, the reference's tic_tac_toe_test.cc / connect_four_test.cc on the mirror.
set -u
OUT=gpurun_out/${1:-r05k}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== pytest (touched)" | tee "$OUT/summary.txt"
timeout 1500 python -m pytest tests/test_observer_types.py tests/test_z12_gpu_struct_api.py tests/test_pyspiel_surface.py tests/test_z6_gpu_reference_tests_on_mirror.py tests/test_abi.py -q -m gpu --durations=8 > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -40 "$OUT/pytest.log" | cut -c1-400 | tee -a "$OUT/summary.txt"
du -sh "$OUT"
