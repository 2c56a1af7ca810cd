#!/bin/bash
# Round 5, GPU pass l: hex(9) on the 12-word record (the meta word folded into the planes' spare bits): the whole GPU suite,
# then the step / tensor / search rates with OSG_HEX_FOLD=0 (13 words) and 1 on the same box.
set -u
OUT=gpurun_out/${1:-r05l}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== pytest -m gpu (everything)" | tee "$OUT/summary.txt"
timeout 1500 python -m pytest tests -q -m gpu -x --durations=5 > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -12 "$OUT/pytest.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  for fold in 0 1; do
    echo "-- OSG_HEX_FOLD=$fold rep $rep" | tee -a "$OUT/summary.txt"
    OSG_HEX_FOLD=$fold timeout 300 python tools/probe_hex_step.py 2>&1 | grep -E "default" | tee -a "$OUT/summary.txt"
    OSG_HEX_FOLD=$fold timeout 300 python tools/probe_kernels.py 2>&1 | grep -E "hex" | grep "2\^24" | cut -c1-200 | tee -a "$OUT/summary.txt"
    OSG_HEX_FOLD=$fold timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep hex | tee -a "$OUT/summary.txt"
  done
done
du -sh "$OUT"
