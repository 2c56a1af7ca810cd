#!/bin/bash
# Round 5, GPU pass x: the folded hex record on every plane width (cells <= 32 NW - 5), the form read from the Params
# (HexT::folded) except in the byte-bound step kernel: device sweep, the hex tests, the step / search rates by form.
set -u
OUT=gpurun_out/${1:-r05x}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== device sweep (59 configurations x 20 000 playouts against the oracle)" | tee "$OUT/summary.txt"
timeout 600 python tools/device_sweep.py 20000 > "$OUT/device_sweep.log" 2>&1
echo "sweep exit $?" | tee -a "$OUT/summary.txt"
grep -E "MISMATCH|hex" "$OUT/device_sweep.log" | tail -30 | cut -c1-200 | tee -a "$OUT/summary.txt"
echo "== pytest -m gpu -k hex + parity + serial + struct" | tee -a "$OUT/summary.txt"
timeout 1200 python -m pytest tests -q -m gpu -k "hex or parity or serial or gather or observation or observer or struct or checkpoint" --durations=5 > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -8 "$OUT/pytest.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for fold in 0 1; do
  echo "-- OSG_HEX_FOLD=$fold" | tee -a "$OUT/summary.txt"
  OSG_HEX_FOLD=$fold timeout 300 python tools/probe_hex_step.py 2>&1 | grep -E "default" | tee -a "$OUT/summary.txt"
  OSG_HEX_FOLD=$fold timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep hex | tee -a "$OUT/summary.txt"
  OSG_HEX_FOLD=$fold timeout 300 python tools/probe_mcts_bench.py 2>&1 | grep hex | tee -a "$OUT/summary.txt"
done
echo "== other boards: step rate by form" | tee -a "$OUT/summary.txt"
for fold in 0 1; do OSG_HEX_FOLD=$fold timeout 300 python tools/probe_hex_boards.py 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/summary.txt"; done
du -sh "$OUT"
