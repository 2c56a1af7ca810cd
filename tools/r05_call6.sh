#!/bin/bash
# Round 5, GPU pass g: k_cfr_sub with 64-byte member records (16-byte written-through stores / bypassing loads), two
# members per thread and round, and the two-level grid barrier — A/B of the barrier (OSG_CFR_SUB_FLAT_BARRIER=1: round 4's).
set -u
OUT=gpurun_out/${1:-r05g}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== pytest (touched)" | tee "$OUT/summary.txt"
timeout 1200 python -m pytest tests/test_gpu_cfr.py -q -m gpu --durations=5 -x -k "sub or persistent or three_player or barrier" > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -12 "$OUT/pytest.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  for flat in 0; do
    echo "-- rep $rep" | tee -a "$OUT/summary.txt"
    PROBE_ONLY_SUB=1 timeout 300 python tools/probe_cfr_sub.py > "$OUT/probe_cfr_sub_flat${flat}_$rep.log" 2>&1
    grep -E "identical|^grid|^sub|^auto" "$OUT/probe_cfr_sub_flat${flat}_$rep.log" | tee -a "$OUT/summary.txt"
  done
done
for wg in 1 101 251; do
  echo "-- stamps of workgroup $((wg-1))" | tee -a "$OUT/summary.txt"
  OSG_CFR_SUB_STAMPS=$wg PROBE_ONLY_SUB=1 timeout 300 python tools/probe_cfr_sub.py 2>&1 | grep -E "k_cfr_sub" | tail -4 | tee -a "$OUT/summary.txt"
done
echo "-- kuhn_poker(players=6), kuhn_poker(players=5), leduc_poker through the subtree kernel" | tee -a "$OUT/summary.txt"
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/summary.txt"
import time, sys
sys.path.insert(0, ".")
import open_spiel_amd as osa
ctx = osa.Context(0)
for game in ("kuhn_poker(players=6)", "kuhn_poker(players=5)", "leduc_poker"):
    for kern in ("grid", "sub"):
        s = osa.TabularSolver(ctx, game, general_kernel=kern)
        s.evaluate_and_update_policy(3); ctx.synchronize()
        t0 = time.perf_counter(); s.evaluate_and_update_policy(100); ctx.synchronize(); dt = time.perf_counter() - t0
        print(f"{game} [{kern}: {s.last_kernel()}] {100 / dt:.0f} it/s, {dt / 100 * 1e6:.1f} us per iteration, histories {s.num_histories}")
PY
du -sh "$OUT"
