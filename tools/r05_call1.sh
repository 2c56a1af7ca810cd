#!/bin/bash
# Round 5, first GPU pass: the new parity tests, the world-8-on-one-device tests, A/B of the register work, a bench line.
set -u
OUT=gpurun_out/r05a
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
python tools/kernel_resources.py > "$OUT/kernel_resources.txt" 2>&1
echo "== pytest (new + touched)" | tee "$OUT/summary.txt"
timeout 1200 python -m pytest tests/test_gpu_timed_batch.py tests/test_gpu_cfr.py tests/test_z10_gpu_oneshot_allreduce.py tests/test_z11_gpu_world8_one_device.py -q -m gpu --durations=12 -s > "$OUT/pytest_a.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
grep -E "config [35]:|passed|failed|error" "$OUT/pytest_a.log" | tail -12 | tee -a "$OUT/summary.txt"
echo "== A/B leduc / kuhn CFR (round-4 register behaviour vs now)" | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  OSG_VARIANT_LIB=tools/variants/libosg_r4regs.so timeout 300 python tools/probe_cfr.py > "$OUT/probe_cfr_r4regs_$rep.log" 2>&1
  timeout 300 python tools/probe_cfr.py > "$OUT/probe_cfr_now_$rep.log" 2>&1
  echo "-- r4regs $rep"; grep -E "^kuhn_poker CFR|split" "$OUT/probe_cfr_r4regs_$rep.log" | cut -c1-160
  echo "-- now $rep"; grep -E "^kuhn_poker CFR|split" "$OUT/probe_cfr_now_$rep.log" | cut -c1-160
done 2>&1 | tee -a "$OUT/summary.txt"
echo "== A/B hex(9) search (round-4 osg_mcts_wave.hip vs now)" | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  OSG_VARIANT_LIB=tools/variants/libosg_r4wave.so timeout 300 python tools/probe_mcts_bench.py > "$OUT/mcts_bench_r4wave_$rep.log" 2>&1
  timeout 300 python tools/probe_mcts_bench.py > "$OUT/mcts_bench_now_$rep.log" 2>&1
  echo "-- r4wave $rep"; grep hex "$OUT/mcts_bench_r4wave_$rep.log"
  echo "-- now $rep"; grep hex "$OUT/mcts_bench_now_$rep.log"
done 2>&1 | tee -a "$OUT/summary.txt"
echo "== counters of the solver kernels" | tee -a "$OUT/summary.txt"
bash tools/pmc_solvers.sh r05a 2>&1 | tee -a "$OUT/summary.txt"
echo "== bench.py" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py > "$OUT/bench_n1.log" 2> "$OUT/bench_n1.err"
echo "bench exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench_n1.log" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
line = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = line["roofline"]
print("value", line["value"], "frac", r["frac"], "hbm_frac", r.get("hbm_frac"), "parity states", line["parity_checked_states"])
s = line["secondary"]
for k in ("cfr", "mccfr", "mcts"):
    d = s[k]
    print(k, d["value"], {x: d.get(x) for x in ("parity_checked_iterations", "parity_checked_trajectories", "parity_checked_roots", "kernel")})
    print("   parity:", json.dumps(d.get("parity"))[:600])
    print("   roofline:", json.dumps(d.get("roofline"))[:900])
print("leduc cfr", s["cfr"].get("leduc", {}).get("value"), "3p", s["cfr"].get("leduc_3_players", {}).get("value"))
PY
du -sh "$OUT"
