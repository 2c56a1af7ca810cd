#!/bin/bash
# Round 5, third GPU pass: row-width template of the kuhn kernel, two environments per thread, the lockstep search with volatile
# node accesses; a bench line.
set -u
OUT=gpurun_out/r05c
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== pytest (touched)" | tee "$OUT/summary.txt"
timeout 1500 python -m pytest tests/test_gpu_cfr.py tests/test_gpu_vector_env.py tests/test_gpu_timed_batch.py tests/test_z8_gpu_single_root_search.py tests/test_z5_gpu_mcts_evaluator.py tests/test_z9_gpu_stock_bots.py tests/test_z1_gpu_reference_vectors.py tests/test_z4_gpu_reference_vectors_r2.py -q -m gpu --durations=8 > "$OUT/pytest_c.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -14 "$OUT/pytest_c.log" | tee -a "$OUT/summary.txt"
echo "== A/B solvers: r4regs (round-4 kernels) / now" | tee -a "$OUT/summary.txt"
for rep in 1 2; do
  for v in r4regs now; do
    if [ $v = now ]; then unset OSG_VARIANT_LIB; else export OSG_VARIANT_LIB=tools/variants/libosg_$v.so; fi
    timeout 300 python tools/probe_cfr.py > "$OUT/probe_cfr_${v}_$rep.log" 2>&1
    echo "-- $v $rep"; grep -E "^kuhn_poker|split|MCCFR|players=3" "$OUT/probe_cfr_${v}_$rep.log" | cut -c1-150
  done
done 2>&1 | tee -a "$OUT/summary.txt"
unset OSG_VARIANT_LIB
echo "== env step: one / two environments per thread" | tee -a "$OUT/summary.txt"
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/summary.txt"
import os, sys, torch
sys.path.insert(0, ".")
import open_spiel_amd as osa
from open_spiel_amd._abi import check, lib
ctx = osa.Context(0)
n = 1 << 20
for form in ("x2", "x1"):
    if form == "x1": os.environ["OSG_ENV_STEP_X1"] = "1"
    eb = osa.StateBatch(ctx, "connect_four", n)
    reset = torch.ones(n, dtype=torch.uint8, device="cuda"); cur = torch.empty(n, dtype=torch.int8, device="cuda")
    typ = torch.empty(n, dtype=torch.uint8, device="cuda"); rew = torch.empty((n, 2), dtype=torch.float64, device="cuda")
    msk = torch.empty((n, 1), dtype=torch.int32, device="cuda"); acts = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    lut = torch.tensor([-1] + [(m & -m).bit_length() - 1 for m in range(1, 128)], dtype=torch.int32, device="cuda")
    def step(t):
        check(lib().osg_env_step(eb._h, acts.data_ptr(), reset.data_ptr(), 1, 0, t, cur.data_ptr(), typ.data_ptr(), rew.data_ptr(), msk.data_ptr()))
    for t in range(10):
        step(t); torch.index_select(lut, 0, msk[:, 0].to(torch.int64), out=acts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for t in range(10, 210):
        e0.record(); step(t); e1.record(); torch.index_select(lut, 0, msk[:, 0].to(torch.int64), out=acts)
        torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
    us = tot / 200 * 1e3
    print(f"k_env_step {form}: {us:.2f} us per 2^20 environments = {60 * n / us / 1e6:.2f} TB/s ({60 * n / us / 1e6 / 8:.3f} of 8 TB/s, 60 B per step)")
PY
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
print("mccfr parity", json.dumps(s["mccfr"].get("parity"))[:400])
print("env_step", json.dumps(s.get("env_step"))[:900])
print("leduc cfr", s["cfr"].get("leduc", {}).get("value"), "3p", s["cfr"].get("leduc_3_players", {}).get("value"))
print("judge", json.dumps(s.get("policy_evaluation", {}).get("per_game"))[:600])
PY
du -sh "$OUT"
