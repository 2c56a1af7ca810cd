#!/bin/bash
# The standard GPU validation pass, as ONE gpurun command (what the first GPU call of a session should be):
#
#   gpurun --timeout 2400 -- 'bash tools/gpu_validation.sh r02'
#
# 0. tools/kernel_resources.py --check (registers / spills / scratch of every entry point; a hot kernel that spills fails);
#    __graft_entry__.smoke();
# 1. the GPU test suite (parity vs the oracle, vs the recorded outputs of the genuine reference,
#    checkpoint interop with oracle/_ref, the world-size-1 collective);
# 2. bench.py (one JSON line: env-steps/s + roofline + cpu_baseline of the genuine reference build);
# 3. rocprofv3 --kernel-trace --stats of the same bench command (per-kernel average durations);
# 4. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) -> tools/pmc_traffic.py.
# Never combines --pmc with trace domains other than kernel-trace (gpurun refuses that combination).
# Everything lands under gpurun_out/<tag>/; copy what should be judged into profiles/ and commit it.
set -u
TAG=${1:-r00}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp

echo "== kernel resources (code object metadata; no hot kernel may spill)" | tee "$OUT/summary.txt"
python tools/kernel_resources.py --check --out "$OUT/kernel_resources.txt"
echo "kernel_resources --check exit $?" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/kernel_resources.txt" | cut -c1-200 | tee -a "$OUT/summary.txt"

echo "== smoke()" | tee -a "$OUT/summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?" | tee -a "$OUT/summary.txt"
tail -2 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"

echo "== pytest -m gpu" | tee -a "$OUT/summary.txt"
timeout 1800 python -m pytest tests -q -m gpu --durations=15 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -5 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"

echo "== bench.py" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py > "$OUT/bench_n1.log" 2> "$OUT/bench_n1.err"
echo "bench exit $?" | tee -a "$OUT/summary.txt"
tail -c 3000 "$OUT/bench_n1.log" | tee -a "$OUT/summary.txt"

echo "== rocprofv3 --kernel-trace --stats" | tee -a "$OUT/summary.txt"
# (OSG_CFR_PLAIN_LAUNCH=1: on ROCm 7.0.2 a process that made a COOPERATIVE launch crashes in an exit handler under
#  rocprofv3 --kernel-trace — after the tool has written its output — tools/probe_exit_under_tracer.sh; the traced run
#  launches the two barrier kernels with ordinary launches, same kernels, same durations)
OSG_CFR_PLAIN_LAUNCH=1 timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python bench.py --no-cpu-baseline > "$OUT/trace.log" 2>&1
echo "trace exit $?" | tee -a "$OUT/summary.txt"
DB=$(find "$OUT/trace" -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py "$DB" > "$OUT/kernel_stats.csv" 2>> "$OUT/trace.log"; fi
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_rocprof.csv" \; 2>/dev/null
head -12 "$OUT/kernel_stats.csv" 2>/dev/null | tee -a "$OUT/summary.txt"
python tools/launch_timing.py "$OUT" > "$OUT/launch_timing.txt" 2>&1; cat "$OUT/launch_timing.txt" | tee -a "$OUT/summary.txt"

for C in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $C" | tee -a "$OUT/summary.txt"
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -- \
    python bench.py --steps 1 --warmup 1 --no-secondary --no-cpu-baseline > "$OUT/pmc_$C.log" 2>&1
  echo "pmc $C exit $?" | tee -a "$OUT/summary.txt"
done
python tools/pmc_traffic.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$TAG" > "$OUT/pmc_traffic.log" 2>&1
tail -30 "$OUT/pmc_traffic.log" | tee -a "$OUT/summary.txt"

# instruction mix of the hex(9) search kernel (8192 roots x 1024 simulations), four separate counter passes
P=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_BRANCH"; do
  P=$((P+1))
  echo "== rocprofv3 --pmc $C (k_mcts_wave)" | tee -a "$OUT/summary.txt"
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_mcts_$P" -- python tools/probe_mcts_one.py > "$OUT/pmc_mcts_$P.log" 2>&1
  echo "pmc mcts $P exit $?" | tee -a "$OUT/summary.txt"
done
python tools/pmc_mcts.py "$OUT" "$TAG" 2>&1 | tee -a "$OUT/summary.txt"

echo "== counters of the solver kernels of configs 3 and 5 (k_cfr_small, k_mccfr_resident_flat)" | tee -a "$OUT/summary.txt"
bash tools/pmc_solvers.sh "$TAG" 2>&1 | cut -c1-400 | tee -a "$OUT/summary.txt"

echo "== the profiles bench.py quotes describe the kernels as built (tools/profile_sources.py)" | tee -a "$OUT/summary.txt"
python tools/profile_sources.py | tee -a "$OUT/summary.txt"
if [ "${PIPESTATUS[0]}" -ne 0 ]; then echo "STALE PROFILE SOURCES: validation FAILED" | tee -a "$OUT/summary.txt"; STALE=1; fi
cp profiles/${TAG}_pmc_solvers.json* profiles/${TAG}_pmc_k_mcts_wave_hex9_8192x1024.csv* "$OUT/" 2>/dev/null

echo "== probes" | tee -a "$OUT/summary.txt"
timeout 600 python tools/probe_mcts_bench.py > "$OUT/mcts_bench.log" 2>&1; grep hex "$OUT/mcts_bench.log" | tee -a "$OUT/summary.txt"
timeout 600 python tools/probe_cfr.py > "$OUT/probe_cfr.log" 2>&1; tail -12 "$OUT/probe_cfr.log" | cut -c1-200 | tee -a "$OUT/summary.txt"
timeout 600 python tools/probe_kernels.py > "$OUT/probe_kernels.log" 2>&1; grep "2^24" "$OUT/probe_kernels.log" | cut -c1-220 | tee -a "$OUT/summary.txt"
timeout 600 python tools/probe_mcts_evaluator.py > "$OUT/mcts_evaluator.log" 2>&1; tail -12 "$OUT/mcts_evaluator.log" | cut -c1-220 | tee -a "$OUT/summary.txt"
timeout 300 python tools/probe_judge.py > "$OUT/probe_judge.log" 2>&1; grep -v amdgpu.ids "$OUT/probe_judge.log" | tail -4 | tee -a "$OUT/summary.txt"
timeout 300 python tools/probe_cfr_sub.py > "$OUT/probe_cfr_sub.log" 2>&1; grep -E "^grid|^sub|^auto" "$OUT/probe_cfr_sub.log" | tee -a "$OUT/summary.txt"
timeout 300 python tools/probe_obs_lds.py > "$OUT/probe_obs_lds.log" 2>&1; grep -v amdgpu.ids "$OUT/probe_obs_lds.log" | tee -a "$OUT/summary.txt"
timeout 300 python tools/probe_single_root.py > "$OUT/probe_single_root.log" 2>&1; tail -6 "$OUT/probe_single_root.log" | cut -c1-200 | tee -a "$OUT/summary.txt"
timeout 300 python tools/probe_hex_step.py > "$OUT/hex_step.log" 2>&1; grep "default" "$OUT/hex_step.log" | tee -a "$OUT/summary.txt"
# keep the merged-back directory small (gpurun merges at most 64 MiB)
find "$OUT" -name '*.db' -size +20M -delete 2>/dev/null
du -sh "$OUT" | tee -a "$OUT/summary.txt"
exit ${STALE:-0}
