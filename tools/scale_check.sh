#!/bin/bash
# First contact with an N-GPU node (the driver's SCALE record runs `bench.py --gpus N` for N = 1, 2, 4, 8):
# runs the same commands and prints, per N, what to compare —
#   * N = 1 `value` against the driver's BENCH record of the same box (box noise: a few %);
#   * per_rank.env_steps_per_s[] (every rank ~ the N = 1 value: the states shard with no collective);
#   * secondary.mcts.strong_scaling_efficiency (target >= 0.70 at N = 8 on hex(9) MCTS);
#   * secondary.mccfr.allreduce_us {rccl, oneshot} and rccl_world == N (RCCL really carried the exchange step);
#   * the world-2 exchange tests that skip on a one-GPU box.
# Usage: tools/scale_check.sh [max_gpus]      (writes gpurun_out/scale_check/)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/scale_check
mkdir -p "$OUT"
HAVE=$(python3 -c "import torch; print(torch.cuda.device_count())")
MAX=${1:-$HAVE}
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 1 2 4 8; do
  [ "$N" -gt "$MAX" ] && break
  [ "$N" -gt "$HAVE" ] && break
  if [ "$N" -eq 1 ]; then
    python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_n$N.log" 2> "$OUT/bench_n$N.err"
  else
    PORT=$(python3 -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
      bench.py --gpus "$N" --steps 20 --warmup 5 > "$OUT/bench_n$N.log" 2> "$OUT/bench_n$N.err"
  fi
  echo "N=$N rc=$?"
  tail -1 "$OUT/bench_n$N.log" | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
s = d.get('secondary', {})
print(json.dumps({'n_gpus': d['n_gpus'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'rccl_world': d.get('rccl_world'),
                  'collective_backend': d.get('collective_backend'),
                  'per_rank_env_steps_per_s': (d.get('per_rank') or {}).get('env_steps_per_s'),
                  'mcts_sims_per_s': (s.get('mcts') or {}).get('value'),
                  'mcts_strong_scaling_efficiency': (s.get('mcts') or {}).get('strong_scaling_efficiency'),
                  'mccfr_trajectories_per_s': (s.get('mccfr') or {}).get('value'),
                  'mccfr_allreduce_us': (s.get('mccfr') or {}).get('allreduce_us'),
                  'line_chars': len(json.dumps(d))}))
" | tee "$OUT/summary_n$N.json"
done
if [ "$HAVE" -ge 2 ]; then
  python3 -m pytest tests/test_z7_gpu_exchange_steps.py -q -m gpu -k "rccl" 2>&1 | tail -3 | tee "$OUT/pytest_rccl.log"
fi
python3 - "$OUT" <<'PY'
import json, os, sys
out = sys.argv[1]
rows = {}
for n in (1, 2, 4, 8):
    p = os.path.join(out, f"summary_n{n}.json")
    if os.path.exists(p):
        rows[n] = json.load(open(p))
if 1 in rows:
    for n, r in rows.items():
        r["weak_scaling_efficiency_env_steps"] = r["value"] / (n * rows[1]["value"])
        if r.get("mcts_sims_per_s") and rows[1].get("mcts_sims_per_s"):
            r["mcts_efficiency_vs_n1_run"] = r["mcts_sims_per_s"] / (n * rows[1]["mcts_sims_per_s"])
json.dump(rows, open(os.path.join(out, "scale_summary.json"), "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
