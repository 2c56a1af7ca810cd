#!/bin/bash
# Round 5, GPU pass o: the environment step of the standard connect_four board on the fused step.
set -u
OUT=gpurun_out/${1:-r05o}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "== pytest (vector env, fullsize env)" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_vector_env.py tests/test_gpu_fullsize.py -q -m gpu -k "env" --durations=5 > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -6 "$OUT/pytest.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for rep in 1 2; do timeout 300 python tools/probe_env_step.py 2>&1 | grep k_env_step | tee -a "$OUT/summary.txt"; done
PROBE_LOG_N=24 timeout 300 python tools/probe_env_step.py 2>&1 | grep k_env_step | tee -a "$OUT/summary.txt"
du -sh "$OUT"
echo "== bench.py (env_step roofline)" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py --no-pmc > "$OUT/bench_n1.log" 2> "$OUT/bench_n1.err"
echo "bench exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench_n1.log" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
line = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
e = line["secondary"]["env_step"]
print("env_step value", e["value"], "us_per_launch", e["us_per_launch"])
print(json.dumps({k: v for k, v in e["roofline"].items() if k != "note"}))
print("headline", line["value"], line["roofline"]["frac"], line["roofline"].get("hbm_frac"))
PY
