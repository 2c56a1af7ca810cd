#!/bin/bash
# round 3, first GPU pass: the whole GPU suite, the N = 1 bench line (with the in-run counter passes), and the
# N = 2 line as the driver starts it (two ranks sharing this box's GPU over gloo)
set -u
OUT=gpurun_out/r03a; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --durations=12 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_n1.log 2> $OUT/bench_n1.err; echo "bench n1 rc $?"; tail -c 600 $OUT/bench_n1.err
OSG_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 > $OUT/bench_n2_gloo.log 2> $OUT/bench_n2_gloo.err; echo "bench n2 rc $?"; tail -c 600 $OUT/bench_n2_gloo.err
python - <<'PY'
import json
for f in ("gpurun_out/r03a/bench_n1.log", "gpurun_out/r03a/bench_n2_gloo.log"):
    try:
        line = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no line", e); continue
    r = line["roofline"]
    print(f, "value %.3e n_gpus %d frac %.3f traffic %s live %s" % (line["value"], line["n_gpus"], r["frac"], r["traffic"], r.get("traffic_measured_in_this_run")))
    print(" persistent", line.get("persistent", {}).get("roofline"))
    s = line.get("secondary", {})
    print(" secondary keys", list(s))
    if "error" in s: print(s["error"], s.get("traceback"))
    for k in ("mcts", "mccfr", "ttt_mcts"):
        if k in s:
            d = dict(s[k]); d.pop("roofline", None); d.pop("cpu_baseline", None)
            print(" ", k, json.dumps(d)[:1500])
PY
