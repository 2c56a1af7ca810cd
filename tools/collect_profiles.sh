#!/bin/bash
# Copies what a tools/gpu_validation.sh pass left under gpurun_out/<tag>/ into profiles/ under the round's naming
# (gpurun_out/ is scratch; profiles/ is what is committed and judged).   usage: tools/collect_profiles.sh r05e
set -u
TAG=${1:?tag}
SRC=gpurun_out/$TAG
cd "$(dirname "$0")/.."
cp_if() { [ -f "$1" ] && cp "$1" "$2"; }
cp_if $SRC/summary.txt            profiles/${TAG}_gpu_validation_summary.txt
cp_if $SRC/pytest_gpu.log         profiles/${TAG}_pytest_gpu.log
cp_if $SRC/bench_n1.log           profiles/${TAG}_bench_n1.log
cp_if $SRC/kernel_stats.csv       profiles/${TAG}_bench_kernel_stats.csv
cp_if $SRC/kernel_stats_rocprof.csv profiles/${TAG}_bench_kernel_stats_rocprof.csv
cp_if $SRC/launch_timing.txt      profiles/${TAG}_launch_timing.txt
cp_if $SRC/kernel_resources.txt   profiles/${TAG}_kernel_resources.txt
cp_if $SRC/pmc_traffic.log        profiles/${TAG}_pmc_traffic.log
for f in mcts_bench probe_cfr probe_cfr_sub probe_judge probe_kernels probe_obs_lds probe_single_root hex_step mcts_evaluator; do
  cp_if $SRC/$f.log profiles/${TAG}_$f.log
done
for C in FETCH_SIZE WRITE_SIZE; do
  f=$(find $SRC/pmc_$C -name '*counter_collection.csv' 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" profiles/${TAG}_pmc_${C}_k_step_c4std.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_step_c4std" in r.get("Kernel_Name", "")]
if rows:
    w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows[:400])
PY
done
ls -la profiles | grep "${TAG}_" | awk '{print $5, $9}'
