#!/bin/bash
# Instruction counts of one hex(9) search (tools/probe_mcts_one.py: 8192 roots x 1024 simulations) per library
# variant: rocprofv3 --pmc passes, kernel-trace only.   bash tools/pmc_variants.sh <tag> lib1.so lib2.so ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for LIB in "$@"; do
  N=$(basename $LIB .so)
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    D=$OUT/${N}_$(echo $C | tr ' ' '_')
    OSG_VARIANT_LIB=$LIB timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python tools/probe_mcts_one.py > $D.log 2>&1
  done
  python - "$OUT" "$N" <<'PY'
import csv, glob, os, sys
out, n = sys.argv[1], sys.argv[2]
tot = {}
for f in glob.glob(os.path.join(out, n + "_SQ*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mcts_wave" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(n, " ".join(f"{k}={v / (8192 * 1024):.1f}" for k, v in sorted(tot.items())), flush=True)
PY
done
