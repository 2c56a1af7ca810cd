#!/bin/bash
set -u
OUT=gpurun_out/r03i; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
P=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum"; do
  P=$((P+1))
  OSG_PROBE_ONE=1 timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$P -- python $ROOT/tools/probe_advance_stride.py > $ROOT/$OUT/pmc_$P.log 2>&1
  echo "pass $P ($C) rc $?"
done
cd $ROOT
python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/r03i/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mcts_advance" not in r["Kernel_Name"]: continue
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
names = sorted({k for d in rows.values() for k in d})
print("dispatch", *names)
for j, i in enumerate(ids[:101]):
    if j in (0, 1, 2, 3, 6, 7, 8, 9, 11, 13, 20, 30, 40, 50, 60, 70, 90):
        print(j, *[f"{rows[i].get(k, float('nan')):.3g}" for k in names])
PY
