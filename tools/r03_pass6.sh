#!/bin/bash
set -u
OUT=gpurun_out/r03f; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/debug_joint.py > $OUT/debug_joint.log 2>&1; tail -4 $OUT/debug_joint.log | cut -c1-200
timeout 300 python tools/probe_joint_breakdown.py > $OUT/joint_breakdown.log 2>&1; cat $OUT/joint_breakdown.log | cut -c1-200
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -- python $GRAFT_REPO_ROOT/tools/probe_joint_breakdown.py 65536 30 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find $OUT/trace -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" > $OUT/kernel_stats.csv 2>> $OUT/trace.log; head -25 $OUT/kernel_stats.csv | cut -c1-260
find $OUT -name '*.db' -size +20M -delete 2>/dev/null
