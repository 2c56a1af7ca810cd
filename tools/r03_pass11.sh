#!/bin/bash
set -u
OUT=gpurun_out/r03j; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_z5_gpu_mcts_evaluator.py -q -m gpu -x > $OUT/pytest_z5.log 2>&1; echo "pytest z5 rc $?"; tail -5 $OUT/pytest_z5.log
timeout 300 python tools/probe_joint_breakdown.py 2>&1 | grep -v amdgpu | grep -v "stride [248]\|stride 16" | cut -c1-200 | tee $OUT/joint_breakdown.log
timeout 300 python tools/probe_mcts_evaluator.py > $OUT/mcts_evaluator.log 2>&1; cat $OUT/mcts_evaluator.log | grep -v amdgpu | cut -c1-330
