#!/bin/bash
set -u
OUT=gpurun_out/r03k; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc $?"; tail -12 $OUT/pytest_gpu.log
timeout 300 python tools/probe_mcts_evaluator.py > $OUT/mcts_evaluator.log 2>&1; cat $OUT/mcts_evaluator.log | grep -v amdgpu | cut -c1-330
timeout 300 python tools/probe_joint_breakdown.py 2>&1 | grep -v amdgpu | grep -v "stride [248]\|stride 16" | cut -c1-200 | tee $OUT/joint_breakdown.log
