#!/bin/bash
set -u
OUT=gpurun_out/r03g; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/probe_joint_breakdown.py > $OUT/joint_breakdown.log 2>&1; cat $OUT/joint_breakdown.log | cut -c1-200
timeout 300 python -m pytest tests/test_z5_gpu_mcts_evaluator.py tests/test_pyspiel_surface.py tests/test_z7_gpu_exchange_steps.py -q -m gpu -x > $OUT/pytest_z5.log 2>&1; echo "pytest z5 rc $?"; tail -5 $OUT/pytest_z5.log
