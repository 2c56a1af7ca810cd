#!/bin/bash
set -u
OUT=gpurun_out/r03h; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_z5_gpu_mcts_evaluator.py tests/test_pyspiel_surface.py -q -m gpu -x > $OUT/pytest_z5.log 2>&1; echo "pytest z5 rc $?"; tail -5 $OUT/pytest_z5.log
timeout 200 python tools/probe_advance_stride.py 2>&1 | grep -v amdgpu | grep "stride  1" 
timeout 300 python tools/probe_joint_breakdown.py 2>&1 | grep -v amdgpu | grep -v "stride [248]\|stride 16" | cut -c1-200
