#!/bin/bash
set -u
OUT=gpurun_out/r03c; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_cfr.py -q -m gpu -x --durations=5 > $OUT/pytest_cfr.log 2>&1; echo "pytest cfr rc $?"; tail -12 $OUT/pytest_cfr.log
timeout 300 python tools/probe_cfr.py > $OUT/probe_cfr.log 2>&1; cat $OUT/probe_cfr.log | cut -c1-220
timeout 900 python -m pytest tests -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc $?"; tail -14 $OUT/pytest_gpu.log
timeout 600 python tools/probe_kernels.py > $OUT/probe_kernels.log 2>&1; grep -E "n=2\^24|k_step hex|k_status hex" $OUT/probe_kernels.log | cut -c1-200
