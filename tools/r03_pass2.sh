#!/bin/bash
set -u
OUT=gpurun_out/r03b; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mcts.py tests/test_z5_gpu_mcts_evaluator.py tests/test_pyspiel_surface.py tests/test_z6_gpu_reference_tests_on_mirror.py tests/test_z7_gpu_exchange_steps.py tests/test_z2_gpu_dropin.py tests/test_host_api.py -q -m gpu -x --durations=8 > $OUT/pytest_subset.log 2>&1; echo "pytest rc $?"; tail -14 $OUT/pytest_subset.log
OSG_MCTS_SCHEDULE=lpt:7 timeout 300 python -m pytest tests/test_gpu_mcts.py tests/test_gpu_fullsize.py -q -m gpu -x -k "mcts or search" > $OUT/pytest_mcts_lpt.log 2>&1; echo "pytest lpt rc $?"; tail -3 $OUT/pytest_mcts_lpt.log
timeout 300 python tools/probe_config1.py > $OUT/config1.log 2>&1; grep -E '"value"|us_per_search|error' $OUT/config1.log | head -12
timeout 600 python tools/probe_mcts_schedule.py > $OUT/mcts_schedule.log 2>&1; cat $OUT/mcts_schedule.log
timeout 600 python tools/probe_hex_step.py > $OUT/hex_step.log 2>&1; cat $OUT/hex_step.log
