#!/bin/bash
set -u
OUT=gpurun_out/r03d; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
for b in reference_cfr_test_on_mirror reference_tabular_exploitability_test reference_best_response_test_on_mirror reference_hex_test reference_kuhn_poker_test reference_leduc_poker_test_on_mirror reference_basic_tests_boards_on_mirror; do
  T0=$SECONDS; timeout 900 tests/_refbuilt/$b > $OUT/$b.log 2> $OUT/$b.err; echo "$b rc $? $((SECONDS-T0)) s"; tail -2 $OUT/$b.err | cut -c1-300; tail -1 $OUT/$b.log | cut -c1-200
done
timeout 600 python -m pytest tests/test_pyspiel_surface.py tests/test_host_api.py tests/test_z2_gpu_dropin.py tests/test_gpu_cfr.py -q -m gpu -x > $OUT/pytest_subset.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_subset.log
timeout 600 python -m pytest tests/test_z5_gpu_mcts_evaluator.py tests/test_pyspiel_surface.py -q -m gpu -x > $OUT/pytest_z5.log 2>&1; echo "pytest z5 rc $?"; tail -12 $OUT/pytest_z5.log
timeout 600 python tools/probe_mcts_evaluator.py > $OUT/mcts_evaluator.log 2>&1; cat $OUT/mcts_evaluator.log | cut -c1-330
