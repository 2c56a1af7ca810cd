#!/bin/bash
set -u
OUT=gpurun_out/r03e; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/debug_joint.py > $OUT/debug_joint.log 2>&1; tail -60 $OUT/debug_joint.log | cut -c1-200
for b in reference_hex_test reference_leduc_poker_test_on_mirror reference_basic_tests_boards_on_mirror; do
  T0=$SECONDS; timeout 900 tests/_refbuilt/$b > $OUT/$b.log 2> $OUT/$b.err; echo "$b rc $? $((SECONDS-T0)) s"; tail -2 $OUT/$b.err | cut -c1-300; tail -1 $OUT/$b.log | cut -c1-200
done
timeout 300 python tools/probe_mcts_evaluator.py > $OUT/mcts_evaluator.log 2>&1; head -3 $OUT/mcts_evaluator.log | cut -c1-500
