#!/bin/bash
# Which workload makes the process crash at exit under rocprofv3 --kernel-trace (seen once: bench.py, after the tool's finalization)?
export TMPDIR=/tmp
run() {
  timeout 300 rocprofv3 --kernel-trace -d /tmp/exit_probe_$1 -- python -c "
import open_spiel_amd as osa
ctx = osa.Context(0)
$2
ctx.synchronize()
" > /tmp/exit_probe_$1.log 2>&1
  echo "$1: exit $?"
}
run split   's = osa.TabularSolver(ctx, "leduc_poker"); s.evaluate_and_update_policy(5)'
run sub     's = osa.TabularSolver(ctx, "leduc_poker(players=3)"); s.evaluate_and_update_policy(2)'
run judge   's = osa.TabularSolver(ctx, "leduc_poker", general_kernel="path"); s.evaluate_and_update_policy(2); print(s.nash_conv())'
run kuhn    's = osa.TabularSolver(ctx, "kuhn_poker"); s.evaluate_and_update_policy(5); print(s.nash_conv())'
run create  's = osa.TabularSolver(ctx, "leduc_poker")'
run batch   'b = osa.StateBatch(ctx, "connect_four", 1024); b.random_steps(3, 5)'
