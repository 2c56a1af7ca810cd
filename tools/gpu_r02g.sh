set -u
OUT=gpurun_out/r02g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_z5_gpu_mcts_evaluator.py tests/test_z3_gpu_comm.py -q -m gpu -s > $OUT/pytest_z5.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt; tail -60 $OUT/pytest_z5.log | cut -c1-300 | tee -a $OUT/summary.txt
