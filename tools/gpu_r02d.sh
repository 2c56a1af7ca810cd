set -u
OUT=gpurun_out/r02d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mcts.py tests/test_host_api.py tests/test_z2_gpu_dropin.py -q -m gpu -x > $OUT/pytest_mcts.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt; tail -30 $OUT/pytest_mcts.log | cut -c1-400 | tee -a $OUT/summary.txt
