set -u
OUT=gpurun_out/r02h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pyspiel_surface.py tests/test_host_api.py tests/test_z2_gpu_dropin.py tests/test_z5_gpu_mcts_evaluator.py -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt; tail -60 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
