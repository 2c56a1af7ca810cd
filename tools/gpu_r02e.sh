set -u
OUT=gpurun_out/r02e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_z4_gpu_reference_vectors_r2.py -q -m gpu > $OUT/pytest_z4.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt; tail -40 $OUT/pytest_z4.log | cut -c1-300 | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_z4_gpu_reference_vectors_r2.py > $OUT/pytest_rest.log 2>&1; echo "pytest rest exit $?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_rest.log | cut -c1-300 | tee -a $OUT/summary.txt
