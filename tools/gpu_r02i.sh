set -u
OUT=gpurun_out/r02i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_z1_gpu_reference_vectors.py tests/test_gpu_vector_env.py tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt; tail -4 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
timeout 900 python tools/probe_kernels.py > $OUT/probe_kernels.log 2>&1; echo "probe exit $?" | tee -a $OUT/summary.txt; grep -E "k_observation|k_step|k_legal|k_status" $OUT/probe_kernels.log | cut -c1-260 | tee -a $OUT/summary.txt
