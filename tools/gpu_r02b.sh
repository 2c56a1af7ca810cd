set -u
OUT=gpurun_out/r02b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
timeout 300 ./tools/step_sweep > $OUT/step_sweep_2p20.log 2>&1; cat $OUT/step_sweep_2p20.log | tee -a $OUT/summary.txt
timeout 300 ./tools/step_sweep 16777216 > $OUT/step_sweep_2p24.log 2>&1; grep -E "fused|stored-result |memory-only|copy|streams|1 stream" $OUT/step_sweep_2p24.log | tee -a $OUT/summary.txt
timeout 300 ./tools/clock_probe > $OUT/clock_probe.log 2>&1; cat $OUT/clock_probe.log | tee -a $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench_n1.log 2> $OUT/bench_n1.err; echo "bench exit $?" | tee -a $OUT/summary.txt; tail -c 6000 $OUT/bench_n1.log | tee -a $OUT/summary.txt; tail -5 $OUT/bench_n1.err
