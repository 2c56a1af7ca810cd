set -u
OUT=gpurun_out/r02k; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "fused_step or connect_four or agree" > $OUT/pytest_step.log 2>&1; tail -1 $OUT/pytest_step.log
timeout 30 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 70 python bench.py > $OUT/bench_n1.log 2> $OUT/bench_n1.err; echo bench rc $?
timeout 60 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python bench.py --no-cpu-baseline --no-secondary > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" > $OUT/kernel_stats.csv 2>> $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE; do timeout 40 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python bench.py --steps 1 --warmup 1 --no-secondary --no-cpu-baseline > $OUT/pmc_$C.log 2>&1; done
find $OUT -name '*.db' -size +20M -delete 2>/dev/null
grep "k_step_c4std" $OUT/kernel_stats.csv | tail -3 | cut -c1-40,100-250
