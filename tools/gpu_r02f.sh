set -u
OUT=gpurun_out/r02f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/probe_mccfr_quality.py > $OUT/mccfr_quality.log 2>&1; echo "exit $?" | tee $OUT/summary.txt; cat $OUT/mccfr_quality.log | tee -a $OUT/summary.txt
