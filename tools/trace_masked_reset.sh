set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_masked
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tools/reset_bench.py > $OUT/run.log 2>&1
f=$(find $OUT/t -name "*kernel_stats.csv" | head -n 1)
cut -c1-150 $f | head -n 24
