#!/bin/bash
# Usage (GPU box, repo root): tools/kt.sh <tag> <python script + args...>  -- rocprofv3 kernel stats of any script into gpurun_out/kt_<tag>.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/kt_$TAG
PY=$(readlink -f "$(command -v python3)")
S=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- $PY $S "$@" > $R/gpurun_out/kt_$TAG.log 2>&1 || echo "rocprofv3 failed"
f=$(find $OUT -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $R/gpurun_out/kt_$TAG.csv
rm -rf $OUT
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$R/gpurun_out/kt_$TAG.csv")))[:12]:
    print("%-60s calls %6s avg %9.1f us  total %9.1f ms" % (r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
