#!/bin/bash
# Usage (GPU box, repo root): tools/fetch_pmc.sh <tag> <kernel substring> <bench.py args...>  -- FETCH_SIZE / WRITE_SIZE per launch of the matching kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; SUB=$2; shift; shift
OUT=$R/gpurun_out/fetch_$TAG
mkdir -p $OUT
PY=$(readlink -f "$(command -v python3)")
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/tmp
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/tmp -- $PY $R/bench.py --steps 24 --warmup 8 --no-cpu-baseline "$@" > $OUT/$c.log 2>&1 || echo "pmc $c failed"
  f=$(find $OUT/tmp -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && { head -n 1 $f > $OUT/${TAG}_$c.csv; grep -E "$SUB" $f >> $OUT/${TAG}_$c.csv; }
  rm -rf $OUT/tmp
done
python3 - <<PY
import csv, statistics
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open("$OUT/${TAG}_%s.csv" % c)) if r["Counter_Name"] == c]
    out[c] = statistics.median(v) if v else float("nan")
print("$TAG  %s: FETCH_SIZE x 2 = %.1f MB, WRITE_SIZE = %.1f MB per launch (KB counters; gfx950: FETCH doubled)" % ("$SUB", out["FETCH_SIZE"] * 2 / 1024, out["WRITE_SIZE"] / 1024))
PY
