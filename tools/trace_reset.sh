#!/bin/bash
# Usage (GPU box): tools/trace_reset.sh <env id> -- kernel trace of create + first reset + a few steps at 1 Mi envs
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_reset
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/bench.py --no-cpu-baseline --steps 8 --warmup 2 --env $1 > $OUT/run.log 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -n 1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for i, r in enumerate(rows[:40]):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(i, r["Kernel_Name"][:60], "start_ms=%.3f" % ((int(r["Start_Timestamp"]) - t0) / 1e6), "dur_us=%.1f" % (d / 1e3))
PY
