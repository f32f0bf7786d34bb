#!/bin/bash
# Usage (GPU box, repo root): tools/kernel_durations.sh <tag> <kernel substring> <python script + args...>
# Every launch of the matching kernel, in launch order: duration in us (rocprofv3 --kernel-trace) -> gpurun_out/durations_<tag>.txt + a histogram line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; SUB=$2; shift; shift
OUT=$R/gpurun_out/dur_$TAG
PY=$(readlink -f "$(command -v python3)")
S=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -- $PY $S "$@" > $R/gpurun_out/dur_$TAG.log 2>&1 || echo "rocprofv3 failed"
f=$(find $OUT -name "*kernel_trace.csv" | head -n 1)
python3 - "$f" "$SUB" > $R/gpurun_out/durations_$TAG.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("# %d launches of %s, us each, in launch order" % (len(d), sys.argv[2]))
print(" ".join("%.1f" % x for x in d))
s = sorted(d)
print("# total %.1f ms; median %.1f; p90 %.1f; p99 %.1f; max %.1f; launches > 100 us: %d (%.1f ms)" % (sum(d) / 1e3, s[len(s) // 2], s[int(len(s) * .9)], s[int(len(s) * .99)], s[-1], sum(x > 100 for x in d), sum(x for x in d if x > 100) / 1e3))
PY
rm -rf $OUT
tail -1 $R/gpurun_out/durations_$TAG.txt
