#!/bin/bash
# per-launch durations of k_levelgen over a run with a new level per episode:  tools/archive_r03/r03_lg_trace.sh <env id> <envs> <steps>
env_id=$1; envs=$2; steps=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
PY=$(readlink -f "$(command -v python3)")
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lgt
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/lgt -- $PY $R/bench.py --steps $steps --warmup 2 --no-cpu-baseline --env $env_id --envs-per-gpu $envs --new-level-each-episode > /tmp/lgt.log 2>&1 || { tail -3 /tmp/lgt.log; exit 1; }
f=$(find /tmp/lgt -name "*kernel_trace.csv" | head -n 1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/lg_trace.txt
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(sys.argv[1])) if "k_levelgen" in r["Kernel_Name"])
d = [x[1] for x in rows]
print("launches", len(d), "total %.1f ms" % (sum(d) / 1e3))
import collections
b = collections.Counter()
for x in d:
    b[10 ** len(str(int(x)))] += 1
print("durations by decade (us, upper bound):", dict(sorted(b.items())))
big = [(i, round(x, 1)) for i, x in enumerate(d) if x > 50]
print("launches over 50 us (index, us):", big[:120])
PY
