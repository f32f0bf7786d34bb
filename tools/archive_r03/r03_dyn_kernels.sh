#!/bin/bash
# Per-kernel durations (rocprofv3 --kernel-trace --stats) of the Dynamic-Obstacles step for builds of libmgx.so under ab/:
#   tools/archive_r03/r03_dyn_kernels.sh <env id> <envs> <name> <name> ...     (MGX_LIB is read by gym_minigrid_amd/_lib.py)
env_id=$1; envs=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
PY=$(readlink -f "$(command -v python3)")
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export MGX_LIB=$R/ab/$v.so
  rm -rf /tmp/dk_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dk_$v -- $PY $R/bench.py --no-cpu-baseline --env $env_id --envs-per-gpu $envs --steps 200 --warmup 20 > /tmp/dk_$v.log 2>&1 || { echo "$v failed"; tail -3 /tmp/dk_$v.log; continue; }
  f=$(find /tmp/dk_$v -name "*kernel_stats.csv" | head -n 1)
  python3 - "$f" "$v" "$env_id" <<'PY' | tee -a $R/gpurun_out/dyn_kernels.log
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 100 and ("k_dynobs" in r["Name"] or "k_step" in r["Name"]):
        print("%-36s %-10s %-40s %8.2f us x %s" % (sys.argv[3], sys.argv[2], r["Name"][:40], float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
done
