#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_dynobs (one counter per pass, --kernel-trace only):  tools/archive_r03/r03_dyn_traffic.sh <env id> <envs> [variant]
env_id=$1; envs=$2; v=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
PY=$(readlink -f "$(command -v python3)")
[ -n "$v" ] && export MGX_LIB=$R/ab/$v.so
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/dt_$c
  timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/dt_$c -- $PY $R/bench.py --steps 24 --warmup 8 --no-cpu-baseline --env $env_id --envs-per-gpu $envs > /tmp/dt_$c.log 2>&1 || { echo "$c failed"; tail -3 /tmp/dt_$c.log; continue; }
  f=$(find /tmp/dt_$c -name "*counter_collection.csv" | head -n 1)
  python3 - "$f" "$c" "$envs" "$env_id" <<'PY' | tee -a $R/gpurun_out/dyn_traffic.log
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[2]]
for k in ("k_dynobs", "k_step"):
    v = sorted(float(r["Counter_Value"]) for r in rows if k in r["Kernel_Name"])
    if v:
        m = v[len(v) // 2]
        mult = 2 if sys.argv[2] == "FETCH_SIZE" else 1   # (FETCH_SIZE x 2 on gfx950: MI355X_MICROARCH.md, HBM section)
        print("%-36s %-10s %-10s median %10.0f KiB -> %6.1f B per env (x%d)" % (sys.argv[4], k, sys.argv[2], m, m * 1024 * mult / float(sys.argv[3]), mult))
PY
done
