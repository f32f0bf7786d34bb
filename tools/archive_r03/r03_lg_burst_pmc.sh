#!/bin/bash
# SQ counters of the ONE k_levelgen launch that makes a level for every env (the burst at step max_steps of a lock-step family):
#   tools/archive_r03/r03_lg_burst_pmc.sh <env id> <envs> <steps>     (steps > max_steps; the longest launch is picked; LG_PICK=median: the median one)
env_id=$1; envs=$2; steps=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
PY=$(readlink -f "$(command -v python3)")
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH" \
 "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_CYCLES SQ_LDS_DATA_FIFO_FULL" ; do
  i=$((i+1))
  rm -rf /tmp/lgb_$i
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/lgb_$i -- $PY $R/bench.py --steps $steps --warmup 2 --no-cpu-baseline --env $env_id --envs-per-gpu $envs --new-level-each-episode > /tmp/lgb_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/lgb_$i.log; continue; }
  f=$(find /tmp/lgb_$i -name "*counter_collection.csv" | head -n 1)
  python3 - "$f" <<'PY' | tee -a $R/gpurun_out/lg_burst_pmc.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_levelgen" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    by[r["Dispatch_Id"]]["_dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
import os
v = sorted(by.values(), key=lambda c: c["_dur"])
d = v[len(v) // 2] if os.environ.get("LG_PICK") == "median" else v[-1]   # LG_PICK=median: a steady-flow family's typical launch
print("%s launch: %.1f us under the counters" % ("median" if os.environ.get("LG_PICK") == "median" else "burst", d["_dur"] / 1e3))
for k in sorted(d):
    if k != "_dur": print("  %-28s %16.0f" % (k, d[k]))
PY
done
