#!/bin/bash
# Usage (GPU box, repo root): tools/seed_prof.sh <tag> [env_id] [n_envs]  -- kernel stats + SQ / FETCH / WRITE counters of a full re-seeding reset
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; ENV=${2:-MiniGrid-DoorKey-8x8-v0}; N=${3:-1048576}
OUT=$R/gpurun_out/seed_$TAG
mkdir -p $OUT
PY=$(readlink -f "$(command -v python3)")
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $PY $R/tools/seed_bench.py $ENV $N 8 > $OUT/kt.log 2>&1 || echo "kernel-trace failed"
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $OUT/p$i
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- $PY $R/tools/seed_bench.py $ENV $N 3 > $OUT/p$i.log 2>&1 || echo "pmc pass $i failed"
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && { head -n 1 $f > $OUT/${TAG}_pmc$i.csv; grep -E "k_seed" $f >> $OUT/${TAG}_pmc$i.csv; }
  rm -rf $OUT/p$i
done
rm -rf $OUT/kt
echo done
