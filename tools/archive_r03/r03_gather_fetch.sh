#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the gather form (current build) and of its no-compute skeleton (tools/ubench/gather_window)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/gather_fetch; mkdir -p $O
PY=$(readlink -f "$(command -v python3)")
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf $O/t; timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/t -- $PY $R/bench.py --steps 24 --warmup 8 --no-cpu-baseline --env MiniGrid-FourRooms-v0 --envs-per-gpu 1048576 > $O/k_$n.log 2>&1
  f=$(find $O/t -name "*counter_collection.csv" | head -n 1); [ -n "$f" ] && { head -n 1 $f > $O/kernel_$n.csv; grep k_step $f >> $O/kernel_$n.csv; }
  rm -rf $O/t; timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/t -- $R/tools/ubench/gather_window > $O/s_$n.log 2>&1
  f=$(find $O/t -name "*counter_collection.csv" | head -n 1); [ -n "$f" ] && cp $f $O/skeleton_$n.csv
  rm -rf $O/t
done
echo done
