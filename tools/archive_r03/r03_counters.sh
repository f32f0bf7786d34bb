#!/bin/bash
# Round 3: FETCH / WRITE passes of the gather form and SQ counters of the 16x16 / gather instances (subset of tools/profile_round.sh).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
PY=$(readlink -f "$(command -v python3)")
cd /tmp && export TMPDIR=/tmp
pmc() { local name=$1 c=$2; shift; shift
  rm -rf $OUT/tmp_pmc
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/tmp_pmc -- $PY $R/bench.py --steps 24 --warmup 8 --no-cpu-baseline "$@" > $OUT/pmc_${name}_$c.log 2>&1 || { echo "pmc $name $c failed"; return 0; }
  local f=$(find $OUT/tmp_pmc -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && { head -n 1 $f > $OUT/${TAG}_pmc_${name}_$c.csv; grep -E "k_step|k_dynobs|k_levelgen" $f >> $OUT/${TAG}_pmc_${name}_$c.csv || true; }
  rm -rf $OUT/tmp_pmc; echo "pmc $name $c ok"; }
for c in FETCH_SIZE WRITE_SIZE; do
  pmc multiroom_n6_256k $c --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 262144
  pmc fourrooms_1M $c --env MiniGrid-FourRooms-v0 --envs-per-gpu 1048576
done
for w in "empty16 --env MiniGrid-Empty-16x16-v0 --envs-per-gpu 524288" "keycorridor_s6r3 --env MiniGrid-KeyCorridorS6R3-v0 --envs-per-gpu 524288" \
         "obstructed_2dlhb --env MiniGrid-ObstructedMaze-2Dlhb-v0 --envs-per-gpu 262144" "fourrooms1m --env MiniGrid-FourRooms-v0 --envs-per-gpu 1048576" \
         "multiroom_n6 --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 262144"; do
  set -- $w; n=$1; shift
  (cd $R && tools/pmc_quick2.sh ${TAG}_$n "$@" > /dev/null && python3 tools/pmc_summary.py ${TAG}_$n k_ > $OUT/${TAG}_sq_counters_$n.txt) || echo "sq $n failed"
  echo "sq $n ok"
done
echo "done counters $TAG"
