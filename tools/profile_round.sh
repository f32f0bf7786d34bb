#!/bin/bash
# Usage (GPU box, repo root): tools/profile_round.sh r02 [part]   -- regenerates the rocprofv3 evidence kept under profiles/
#   part = stats | dyn | newlevel | pmc | sq | all (default).  Writes gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
# Every rocprofv3 command has `python3 bench.py ...` directly after `--` and runs from /tmp with TMPDIR=/tmp; counters are
# collected in passes of their own (--pmc with --kernel-trace only).  Nothing is re-measured silently: a run whose slowest
# launch of the top kernel took > 100x its fastest is KEPT and listed in <tag>_outliers.txt with that launch's duration.
set -e
TAG=${1:-r04}
PART=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
PY=$(readlink -f "$(command -v python3)")   # the real interpreter binary directly after `--` (no shim, no env hop)
cd /tmp && export TMPDIR=/tmp
stats() { # name, bench args...
  local name=$1; shift
  rm -rf $OUT/tmp_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$name -- $PY $R/bench.py --no-cpu-baseline "$@" > $OUT/$name.log 2>&1 || { echo "stats $name failed"; return 0; }
  local f=$(find $OUT/tmp_$name -name "*kernel_stats.csv" | head -n 1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats_$name.csv
  grep "^{\"metric\"" $OUT/$name.log | tail -n 1 > $OUT/${TAG}_bench_$name.json
  python3 - "$OUT/${TAG}_kernel_stats_$name.csv" "$name" >> $OUT/${TAG}_outliers.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = rows[0]
if int(top["Calls"]) > 8 and float(top["MaxNs"]) > 100 * float(top["MinNs"]):
    print("%s: %s -- one launch of %d took %.1f us (fastest %.1f us, average %.1f us): kept, see the stats file" % (
        sys.argv[2], top["Name"][:60], int(top["Calls"]), float(top["MaxNs"]) / 1e3, float(top["MinNs"]) / 1e3, float(top["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/tmp_$name
  echo "stats $name ok"
}
plain() { # name, bench args...: the bench line WITHOUT a profiler attached, over the one the profiled run printed -- for the handles whose level
  # generator runs on a stream of its own beside the steps: rocprofv3 stretches every event pair between the two streams (51.6 -> 74 us per step)
  local name=$1; shift
  timeout -k 10 300 $PY $R/bench.py --no-cpu-baseline "$@" > $OUT/$name.plain.log 2>&1 || { echo "plain $name failed"; return 0; }
  grep "^{\"metric\"" $OUT/$name.plain.log | tail -n 1 > $OUT/${TAG}_bench_$name.json
  echo "plain $name ok"
}
pmc() { # name, counter, bench args...: per-dispatch counter values of one pass
  local name=$1 c=$2; shift; shift
  rm -rf $OUT/tmp_pmc
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/tmp_pmc -- $PY $R/bench.py --steps 24 --warmup 8 --no-cpu-baseline "$@" > $OUT/pmc_${name}_$c.log 2>&1 || { echo "pmc $name $c failed"; return 0; }
  local f=$(find $OUT/tmp_pmc -name "*counter_collection.csv" | head -n 1)
  # keep the step kernels' rows only (the files are per dispatch: ~100 bytes each)
  [ -n "$f" ] && { head -n 1 $f > $OUT/${TAG}_pmc_${name}_$c.csv; grep -E "k_step|k_dynobs|k_levelgen" $f >> $OUT/${TAG}_pmc_${name}_$c.csv || true; }
  rm -rf $OUT/tmp_pmc
  echo "pmc $name $c ok"
}
: > $OUT/${TAG}_outliers.txt
if [ $PART = stats ] || [ $PART = all ]; then
  stats warmup --steps 64 --warmup 8      # a fresh box: first profiled process (kept like the rest)
  stats empty8x8_1M --config empty8 --steps 1024 --warmup 64
  stats empty8x8_1M_k20 --gpus 1 --steps 20 --warmup 5   # the driver's own command line (BENCH_rNN.json)
  stats doorkey8x8_1M --config doorkey8 --steps 1024 --warmup 64
  stats lavacrossing_512k --config lava4m --steps 1024 --warmup 64
  stats empty16x16_full_256k --config empty16full --steps 1024 --warmup 64
  stats empty8x8_4M --config empty8 --envs-per-gpu 4194304 --steps 256 --warmup 16
  stats lavacrossing_4M --config lava4m --envs-per-gpu 4194304 --steps 256 --warmup 16
  stats lavacrossing_1M --config lava4m --envs-per-gpu 1048576 --steps 512 --warmup 32
  stats lavacrossing_1M_newlevel --config lava4m --envs-per-gpu 1048576 --new-level-each-episode --steps 256
  plain lavacrossing_1M_newlevel --config lava4m --envs-per-gpu 1048576 --new-level-each-episode --steps 256
  stats lavacrossing_512k_newlevel --config lava4m --new-level-each-episode --steps 256
  plain lavacrossing_512k_newlevel --config lava4m --new-level-each-episode --steps 256
  stats doorkey8x8_1M_newlevel --config doorkey8 --new-level-each-episode --steps 1400 --warmup 64   # (episodes time out at 640 steps: two boundaries inside)
  plain doorkey8x8_1M_newlevel --config doorkey8 --new-level-each-episode --steps 1400 --warmup 64
  stats keycorridor_s3r3_256k_newlevel --env MiniGrid-KeyCorridorS3R3-v0 --envs-per-gpu 262144 --new-level-each-episode --steps 600 --warmup 30
  stats dynobs8x8_1M --env MiniGrid-Dynamic-Obstacles-8x8-v0 --steps 512 --warmup 64
  stats dynobs16x16_1M --env MiniGrid-Dynamic-Obstacles-16x16-v0 --steps 256 --warmup 32
  stats multiroom_n6_256k_newlevel --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 262144 --new-level-each-episode --steps 600 --warmup 30
  stats fourrooms_full_128k --env MiniGrid-FourRooms-v0 --envs-per-gpu 131072 --obs-mode full --steps 256
  stats fourrooms_full_512k --env MiniGrid-FourRooms-v0 --envs-per-gpu 524288 --obs-mode full --steps 256
  stats multiroom_n6_full_128k --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 131072 --obs-mode full --steps 256
  stats multiroom_n6_256k --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 262144 --steps 256
  stats fourrooms_1M --env MiniGrid-FourRooms-v0 --envs-per-gpu 1048576 --steps 256
  stats empty16x16_512k --env MiniGrid-Empty-16x16-v0 --envs-per-gpu 524288 --steps 256
  stats keycorridor_s6r3_512k --env MiniGrid-KeyCorridorS6R3-v0 --envs-per-gpu 524288 --steps 256
  stats obstructedmaze_2dlhb_256k --env MiniGrid-ObstructedMaze-2Dlhb-v0 --envs-per-gpu 262144 --steps 256
  stats empty8x8_1M_partial_onehot --obs-mode partial_onehot --steps 256
fi
if [ $PART = dyn ]; then   # only the Dynamic-Obstacles runs (re-taken after a change to k_dynobs or the k_step behind it)
  stats dynobs8x8_1M --env MiniGrid-Dynamic-Obstacles-8x8-v0 --steps 512 --warmup 64
  stats dynobs16x16_1M --env MiniGrid-Dynamic-Obstacles-16x16-v0 --steps 256 --warmup 32
fi
if [ $PART = newlevel ]; then   # only the runs in which k_levelgen works (re-taken after a change to it)
  stats lavacrossing_1M_newlevel --config lava4m --envs-per-gpu 1048576 --new-level-each-episode --steps 256
  plain lavacrossing_1M_newlevel --config lava4m --envs-per-gpu 1048576 --new-level-each-episode --steps 256
  stats lavacrossing_512k_newlevel --config lava4m --new-level-each-episode --steps 256
  plain lavacrossing_512k_newlevel --config lava4m --new-level-each-episode --steps 256
  stats doorkey8x8_1M_newlevel --config doorkey8 --new-level-each-episode --steps 1400 --warmup 64
  plain doorkey8x8_1M_newlevel --config doorkey8 --new-level-each-episode --steps 1400 --warmup 64
  stats multiroom_n6_256k_newlevel --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 262144 --new-level-each-episode --steps 600 --warmup 30
  stats keycorridor_s3r3_256k_newlevel --env MiniGrid-KeyCorridorS3R3-v0 --envs-per-gpu 262144 --new-level-each-episode --steps 600 --warmup 30
fi
if [ $PART = pmc ] || [ $PART = all ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    pmc empty8x8_1M $c --config empty8
    pmc doorkey8x8_1M $c --config doorkey8
    pmc lavacrossing_512k $c --config lava4m
    pmc empty16x16_full_256k $c --config empty16full
    pmc empty8x8_4M $c --config empty8 --envs-per-gpu 4194304
    pmc lavacrossing_4M $c --config lava4m --envs-per-gpu 4194304
    pmc lavacrossing_1M $c --config lava4m --envs-per-gpu 1048576
    pmc multiroom_n6_256k $c --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 262144     # the gather form k_step<0,0,3,7>
    pmc fourrooms_1M $c --env MiniGrid-FourRooms-v0 --envs-per-gpu 1048576
    pmc empty16x16_512k $c --env MiniGrid-Empty-16x16-v0 --envs-per-gpu 524288
    pmc dynobs8x8_1M $c --env MiniGrid-Dynamic-Obstacles-8x8-v0
  done
fi
if [ $PART = sq ] || [ $PART = all ]; then
  for w in "lava512k --config lava4m" "lava1m --config lava4m --envs-per-gpu 1048576" "empty8 --config empty8" "dyn1m --env MiniGrid-Dynamic-Obstacles-8x8-v0" \
           "empty16 --env MiniGrid-Empty-16x16-v0 --envs-per-gpu 524288" "keycorridor_s6r3 --env MiniGrid-KeyCorridorS6R3-v0 --envs-per-gpu 524288" \
           "obstructed_2dlhb --env MiniGrid-ObstructedMaze-2Dlhb-v0 --envs-per-gpu 262144" "fourrooms1m --env MiniGrid-FourRooms-v0 --envs-per-gpu 1048576" \
           "multiroom_n6 --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 262144"; do
    set -- $w; n=$1; shift
    (cd $R && tools/pmc_quick2.sh ${TAG}_$n "$@" > /dev/null && python3 tools/pmc_summary.py ${TAG}_$n k_ > $OUT/${TAG}_sq_counters_$n.txt) || echo "sq $n failed"
    echo "sq $n ok"
  done
fi
cd $R && timeout -k 10 400 python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err || echo "default bench failed"
echo "done $TAG $PART"
