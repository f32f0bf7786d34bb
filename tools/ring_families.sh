#!/bin/bash
# Usage (GPU box, repo root): tools/ring_families.sh > gpurun_out/ring_families.txt
# new_level_each_episode, us per step: replay | one next-level buffer, k_levelgen behind every step (MGX_LG_RING=off) | ring of 16, generator beside the steps,
# a launch per step's flags (16) or one launch per run of four steps (16m); default = what mgx_create's family rules pick
R=${GRAFT_REPO_ROOT:-$(pwd)}
us() { python3 -c "import json,sys; print('%.1f' % (1000 * json.loads(sys.stdin.readlines()[-1])['ms_per_step']))"; }
run() { # env n [extra bench args]
  local e=$1 n=$2; shift; shift
  local line="$e N=$n $*:"
  line="$line replay $(timeout -k 10 120 python3 $R/bench.py --config lava4m --env $e --envs-per-gpu $n --steps 600 --warmup 64 --no-cpu-baseline "$@" 2>/dev/null | us)"
  for f in off 16 16m; do
    line="$line | ring=$f $(MGX_LG_MERGE=$([ $f = 16m ] && echo 1 || echo 0) MGX_LG_RING=${f%m} timeout -k 10 120 python3 $R/bench.py --config lava4m --env $e --envs-per-gpu $n --new-level-each-episode --steps 600 --warmup 64 --no-cpu-baseline "$@" 2>/dev/null | us)"
  done
  line="$line | default $(timeout -k 10 120 python3 $R/bench.py --config lava4m --env $e --envs-per-gpu $n --new-level-each-episode --steps 600 --warmup 64 --no-cpu-baseline "$@" 2>/dev/null | us)"
  echo "$line"
}
run MiniGrid-Fetch-8x8-N3-v0 1048576
run MiniGrid-PutNear-6x6-N2-v0 1048576
run MiniGrid-GoToObject-6x6-N2-v0 1048576
run MiniGrid-GoToDoor-8x8-v0 1048576
run MiniGrid-Unlock-v0 1048576
run MiniGrid-RedBlueDoors-8x8-v0 1048576
run MiniGrid-DistShift1-v0 1048576
run MiniGrid-MemoryS13Random-v0 524288
run MiniGrid-FourRooms-v0 524288
run MiniGrid-LockedRoom-v0 262144
run MiniGrid-Playground-v0 262144
run MiniGrid-ObstructedMaze-1Dlhb-v0 262144
run MiniGrid-KeyCorridorS3R3-v0 262144
run MiniGrid-MultiRoom-N6-v0 262144
run MiniGrid-MultiRoom-N2-S4-v0 1048576
run MiniGrid-DoorKey-16x16-v0 524288
run MiniGrid-LavaCrossingS9N1-v0 1048576
run MiniGrid-LavaCrossingS9N1-v0 524288
run MiniGrid-DoorKey-8x8-v0 1048576
run MiniGrid-LavaCrossingS9N1-v0 1048576 --view 5
run MiniGrid-LavaCrossingS9N1-v0 1048576 --obs-mode partial_onehot
