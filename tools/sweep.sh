#!/bin/bash
# Usage (GPU box, repo root): tools/sweep.sh  -- one bench line per family/option into gpurun_out/sweep.jsonl
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sweep.jsonl
: > $OUT
run() { timeout -k 10 240 python $R/bench.py --no-cpu-baseline --steps 256 --warmup 32 "$@" 2>/dev/null | tail -n 1 >> $OUT || echo "{\"failed\": \"$*\"}" >> $OUT; }
run --env MiniGrid-Empty-5x5-v0
run --env MiniGrid-Empty-6x6-v0
run --env MiniGrid-Empty-8x8-v0
run --env MiniGrid-Empty-16x16-v0 --envs-per-gpu 524288
run --env MiniGrid-DoorKey-5x5-v0
run --env MiniGrid-DoorKey-8x8-v0
run --env MiniGrid-DoorKey-16x16-v0 --envs-per-gpu 524288
run --env MiniGrid-LavaCrossingS9N1-v0
run --env MiniGrid-SimpleCrossingS11N5-v0
run --env MiniGrid-LavaGapS7-v0
run --env MiniGrid-DistShift1-v0
run --env MiniGrid-FourRooms-v0 --envs-per-gpu 262144
run --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 262144
run --env MiniGrid-Fetch-8x8-N3-v0
run --env MiniGrid-GoToDoor-8x8-v0
run --env MiniGrid-Dynamic-Obstacles-8x8-v0
run --env MiniGrid-Empty-8x8-v0 --view 3
run --env MiniGrid-Empty-8x8-v0 --view 5
run --env MiniGrid-DoorKey-8x8-v0 --view 9
run --env MiniGrid-DoorKey-8x8-v0 --view 11
run --env MiniGrid-Empty-8x8-v0 --obs-mode full
run --env MiniGrid-Empty-16x16-v0 --obs-mode full --envs-per-gpu 262144
run --env MiniGrid-FourRooms-v0 --obs-mode full --envs-per-gpu 131072
run --env MiniGrid-LavaCrossingS9N1-v0 --new-level-each-episode
run --env MiniGrid-DoorKey-8x8-v0 --new-level-each-episode
echo done
