#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --steps 256 --warmup 32 "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-66s %-32s %8.2f us  %.3f  %.2f G' % (' '.join(sys.argv[1:]), r['kernel'], r['span_us_per_step'], r['frac'], d['value']/1e9))" "$@"; }
{
b --config empty16full
b --config empty16full --envs-per-gpu 1048576
b --env MiniGrid-FourRooms-v0 --envs-per-gpu 131072 --obs-mode full
b --env MiniGrid-FourRooms-v0 --envs-per-gpu 524288 --obs-mode full
b --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 131072 --obs-mode full
b --env MiniGrid-DoorKey-8x8-v0 --obs-mode full
b --env MiniGrid-LavaCrossingS9N1-v0 --obs-mode full
b --env MiniGrid-Dynamic-Obstacles-8x8-v0
b --env MiniGrid-LavaCrossingS9N1-v0 --new-level-each-episode
python tools/rollout_bench.py MiniGrid-DoorKey-8x8-v0
python tools/rollout_bench.py MiniGrid-LavaCrossingS9N1-v0
} 2>&1 | tee $O/ab_full.txt
