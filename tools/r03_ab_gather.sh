#!/bin/bash
# gather form after the front-cell cache; gather vs staged on the 16x16 instances
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --steps 256 --warmup 32 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-46s %-28s %8.2f us  %.3f  %.2f G' % (' '.join(sys.argv[1:]), r['kernel'], r['span_us_per_step'], r['frac'], d['value']/1e9))" "$@"; }
{
b --env MiniGrid-FourRooms-v0 --envs-per-gpu 1048576
b --env MiniGrid-FourRooms-v0 --envs-per-gpu 262144
b --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 262144
b --env MiniGrid-LockedRoom-v0 --envs-per-gpu 262144
b --env MiniGrid-MemoryS17Random-v0 --envs-per-gpu 262144
for e in MiniGrid-Empty-16x16-v0 MiniGrid-KeyCorridorS6R3-v0 MiniGrid-DoorKey-16x16-v0; do
  b --env $e --envs-per-gpu 524288
  MGX_PARTIAL_KERNEL=gather b --env $e --envs-per-gpu 524288
done
} 2>&1 | tee $O/ab_gather.txt
