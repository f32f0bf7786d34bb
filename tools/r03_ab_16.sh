#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --steps 256 --warmup 32 "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-7s %-56s %-24s %8.2f us  %.2f G' % (os.environ.get('MGX_PARTIAL_KERNEL','rule'), ' '.join(sys.argv[1:]), r['kernel'], r['span_us_per_step'], d['value']/1e9))" "$@"; }
{
for e in "MiniGrid-Empty-16x16-v0 524288" "MiniGrid-KeyCorridorS6R3-v0 524288" "MiniGrid-ObstructedMaze-2Dlhb-v0 262144" "MiniGrid-ObstructedMaze-Full-v0 262144" "MiniGrid-MemoryS13Random-v0 524288" "MiniGrid-FourRooms-v0 1048576" "MiniGrid-TwoGoals-Random-16x16-v0 524288"; do
  set -- $e
  b --env $1 --envs-per-gpu $2
  MGX_PARTIAL_KERNEL=staged b --env $1 --envs-per-gpu $2
done
python tools/rollout_bench.py MiniGrid-DoorKey-16x16-v0
} 2>&1 | tee $O/ab_16.txt
