#!/bin/bash
# the level image for probes on larger grids: same-box A/B of MGX_LGF_MAXS builds (new_level_each_episode, us per step)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { MGX_LIB=$R/ab/$3.so timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu ${2:-262144} --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-36s n=%-8d %-8s %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], '$3', j['value'], j['ms_per_step']*1e3))"; }
for e in "MiniGrid-ObstructedMaze-2Dlhb-v0 262144" "MiniGrid-ObstructedMaze-1Dlhb-v0 262144" "MiniGrid-Playground-v0 262144" "MiniGrid-LockedRoom-v0 262144" "MiniGrid-KeyCorridorS5R3-v0 262144" "MiniGrid-DoorKey-16x16-v0 262144" "MiniGrid-FourRooms-v0 262144" "MiniGrid-MemoryS13Random-v0 262144" "MiniGrid-RedBlueDoors-8x8-v0 262144" "MiniGrid-MultiRoom-N6-v0 262144"; do
  for v in "$@"; do b $e $v; done
done 2>&1 | tee $O/stream14.txt
