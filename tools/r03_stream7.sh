#!/bin/bash
# new_level_each_episode after the one-round-trip window refill: every family of r03_levelgen_paths.txt at the rule, + lanes sweeps
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu ${2:-262144} --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-34s n=%-8d lanes=%-4s fast_waves=%-4s span=%-5s %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], os.environ.get('MGX_LG_LANES','rule'), os.environ.get('MGX_LG_FAST_WAVES','rule'), os.environ.get('MGX_LG_SPAN','rule'), j['value'], j['ms_per_step']*1e3))"; }
{
b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=24 b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=32 b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=64 b MiniGrid-MultiRoom-N6-v0
b MiniGrid-KeyCorridorS3R3-v0
MGX_LG_LANES=64 b MiniGrid-KeyCorridorS3R3-v0
MGX_LG_LANES=16 b MiniGrid-KeyCorridorS3R3-v0
b MiniGrid-MultiRoom-N4-S5-v0
b MiniGrid-Playground-v0
b MiniGrid-GoToObject-8x8-N2-v0
b MiniGrid-PutNear-8x8-N3-v0
b MiniGrid-Fetch-8x8-N3-v0
b MiniGrid-LockedRoom-v0
b MiniGrid-ObstructedMaze-2Dlhb-v0
b MiniGrid-MultiRoom-N6-v0 1048576
b MiniGrid-LavaCrossingS9N1-v0 1048576
b MiniGrid-DoorKey-8x8-v0 1048576
} 2>&1 | tee $O/stream7.txt
