#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu ${2:-262144} --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-34s n=%-8d lanes=%-4s fast_waves=%-4s span=%-5s %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], os.environ.get('MGX_LG_LANES','rule'), os.environ.get('MGX_LG_FAST_WAVES','rule'), os.environ.get('MGX_LG_SPAN','rule'), j['value'], j['ms_per_step']*1e3))"; }
{
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "stream or levelgen or seeded_reset or new_level or task_families_on_device or obstructed" 2>&1 | tail -n 3
b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=32 b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=64 MGX_LG_SPAN=128 b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=8 b MiniGrid-MultiRoom-N6-v0
b MiniGrid-KeyCorridorS3R3-v0
MGX_LG_FAST_WAVES=4 MGX_LG_LANES=16 b MiniGrid-KeyCorridorS3R3-v0
MGX_LG_FAST_WAVES=4 MGX_LG_LANES=32 b MiniGrid-KeyCorridorS3R3-v0
MGX_LG_FAST_WAVES=1 MGX_LG_LANES=64 b MiniGrid-KeyCorridorS3R3-v0
b MiniGrid-MultiRoom-N4-S5-v0
for e in MiniGrid-LockedRoom-v0 MiniGrid-Fetch-8x8-N3-v0 MiniGrid-ObstructedMaze-2Dlhb-v0 MiniGrid-Playground-v0 MiniGrid-GoToObject-8x8-N2-v0 MiniGrid-PutNear-8x8-N3-v0; do b $e; done
b MiniGrid-LavaCrossingS9N1-v0 1048576
b MiniGrid-DoorKey-8x8-v0 1048576
} 2>&1 | tee $O/stream5.txt
