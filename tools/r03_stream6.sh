#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu ${2:-262144} --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-34s n=%-8d lanes=%-4s fast_waves=%-4s span=%-5s %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], os.environ.get('MGX_LG_LANES','rule'), os.environ.get('MGX_LG_FAST_WAVES','rule'), os.environ.get('MGX_LG_SPAN','rule'), j['value'], j['ms_per_step']*1e3))"; }
{
b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=12 b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=20 b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=24 b MiniGrid-MultiRoom-N6-v0
MGX_LG_SPAN=128 b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=24 MGX_LG_SPAN=128 b MiniGrid-MultiRoom-N6-v0
b MiniGrid-KeyCorridorS3R3-v0
MGX_LG_LANES=48 b MiniGrid-KeyCorridorS3R3-v0
MGX_LG_LANES=32 MGX_LG_SPAN=256 b MiniGrid-KeyCorridorS3R3-v0
MGX_LG_LANES=64 MGX_LG_SPAN=256 b MiniGrid-KeyCorridorS3R3-v0
b MiniGrid-MultiRoom-N6-v0 1048576
b MiniGrid-LavaCrossingS9N1-v0 1048576
b MiniGrid-Fetch-8x8-N3-v0
} 2>&1 | tee $O/stream6.txt
