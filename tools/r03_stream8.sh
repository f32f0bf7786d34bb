#!/bin/bash
# MultiRoom-N6 / KeyCorridor knob sweeps after the one-round-trip refill
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu ${2:-262144} --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-34s n=%-8d lanes=%-4s fast_waves=%-4s span=%-5s %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], os.environ.get('MGX_LG_LANES','rule'), os.environ.get('MGX_LG_FAST_WAVES','rule'), os.environ.get('MGX_LG_SPAN','rule'), j['value'], j['ms_per_step']*1e3))"; }
{
b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=12 b MiniGrid-MultiRoom-N6-v0
MGX_LG_LANES=20 b MiniGrid-MultiRoom-N6-v0
MGX_LG_SPAN=64 b MiniGrid-MultiRoom-N6-v0
MGX_LG_SPAN=256 b MiniGrid-MultiRoom-N6-v0
MGX_LG_FAST_WAVES=2 b MiniGrid-MultiRoom-N6-v0
MGX_LG_FAST_WAVES=3 b MiniGrid-MultiRoom-N6-v0
MGX_LG_FAST_WAVES=2 MGX_LG_SPAN=64 b MiniGrid-MultiRoom-N6-v0
MGX_LG_SPAN=64 b MiniGrid-KeyCorridorS3R3-v0
MGX_LG_SPAN=256 b MiniGrid-KeyCorridorS3R3-v0
MGX_LG_LANES=16 MGX_LG_SPAN=64 b MiniGrid-KeyCorridorS3R3-v0
MGX_LG_LANES=24 b MiniGrid-KeyCorridorS3R3-v0
} 2>&1 | tee $O/stream8.txt
