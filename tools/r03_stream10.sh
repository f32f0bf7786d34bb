#!/bin/bash
# KeyCorridor lanes-per-wave sweep on one box (new_level_each_episode, 262,144 envs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu ${2:-262144} --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-34s n=%-8d lanes=%-4s fast_waves=%-4s span=%-5s %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], os.environ.get('MGX_LG_LANES','rule'), os.environ.get('MGX_LG_FAST_WAVES','rule'), os.environ.get('MGX_LG_SPAN','rule'), j['value'], j['ms_per_step']*1e3))"; }
{
for e in MiniGrid-KeyCorridorS3R3-v0 MiniGrid-KeyCorridorS6R3-v0; do
b $e
for l in 8 12 16 20 24; do MGX_LG_LANES=$l b $e; done
MGX_LG_LANES=16 MGX_LG_SPAN=64 b $e
done
} 2>&1 | tee $O/stream10.txt
