#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu 262144 --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-34s fast_waves=%-4s mr_fast=%-2s %.3g steps/s %.1f us/step' % (j['config']['env_id'], os.environ.get('MGX_LG_FAST_WAVES','rule'), os.environ.get('MGX_LG_MULTIROOM_FAST','-'), j['value'], j['ms_per_step']*1e3))"; }
{
b MiniGrid-MultiRoom-N6-v0
MGX_LG_MULTIROOM_FAST=1 b MiniGrid-MultiRoom-N6-v0
MGX_LG_MULTIROOM_FAST=1 MGX_LG_FAST_WAVES=2 b MiniGrid-MultiRoom-N6-v0
MGX_LG_MULTIROOM_FAST=1 MGX_LG_FAST_WAVES=4 b MiniGrid-MultiRoom-N6-v0
for e in MiniGrid-KeyCorridorS3R3-v0 MiniGrid-LockedRoom-v0 MiniGrid-ObstructedMaze-2Dlhb-v0 MiniGrid-Fetch-8x8-N3-v0 MiniGrid-MultiRoom-N4-S5-v0; do
  b $e
  MGX_LG_FAST_WAVES=0 b $e
  MGX_LG_FAST_WAVES=2 b $e
  MGX_LG_FAST_WAVES=4 b $e
done
} 2>&1 | tee $O/stream2.txt
