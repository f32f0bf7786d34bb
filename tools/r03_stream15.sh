#!/bin/bash
# steady-flow families: block span sweep after the round-robin dealing and the level image (new_level_each_episode, us per step)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu ${2:-262144} --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-34s n=%-8d span=%-5s fast_waves=%-4s %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], os.environ.get('MGX_LG_SPAN','rule'), os.environ.get('MGX_LG_FAST_WAVES','rule'), j['value'], j['ms_per_step']*1e3))"; }
for e in "MiniGrid-PutNear-8x8-N3-v0 262144" "MiniGrid-Fetch-8x8-N3-v0 262144" "MiniGrid-GoToObject-8x8-N2-v0 262144" "MiniGrid-LavaCrossingS9N1-v0 1048576" "MiniGrid-ObstructedMaze-2Dlhb-v0 262144"; do
  b $e
  for s in 128 256 1024; do MGX_LG_SPAN=$s b $e; done
done 2>&1 | tee $O/stream15.txt
