#!/bin/bash
# reset-heavy tiny grids (tens of thousands of levels per step): generating lanes per wave (new_level_each_episode, us per step)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu ${2:-262144} --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-34s n=%-8d lanes=%-4s span=%-5s %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], os.environ.get('MGX_LG_LANES','rule'), os.environ.get('MGX_LG_SPAN','rule'), j['value'], j['ms_per_step']*1e3))"; }
for e in "MiniGrid-GoToObject-6x6-N2-v0 262144" "MiniGrid-GoToDoor-5x5-v0 262144" "MiniGrid-TwoGoals-Random-5x5-v0 262144" "MiniGrid-PutNear-6x6-N2-v0 262144" "MiniGrid-Fetch-5x5-N2-v0 262144" "MiniGrid-GoToObject-6x6-N2-v0 1048576"; do
  b $e
  MGX_LG_LANES=32 b $e
  MGX_LG_LANES=16 b $e
  MGX_LG_LANES=32 MGX_LG_SPAN=256 b $e
done 2>&1 | tee $O/stream18.txt
