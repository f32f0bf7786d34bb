#!/bin/bash
# MultiRoom's room search as a per-lane state machine with lanes taking the next level when done: same-box A/B of builds (new_level_each_episode, us per step)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { MGX_LIB=$R/ab/$3.so timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu ${2:-262144} --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-34s n=%-8d %-8s lanes=%-4s %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], '$3', os.environ.get('MGX_LG_LANES','rule'), j['value'], j['ms_per_step']*1e3))"; }
for e in "MiniGrid-MultiRoom-N6-v0 262144" "MiniGrid-MultiRoom-N4-S5-v0 262144" "MiniGrid-MultiRoom-N2-S4-v0 262144" "MiniGrid-MultiRoom-N6-v0 1048576" "MiniGrid-Fetch-8x8-N3-v0 262144" "MiniGrid-LavaCrossingS9N1-v0 1048576"; do
  for v in "$@"; do b $e $v; done
done 2>&1 | tee $O/stream20.txt
for l in 24 32 64; do MGX_LG_LANES=$l b MiniGrid-MultiRoom-N6-v0 262144 mrsm; done 2>&1 | tee -a $O/stream20.txt
