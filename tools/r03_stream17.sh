#!/bin/bash
# same-box A/B of builds on the ids that pay most for a new level per episode (tiny task grids), us per step:  tools/r03_stream17.sh <build> <build> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { MGX_LIB=$R/ab/$3.so timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu ${2:-262144} --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-34s n=%-8d %-8s %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], '$3', j['value'], j['ms_per_step']*1e3))"; }
for e in "MiniGrid-GoToObject-6x6-N2-v0 262144" "MiniGrid-GoToDoor-5x5-v0 262144" "MiniGrid-KeyCorridorS3R1-v0 262144" "MiniGrid-PutNear-6x6-N2-v0 262144" "MiniGrid-Fetch-5x5-N2-v0 262144" "MiniGrid-KeyCorridorS3R3-v0 262144" "MiniGrid-TwoGoals-Random-5x5-v0 262144" "MiniGrid-MultiRoom-N6-v0 262144" "MiniGrid-LavaCrossingS9N1-v0 1048576"; do
  for v in "$@"; do b $e $v; done
done 2>&1 | tee $O/stream17.txt
