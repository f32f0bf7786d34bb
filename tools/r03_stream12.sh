#!/bin/bash
# KeyCorridor after connect_all's search became a component count: new level per episode, 262,144 envs, and the per-launch trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu ${2:-262144} --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json,os; j=json.loads(sys.stdin.read()); print('%-34s n=%-8d %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], j['value'], j['ms_per_step']*1e3))"; }
{ b MiniGrid-KeyCorridorS3R3-v0; b MiniGrid-KeyCorridorS3R2-v0; b MiniGrid-KeyCorridorS4R3-v0; b MiniGrid-MultiRoom-N6-v0; b MiniGrid-Fetch-8x8-N3-v0; } 2>&1 | tee $O/stream12.txt
