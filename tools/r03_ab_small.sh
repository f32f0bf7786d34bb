#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --steps 512 --warmup 64 "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-7s %-56s %-24s %8.2f us  %.2f G' % (os.environ.get('MGX_PARTIAL_KERNEL','rule'), ' '.join(sys.argv[1:]), r['kernel'], r['span_us_per_step'], d['value']/1e9))" "$@"; }
{
for r in 1 2; do
for e in "MiniGrid-Empty-8x8-v0 1048576" "MiniGrid-DoorKey-8x8-v0 1048576" "MiniGrid-LavaCrossingS9N1-v0 524288" "MiniGrid-LavaCrossingS9N1-v0 1048576" "MiniGrid-SimpleCrossingS11N5-v0 1048576"; do
  set -- $e
  b --env $1 --envs-per-gpu $2
  MGX_PARTIAL_KERNEL=gather b --env $1 --envs-per-gpu $2
done; done
} 2>&1 | tee $O/ab_small.txt
