#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { python bench.py --no-cpu-baseline --steps 256 --warmup 32 "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-8s %-46s %-22s %8.2f us  %.3f  %.2f G' % (os.environ.get('MGX_LIB','product')[-6:], ' '.join(sys.argv[1:]), r['kernel'], r['span_us_per_step'], r['frac'], d['value']/1e9))" "$@"; }
{
for lib in "" ab/g1.so ab/g5.so ab/g6.so; do
  [ -n "$lib" ] && export MGX_LIB=$R/$lib || unset MGX_LIB
  b --env MiniGrid-FourRooms-v0 --envs-per-gpu 1048576
  b --env MiniGrid-FourRooms-v0 --envs-per-gpu 262144
  b --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 262144
  MGX_PARTIAL_KERNEL=gather b --env MiniGrid-Empty-16x16-v0 --envs-per-gpu 524288
  MGX_PARTIAL_KERNEL=gather b --env MiniGrid-KeyCorridorS6R3-v0 --envs-per-gpu 524288
done
unset MGX_LIB
b --env MiniGrid-LockedRoom-v0 --envs-per-gpu 262144
b --env MiniGrid-MemoryS17Random-v0 --envs-per-gpu 262144
b --env MiniGrid-MultiRoom-N6-v0 --envs-per-gpu 1048576
} 2>&1 | tee $O/ab_gather2.txt
