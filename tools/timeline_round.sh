#!/bin/bash
# Wave timelines of the step kernel (instrumented build ab/tl.so: tools/build_variant.sh tl -DMGX_TIMELINE=1) for LavaCrossing at
# 524,288 and 1 Mi envs, with the launch-shaping rules off / tail priority only / tail priority + first-round stagger.
set -e
mkdir -p gpurun_out
export MGX_LIB=$PWD/ab/tl.so   # the instrumented build is selected by path (gym_minigrid_amd/_lib.py); csrc/libmgx.so stays the product build
for n in 524288 1048576; do
  for mode in "0 0 plain" "512 0 tail" "512 5 tail_stagger"; do
    set -- $mode
    MGX_TAIL_BLOCKS=$1 MGX_STAGGER=$2 MGX_TL_FILE=/tmp/tl.bin python bench.py --config lava4m --envs-per-gpu $n --steps 400 --warmup 50 --no-cpu-baseline > /dev/null 2>&1
    python tools/timeline_report.py /tmp/tl.bin > gpurun_out/timeline_lava_${n}_$3.txt
  done
done
