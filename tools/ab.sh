#!/bin/bash
# Same-box A/B of builds of libmgx.so (ab/<name>.so from tools/build_variant.sh): alternate them under bench.py and print the
# step kernel's span per launch.  Usage (on the GPU box): tools/ab.sh "<bench args>" <rounds> <name> <name> ...
# The variant is selected with MGX_LIB (gym_minigrid_amd/_lib.py): the product build in csrc/ is never overwritten.
args=${1:---config lava4m}
rounds=${2:-3}
shift 2
mkdir -p gpurun_out
for r in $(seq $rounds); do
  for v in "$@"; do
    MGX_LIB=$PWD/ab/$v.so python bench.py $args --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-12s' % '$v', round(d['roofline']['span_us_per_step'],2), round(d['roofline']['frac'],3))"
  done
done | tee -a gpurun_out/ab.log
