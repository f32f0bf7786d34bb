#!/bin/bash
# Same-box A/B of one environment variable's settings under bench.py (tuning aid).
#   tools/ab_env.sh MGX_TAIL_BLOCKS "0 512" 2 "--config lava4m"
var=$1; vals=$2; rounds=${3:-2}; args=${4:---config empty8}
mkdir -p gpurun_out
for r in $(seq $rounds); do
  for v in $vals; do
    env $var=$v python bench.py $args --steps 600 --warmup 100 --no-cpu-baseline 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-28s %-8s' % ('$args'[:28], '$var=$v'[-8:]), round(d['roofline']['span_us_per_step'],2), round(d['roofline']['frac'],3))"
  done
done | tee -a gpurun_out/ab_env.log
