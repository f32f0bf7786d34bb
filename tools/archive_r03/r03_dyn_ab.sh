#!/bin/bash
# Same-box A/B of libmgx.so variants on Dynamic-Obstacles (k_dynobs + k_step per step): us per step, whole step.
#   tools/archive_r03/r03_dyn_ab.sh <rounds> <name> <name> ...   (variants from tools/build_variant.sh; MGX_EXP_DYN bits are timing-only)
rounds=${1:-2}; shift
mkdir -p gpurun_out
for env in MiniGrid-Dynamic-Obstacles-8x8-v0 MiniGrid-Dynamic-Obstacles-16x16-v0; do
for r in $(seq $rounds); do
  for v in "$@"; do
    MGX_LIB=$PWD/ab/$v.so python bench.py --env $env --envs-per-gpu ${DYN_ENVS:-1048576} --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-36s %-10s' % ('$env', '$v'), round(d['ms_per_step']*1e3,2), 'us/step')"
  done
done
done | tee -a gpurun_out/dyn_ab.log
