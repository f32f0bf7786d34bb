#!/bin/bash
# Usage (GPU box, repo root; needs ab/tuning.so = tools/build_variant.sh tuning -DMGX_TUNING): tools/lg_layout_sweep.sh [env] [n]
# k_levelgen's block shape (generating lanes per fast wave, envs per block) under the ring form: us per step of a new_level_each_episode handle.
R=${GRAFT_REPO_ROOT:-$(pwd)}
ENV=${1:-MiniGrid-LavaCrossingS9N1-v0}; N=${2:-1048576}
us() { python3 -c "import json,sys; print('%.1f' % (1000 * json.loads(sys.stdin.readlines()[-1])['ms_per_step']))"; }
echo "# $ENV N=$N, us per step (ring default); rows: lanes per fast wave, columns: envs per block"
for lanes in default 16 24 32 48 64; do
  line="lanes=$lanes:"
  for span in default 256 512 1024 2048; do
    L=""; S=""
    [ $lanes != default ] && L="MGX_LG_LANES=$lanes"
    [ $span != default ] && S="MGX_LG_SPAN=$span"
    line="$line span=$span $(env MGX_LIB=$R/ab/tuning.so $L $S timeout -k 10 120 python3 $R/bench.py --config lava4m --env $ENV --envs-per-gpu $N --new-level-each-episode --steps 400 --warmup 48 --no-cpu-baseline 2>/dev/null | us) |"
  done
  echo "$line"
done
