#!/bin/bash
# Same-box A/B of a FORM-SELECTION knob (read in every build: MGX_PARTIAL_KERNEL staged|gather, MGX_FULL_KERNEL lds|direct, MGX_GATHER_CACHE on|off,
# MGX_DYNOBS fused|split, MGX_ONEHOT split|fused, MGX_SEED_FORM window|full, MGX_ROLLOUT graph): bench.py per workload under each value.
#   [BENCH_EXTRA="--obs-mode full"] tools/form_ab.sh <tag> <KNOB> <value,value,...> env[:n_envs] ...      (value `rule` = knob unset)
# (one script for what round 3 kept as tools/archive_r03/r03_ab_16.sh, r03_ab_small.sh, r03_ab_gather*.sh, r03_ab_full.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=$1; KNOB=$2; VALUES=${3//,/ }; shift 3
EXTRA=${BENCH_EXTRA:-}   # extra bench.py arguments for every run, e.g. BENCH_EXTRA="--obs-mode full"
b() { env $1 python bench.py --no-cpu-baseline --steps 256 --warmup 32 $EXTRA --env $2 --envs-per-gpu $3 2>/dev/null |
  python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-28s %-40s n=%-8d %-28s %8.2f us  %.2f G' % ('$1', '$2', $3, r['kernel'], r['span_us_per_step'], d['value']/1e9))"; }
for w in "$@"; do
  e=${w%%:*}; n=${w#*:}; [ "$n" = "$w" ] && n=1048576
  for v in $VALUES; do if [ "$v" = rule ]; then b "MGX_NOOP=1" $e $n; else b "$KNOB=$v" $e $n; fi; done
done 2>&1 | tee $O/form_$TAG.txt
