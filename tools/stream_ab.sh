#!/bin/bash
# new_level_each_episode, us per step and env-steps/s, for a list of workloads -- optionally across builds (ab/<name>.so from
# tools/build_variant.sh / build_lg_variant.sh) and across settings of the level generator's tuning knobs (tuning builds only).
#   tools/stream_ab.sh <tag> [-l name,name,...] [-k "MGX_LG_LANES=16 MGX_LG_LANES=32 ..."] env[:n_envs] ...
# (one script for what round 3 kept as tools/archive_r03/r03_stream7.sh ... r03_stream20.sh; their tables are in profiles/r03_levelgen_paths.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=$1; shift
LIBS="-"; KNOBS="-"
while getopts "l:k:" o; do case $o in l) LIBS=${OPTARG//,/ };; k) KNOBS=$OPTARG;; esac; done
shift $((OPTIND - 1))
b() { # env n lib knob
  local lib=""; [ "$3" != "-" ] && lib="MGX_LIB=$R/ab/$3.so"
  local knob=""; [ "$4" != "-" ] && knob="$4"
  env $lib $knob timeout -k 10 200 python bench.py --no-cpu-baseline --env $1 --envs-per-gpu $2 --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-36s n=%-8d %-10s %-22s %.3g steps/s %.1f us/step' % (j['config']['env_id'], j['config']['envs_per_gpu'], '$3', '$4', j['value'], j['ms_per_step']*1e3))"
}
for w in "$@"; do
  e=${w%%:*}; n=${w#*:}; [ "$n" = "$w" ] && n=262144
  for lib in $LIBS; do for knob in $KNOBS; do b $e $n $lib $knob; done; done
done 2>&1 | tee $O/stream_$TAG.txt
