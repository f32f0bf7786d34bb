#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_quick.sh <tag> <bench.py args...>  -- the three SQ counter groups only
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" ; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python $R/bench.py --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
echo "done $TAG"
