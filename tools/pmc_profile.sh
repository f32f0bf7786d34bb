#!/bin/bash
# Usage (on the GPU box, from the repo root):  tools/pmc_profile.sh <tag> <bench.py args...>
# Runs the bench under rocprofv3 once per counter group (SQ: 8 slots/pass; TCC: few slots/pass), each in its own
# pass with --kernel-trace only (never combined with other trace domains), writing CSVs to gpurun_out/pmc_<tag>/.
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_LEVEL_WAVES" \
 "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" \
 "TCC_TAG_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_BUSY" \
 "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_WRITE_TAGCONFLICT_STALL_CYCLES" \
 "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
echo "done $TAG"
