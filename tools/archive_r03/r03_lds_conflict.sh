#!/bin/bash
# where do the LDS bank conflicts of k_step<8,8,0,7> come from, and what are they worth?
# product build vs ab/nogc.so (-DMGX_EXP_FIXED_GATHER: the 49 view reads at a lane-independent offset = conflict-free; wrong observations)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/lds_conflict; mkdir -p $O; cd $R
PY=$(readlink -f "$(command -v python3)")
for v in product nogc; do
  [ $v = nogc ] && export MGX_LIB=$R/ab/nogc.so || unset MGX_LIB
  for r in 1 2 3; do python bench.py --no-cpu-baseline --steps 1024 --warmup 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', 'span us/step', round(d['roofline']['span_us_per_step'],2))"; done
  (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/t_$v -- $PY $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline > $O/$v.log 2>&1)
  f=$(find $O/t_$v -name "*counter_collection.csv" | head -n 1)
  python - "$f" $v <<'PY'
import csv, sys, collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_step" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sorted(v)[len(v)//2] for k,v in d.items()}
w=m["SQ_WAVES"]
print(sys.argv[2], " ".join("%s=%.1f/wave" % (k, m[k]/w) for k in sorted(m) if k!="SQ_WAVES"), "conflict share %.2f" % (m["SQ_LDS_BANK_CONFLICT"]/m["SQ_LDS_IDX_ACTIVE"]))
PY
  rm -rf $O/t_$v
done 2>&1 | tee $O/summary.txt
