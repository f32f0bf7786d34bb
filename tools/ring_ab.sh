#!/bin/bash
# Usage (GPU box, repo root): tools/ring_ab.sh > gpurun_out/ring_ab.txt
# new_level_each_episode handles: us per step (bench.py, no profiler attached: a profiler stretches the event pairs between the two streams) with one
# next-level buffer and k_levelgen behind every step (MGX_LG_RING=off) against rings of 4 / 8 / 16 buffers with the generator beside the steps,
# the replay step beside them; then the kernel timeline of a few steps of the default form (rocprofv3 --kernel-trace).
R=${GRAFT_REPO_ROOT:-$(pwd)}
us() { python3 -c "import json,sys; print('%.1f' % (1000 * json.loads(sys.stdin.readlines()[-1])['ms_per_step']))"; }
run() { # env n
  local line="$1 N=$2:"
  line="$line replay $(timeout -k 10 120 python3 $R/bench.py --config lava4m --env $1 --envs-per-gpu $2 --steps 600 --warmup 64 --no-cpu-baseline 2>/dev/null | us)"
  for f in off 4 8 16; do
    line="$line | ring=$f $(MGX_LG_RING=$f timeout -k 10 120 python3 $R/bench.py --config lava4m --env $1 --envs-per-gpu $2 --new-level-each-episode --steps 600 --warmup 64 --no-cpu-baseline 2>/dev/null | us)"
  done
  echo "$line"
}
echo "# us per step, new level per episode (replay = the same level every episode)"
run MiniGrid-LavaCrossingS9N1-v0 1048576
run MiniGrid-LavaCrossingS9N1-v0 524288
run MiniGrid-LavaCrossingS9N1-v0 65536
run MiniGrid-LavaCrossingS9N1-v0 16384
run MiniGrid-LavaCrossingS9N3-v0 1048576
run MiniGrid-LavaGapS7-v0 1048576
run MiniGrid-DoorKey-8x8-v0 1048576
run MiniGrid-DoorKey-5x5-v0 1048576
run MiniGrid-Empty-Random-6x6-v0 1048576
run MiniGrid-SimpleCrossingS11N5-v0 1048576
echo "# kernel timeline, LavaCrossingS9N1 1 Mi envs, default form (us from the first row; q = HSA queue)"
PY=$(readlink -f "$(command -v python3)")
OUT=$R/gpurun_out/tl_ring
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- $PY $R/bench.py --config lava4m --envs-per-gpu 1048576 --new-level-each-episode --steps 100 --warmup 20 --no-cpu-baseline > /dev/null 2>&1 || echo "rocprofv3 failed"
f=$(find $OUT -name "*kernel_trace.csv" | head -n 1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mid = len(rows) // 2
t0 = int(rows[mid]["Start_Timestamp"])
for r in rows[mid:mid + 36]:
    print("%-12s q%-3s start %8.1f end %8.1f dur %6.1f" % (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:10], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $OUT
