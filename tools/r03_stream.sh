#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "stream or levelgen or seeded_reset or new_level or task_families_on_device" 2>&1 | tail -n 3
for e in MiniGrid-MultiRoom-N6-v0 MiniGrid-MultiRoom-N4-S5-v0 MiniGrid-KeyCorridorS3R3-v0 MiniGrid-Fetch-8x8-N3-v0 MiniGrid-LockedRoom-v0 MiniGrid-ObstructedMaze-2Dlhb-v0; do timeout -k 10 200 python bench.py --no-cpu-baseline --env $e --envs-per-gpu 262144 --new-level-each-episode --steps 600 --warmup 32 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j[\"config\"][\"env_id\"], \"%.3g steps/s\"%j[\"value\"], \"%.1f us/step\"%(j[\"ms_per_step\"]*1e3), j.get(\"episodes\"))"; done
timeout -k 10 200 python bench.py --no-cpu-baseline --config lava4m --envs-per-gpu 1048576 --new-level-each-episode --steps 256 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j[\"config\"][\"env_id\"], \"%.3g steps/s\"%j[\"value\"], \"%.1f us/step\"%(j[\"ms_per_step\"]*1e3), j.get(\"episodes\"))"
} 2>&1 | tee $O/stream.txt
