R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sweep2.jsonl
: > $OUT
run() { timeout -k 10 240 python $R/bench.py --no-cpu-baseline --steps 256 --warmup 32 "$@" 2>/dev/null | tail -n 1 >> $OUT || echo "{\"failed\": \"$*\"}" >> $OUT; }
run --env MiniGrid-GoToObject-8x8-N2-v0
run --env MiniGrid-PutNear-8x8-N3-v0
run --env MiniGrid-RedBlueDoors-8x8-v0
run --env MiniGrid-MemoryS13Random-v0 --envs-per-gpu 524288
run --env MiniGrid-MemoryS17Random-v0 --envs-per-gpu 262144
run --env MiniGrid-UnlockPickup-v0
run --env MiniGrid-KeyCorridorS3R3-v0
run --env MiniGrid-KeyCorridorS6R3-v0 --envs-per-gpu 524288
run --env MiniGrid-LockedRoom-v0 --envs-per-gpu 262144
run --env MiniGrid-Playground-v0 --envs-per-gpu 262144
run --env MiniGrid-KeyCorridorS3R3-v0 --new-level-each-episode
run --env MiniGrid-PutNear-8x8-N3-v0 --new-level-each-episode
run --env MiniGrid-LockedRoom-v0 --envs-per-gpu 262144 --new-level-each-episode
run --env MiniGrid-Fetch-8x8-N3-v0 --new-level-each-episode
echo done
