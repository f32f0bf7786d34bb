#!/bin/bash
# Round 3, first GPU call: the suite, the driver's bench command, the one-GPU rehearsal of the N > 1 path (gloo: RCCL refuses two ranks
# on one GPU), and the counters VERDICT r02 asked for (gather form FETCH/WRITE, SQ counters of the 16x16 instances).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "gputest rc=$?" | tee -a $O/summary.txt
tail -n 3 $O/gputest.log | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err && echo "bench k20 ok" | tee -a $O/summary.txt
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err && echo "bench default ok" | tee -a $O/summary.txt
timeout -k 10 400 python bench.py --gpus 2 --rehearse-on-one-gpu --config lava4m --steps 20 --warmup 5 > $O/rehearse_k20.json 2> $O/rehearse_k20.err; echo "rehearse rc=$?" | tee -a $O/summary.txt
timeout -k 10 400 python bench.py --gpus 2 --rehearse-on-one-gpu --config lava4m --steps 1024 --warmup 64 --log-every 64 > $O/rehearse_k1024.json 2> $O/rehearse_k1024.err; echo "rehearse1024 rc=$?" | tee -a $O/summary.txt
