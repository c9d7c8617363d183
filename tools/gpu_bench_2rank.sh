#!/bin/bash
# the N > 1 control flow of bench.py on a one-GPU box: 2 ranks share cuda:0, collective over gloo (RTW_BENCH_ONE_DEVICE=1)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for coll in reduce gather; do
RTW_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --spp 100 --no-cpu-baseline --collective $coll 2>&1 | grep -E "^\{|rror|Traceback" | cut -c1-600
done
