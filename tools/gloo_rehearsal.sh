#!/bin/bash
# Two gloo ranks on one GPU rehearse `bench.py --gpus 2 --workload shard2p24` (launch line, shard arithmetic, gather):
#   gpurun -- 'bash tools/gloo_rehearsal.sh <tag>'  ->  gpurun_out/<tag>/shard2p24_gloo2.json
set -e -o pipefail
tag=${1:-gloo}; out=gpurun_out/$tag; mkdir -p "$out"
BF_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29533 bench.py --gpus 2 --workload shard2p24 --steps 2 --warmup 1 > "$out/shard2p24_gloo2.json" 2> "$out/shard2p24_gloo2.err"
tail -c 1500 "$out/shard2p24_gloo2.json"
