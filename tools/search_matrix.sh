#!/bin/bash
# configs[4] on the GPU box: depth x batch x host threads;  gpurun -- 'bash tools/search_matrix.sh <tag>'
set -e -o pipefail
tag=${1:-search}; out=gpurun_out/$tag; mkdir -p "$out"
run() {  # name, args...
    local name=$1; shift
    timeout -k 10 280 python3 tools/search_bench.py "$@" --out "$out/$name.json" > "$out/$name.log" 2>&1
    tail -1 "$out/$name.log" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', 'threads', d['host_threads'], 'nodes/s', round(d['nodes_per_s']), 'mean eval batch', round(d['mean_eval_batch']), 'max', d['max_eval_batch'], 'nodes/iter', round(d['mean_nodes_per_iteration']), 'wait', round(d['eval_wait_fraction'],3), {k: round(v,3) for k,v in d['stage_share'].items()})"
}
run d4_b2048_t1  --depth 4 --batch 2048 --dtype bf16 --threads 1
run d4_b2048_t12 --depth 4 --batch 2048 --dtype bf16 --threads 12
run d4_b2048_t12_f32 --depth 4 --batch 2048 --dtype f32 --threads 12
run d4_b2048_t12_f32s --depth 4 --batch 2048 --dtype f32s --threads 12
run d4_b2048_t12_f32s3 --depth 4 --batch 2048 --dtype f32s3 --threads 12
run d5_b2048_t12 --depth 5 --batch 2048 --dtype bf16 --threads 12
run d5_b8192_t1  --depth 5 --batch 8192 --dtype bf16 --threads 1
run d5_b8192_t6 --depth 5 --batch 8192 --dtype bf16 --threads 6
run d5_b8192_t12 --depth 5 --batch 8192 --dtype bf16 --threads 12
run d5_b8192_t15 --depth 5 --batch 8192 --dtype bf16 --threads 15
run d5_b8192_tdefault --depth 5 --batch 8192 --dtype bf16
run d5_b8192_t24 --depth 5 --batch 8192 --dtype bf16 --threads 24
run d5_b32768_t12 --depth 5 --batch 32768 --dtype bf16 --threads 12
run d6_b32768_t12 --depth 6 --batch 32768 --dtype bf16 --threads 12 --fens 3
run d5_b8192_t12_material --depth 5 --batch 8192 --threads 12 --cpu-material
