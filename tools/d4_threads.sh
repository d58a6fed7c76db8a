#!/bin/bash
# depth 4 on one box: every batch on all pool threads (BFS_WIDE_BATCH=0), the default switch (320 nodes), every batch
# on the narrow group (three threads), interleaved
set -e -o pipefail
mkdir -p gpurun_out/d4t
for round in 1 2 3; do
 for w in 0 320 1000000; do
  BFS_WIDE_BATCH=$w timeout -k 10 200 python3 tools/search_bench.py --depth 4 --batch 2048 --dtype bf16 --threads 12 --out gpurun_out/d4t/w${w}_$round.json > gpurun_out/d4t/w${w}_$round.log 2>&1
  tail -n 1 gpurun_out/d4t/w${w}_$round.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('wide_batch', $w, 'round', $round, 'nodes/s', round(d['nodes_per_s']), {k: round(v,3) for k,v in d['stage_share'].items()})"
 done
done
