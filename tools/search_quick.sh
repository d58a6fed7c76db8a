set -e -o pipefail
out=gpurun_out/sr2j; mkdir -p $out
for cfg in "d4_b2048_t12 --depth 4 --batch 2048 --dtype bf16 --threads 12" "d4_b2048_t1 --depth 4 --batch 2048 --dtype bf16 --threads 1" "d5_b2048_t12 --depth 5 --batch 2048 --dtype bf16 --threads 12" "d5_b8192_t12 --depth 5 --batch 8192 --dtype bf16 --threads 12"; do
  set -- $cfg; name=$1; shift
  timeout -k 10 280 python3 tools/search_bench.py "$@" --out $out/$name.json > $out/$name.log 2>&1
  tail -n 1 $out/$name.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', 'nodes/s', round(d['nodes_per_s']), 'mean eval batch', round(d['mean_eval_batch']), 'wait', round(d['eval_wait_fraction'],3), {k: round(v,3) for k,v in d['stage_share'].items()})"
done
