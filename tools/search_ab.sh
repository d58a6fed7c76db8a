#!/bin/bash
# Two builds of the search library on ONE box, interleaved:  gpurun -- 'bash tools/search_ab.sh <other.so> [search_bench args]'
# (box-to-box differences of the host are as large as most changes to the search host)
set -e -o pipefail
other=$1; shift
out=gpurun_out/search_ab; mkdir -p $out
for round in 1 2 3; do
  for v in this other; do
    if [ $v = other ]; then export BATCHFIRST_SEARCH_LIB=$other; else unset BATCHFIRST_SEARCH_LIB; fi
    timeout -k 10 200 python3 tools/search_bench.py "$@" --out $out/$v$round.json > $out/$v$round.log 2>&1
    tail -n 1 $out/$v$round.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', $round, 'nodes/s', round(d['nodes_per_s']), 'nodes out', d['nodes_out'], 'boards', d['boards_evaluated'], 'iterations', d['iterations'], {k: round(v,3) for k,v in d['stage_share'].items()})"
  done
done
