#!/bin/bash
# HBM traffic of one evaluation kernel launched back to back with nothing else in the process:
#   gpurun -- 'bash tools/pmc_alone.sh <tag>'   ->  gpurun_out/<tag>/pmc_alone_{f32,bf16}.json
set -e -o pipefail
tag=${1:-pmc_alone}; out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
for dt in ${PMC_DTYPES:-f32 f32s f32s3 bf16}; do
  k=k_eval_f32; [ $dt = bf16 ] && k=k_eval_t16
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$out/f_$dt" -- python3 tools/kernel_only.py $dt 30 > "$out/f_$dt.log" 2>&1
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$out/w_$dt" -- python3 tools/kernel_only.py $dt 30 > "$out/w_$dt.log" 2>&1
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$out/s_$dt" -- python3 tools/kernel_only.py $dt 30 > "$out/s_$dt.log" 2>&1
  sha=$(grep -o "source_sha16 [0-9a-f]*" "$out/f_$dt.log" | cut -d' ' -f2)
  PMC_SOURCE_SHA16=$sha python3 tools/pmc_summary.py $k "$out/pmc_alone_$dt.json" "$out/f_$dt" "$out/w_$dt" "$out/s_$dt" > /dev/null
  rm -rf "$out/f_$dt" "$out/w_$dt" "$out/s_$dt"
  python3 -c "
import json; d=json.load(open('$out/pmc_alone_$dt.json')); print('$dt', d.get('source_sha16'), d['hbm_bytes_per_launch'], d['hbm_bytes_per_launch_median'])"
done
