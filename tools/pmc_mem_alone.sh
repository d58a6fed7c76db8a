#!/bin/bash
# Memory-path counters of one evaluation kernel alone in the process:  gpurun -- 'bash tools/pmc_mem_alone.sh <tag> <dtype>'
set -e -o pipefail
tag=${1:-pmc_mem}; dt=${2:-f32s}; out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
k=k_eval_f32; [ $dt = bf16 ] && k=k_eval_t16
timeout -s KILL 120 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -d "$out/m1_$dt" -- python3 tools/kernel_only.py $dt 20 > "$out/m1_$dt.log" 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d "$out/m2_$dt" -- python3 tools/kernel_only.py $dt 20 > "$out/m2_$dt.log" 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum -d "$out/m3_$dt" -- python3 tools/kernel_only.py $dt 20 > "$out/m3_$dt.log" 2>&1 || echo "pass 3 failed (counter names)"
python3 tools/pmc_summary.py $k "$out/pmc_mem_$dt.json" "$out/m1_$dt" "$out/m2_$dt" "$out/m3_$dt" > /dev/null || python3 tools/pmc_summary.py $k "$out/pmc_mem_$dt.json" "$out/m1_$dt" "$out/m2_$dt" > /dev/null
rm -rf "$out/m1_$dt" "$out/m2_$dt" "$out/m3_$dt"
python3 -c "
import json; d=json.load(open('$out/pmc_mem_$dt.json'))
for k,v in d['counters'].items(): print('$dt', k, v['median_per_launch'])"
