#!/bin/bash
# PMC passes over the bf16 kernel:  gpurun -- 'bash tools/pmc16.sh <tag>'
set -e -o pipefail
tag=${1:-pmc16}; out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
B="python3 bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline"
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$out/p1" -- $B > "$out/p1.log" 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM -d "$out/p2" -- $B > "$out/p2.log" 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -d "$out/p3" -- $B > "$out/p3.log" 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$out/p4" -- $B > "$out/p4.log" 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$out/p5" -- $B > "$out/p5.log" 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > "$out/stats.log" 2>&1
cp "$out"/stats/*/*_kernel_stats.csv "$out/kernel_stats_bf16.csv"
python3 tools/pmc_summary.py ${BF16_KERNEL:-k_eval_t16} "$out/pmc_${BF16_KERNEL:-k_eval_t16}.json" "$out/p1" "$out/p2" "$out/p3" "$out/p4" "$out/p5"
rm -rf "$out/p1" "$out/p2" "$out/p3" "$out/p4" "$out/p5" "$out/stats"
