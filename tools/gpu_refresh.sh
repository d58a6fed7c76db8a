#!/bin/bash
# Round evidence on the GPU box: parity tests, the bench line, rocprofv3 kernel stats and PMC passes.
#   gpurun --timeout 900 -- 'bash tools/gpu_refresh.sh <tag>'
# Everything lands under gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -e -o pipefail
tag=${1:-refresh}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$out/pytest_gpu.log" 2>&1 && tail -2 "$out/pytest_gpu.log"
timeout -k 10 240 python bench.py --steps 20 --warmup 3 > "$out/bench.json" 2> "$out/bench.err" && cat "$out/bench.json"
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$out/stats.log" 2>&1
cp "$out"/stats/*/*_kernel_stats.csv "$out/kernel_stats.csv"
head -4 "$out/kernel_stats.csv" | cut -c1-200
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$out/pmc_sq" -- $B > "$out/pmc_sq.log" 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d "$out/pmc_fetch" -- $B > "$out/pmc_fetch.log" 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d "$out/pmc_write" -- $B > "$out/pmc_write.log" 2>&1
python3 tools/pmc_summary.py k_eval_f32 "$out/pmc_k_eval_f32.json" "$out/pmc_sq" "$out/pmc_fetch" "$out/pmc_write" > /dev/null
python3 tools/pmc_summary.py k_to_planes "$out/pmc_k_to_planes.json" "$out/pmc_sq" "$out/pmc_fetch" "$out/pmc_write" > /dev/null
rm -rf "$out/pmc_sq" "$out/pmc_fetch" "$out/pmc_write" "$out/stats"
echo done
