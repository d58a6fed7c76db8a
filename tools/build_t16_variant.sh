#!/bin/bash
# Variant of the library that differs in eval_t16.hip's build flags only (the other objects are reused):
#   tools/build_t16_variant.sh <name> [extra hipcc flags...]   ->  build/variants/libbf_<name>.so
# Load it with BATCHFIRST_LIB=build/variants/libbf_<name>.so (batch_first_amd/_lib.py) or tools/ab_time.py.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/build/variants/obj_$name"
cd "$root/build/variants/obj_$name"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -mllvm -pragma-unroll-threshold=4000000 \
    -mllvm -amdgpu-mfma-vgpr-form "$@" --save-temps -c "$root/batch_first_amd/csrc/eval_t16.hip" -o eval_t16.o
python3 "$root/tools/isa_hazards.py" eval_t16-hip-amdgcn-amd-amdhsa-gfx950.s
grep -A90 "k_eval_t16ILi0E" eval_t16-hip-amdgcn-amd-amdhsa-gfx950.s | grep -m3 "spill_count\|private_segment_fixed" || true
cd "$root/batch_first_amd/csrc"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 api.o encode.o eval_f32.o eval_bf16.o "$root/build/variants/obj_$name/eval_t16.o" \
    pack.o pack_bf16.o pack_t16.o host_encode.o -o "$root/build/variants/libbf_$name.so"
rm -f "$root/build/variants/obj_$name"/*.hipi "$root/build/variants/obj_$name"/*.bc
echo "built build/variants/libbf_$name.so"
