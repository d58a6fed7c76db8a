#!/bin/bash
# Variant of the library that differs in ONE kernel source's build flags (the other objects are reused):
#   tools/build_variant.sh <name> <eval_f32|eval_t16|eval_t16s> [extra hipcc flags...]  ->  build/variants/libbf_<name>.so
# Load it with BATCHFIRST_LIB=build/variants/libbf_<name>.so (batch_first_amd/_lib.py), tools/ab_time.py or tools/stamps.py.
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
obj="$root/build/variants/obj_$name"
mkdir -p "$obj"
cd "$obj"
extra=""
[ "$src" != eval_f32 ] && extra="-mllvm -pragma-unroll-threshold=4000000 -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm $extra "$@" --save-temps \
    -c "$root/batch_first_amd/csrc/$src.hip" -o $src.o
python3 "$root/tools/isa_hazards.py" $src-hip-amdgcn-amd-amdhsa-gfx950.s
grep "vgpr_spill_count\|\.vgpr_count\|\.agpr_count" $src-hip-amdgcn-amd-amdhsa-gfx950.s | head -9 || true
cd "$root/batch_first_amd/csrc"
objs=""
for o in api encode eval_f32 eval_f32q eval_t16 eval_t16s pack pack_t16 host_encode; do
    if [ "$o" = "$src" ]; then objs="$objs $obj/$o.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o "$root/build/variants/libbf_$name.so"
rm -f "$obj"/*.hipi "$obj"/*.bc
echo "built build/variants/libbf_$name.so"
