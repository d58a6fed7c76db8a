#!/bin/bash
# Build a variant of the library for A/B timing:  tools/build_variant.sh <name> [extra hipcc flags...]
# Output: build/variants/libbf_<name>.so  (travels to the GPU box; build/ is git-ignored)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/build/variants"
cd "$root/batch_first_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -mllvm -pragma-unroll-threshold=4000000 -mllvm -amdgpu-mfma-vgpr-form "$@" \
    api.hip encode.hip eval_f32.hip eval_bf16.hip pack.cpp pack_bf16.cpp host_encode.cpp -o "$root/build/variants/libbf_$name.so"
echo "built build/variants/libbf_$name.so"
