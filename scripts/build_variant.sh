#!/bin/bash
# build_variant.sh NAME [-DFLAG...]: libafter_hip.so with denoiser.hip compiled under extra flags -> scripts/variants/NAME/
# (kernel A/B experiments on one GPU lease: scripts/gpu_pass.sh swaps the variant in for a timing run and restores the default)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/scripts/variants/$name
mkdir -p "$d"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form "$@" \
    -c "$root/after_amd/csrc/denoiser.hip" -o "$d/denoiser.o" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "sample_seg_kernelILi6" | grep "VGPRs Spill\|VGPRs:" | sed "s/^.*remark: /$name:/"
objs=$(ls "$root"/after_amd/lib/*.o | grep -v denoiser.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/libafter_hip.so" $objs "$d/denoiser.o"
