#!/bin/bash
# Compile csrc/gabor.hip with -save-temps into /tmp/gasm/ and print the resource table of the MFMA kernels.
# usage: tools/gabor_asm.sh [extra hipcc flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/gasm && cd /tmp/gasm && rm -f gabor-hip-*.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -fno-slp-vectorize "$@" -c -save-temps -o /tmp/gasm/gabor.o \
    $ROOT/gabor_color_image_segmentation_amd/csrc/gabor.hip 2>&1 | grep -E "error" 
python3 $ROOT/tools/kernel_resources.py /tmp/gasm/gabor-hip-amdgcn-amd-amdhsa-gfx950.s | grep mfma | sed 's/_Z17gabor_mfma_kernel//'
