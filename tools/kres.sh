#!/bin/bash
# Compile one translation unit of libgcs.so with -save-temps under /tmp and print per-kernel VGPR / spill / LDS figures:
#   tools/kres.sh kmeans [grep pattern] [extra hipcc flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
f=${1:-kmeans}; pat=${2:-.}; shift; shift
mkdir -p /tmp/kres && cd /tmp/kres && rm -f $f-*
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -Wall -Wno-unused-function -Wno-unused-variable \
    -fno-slp-vectorize "$@" -c -save-temps -o /tmp/kres/$f.o $ROOT/gabor_color_image_segmentation_amd/csrc/$f.hip 2>&1 | grep -E "error|warning" | head -20
python3 $ROOT/tools/kernel_resources.py /tmp/kres/$f-hip-amdgcn-amd-amdhsa-gfx950.s | grep -E "$pat"
