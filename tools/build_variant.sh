#!/bin/bash
# Build a variant of libgcs.so for same-box A/B runs without touching the in-tree objects:
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DGCS_GABOR_STRIPS_BESIDE=0]   ->  build_ab/<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
B=/tmp/gcs_variant_$name
mkdir -p $B $ROOT/build_ab
for f in abi gabor kmeans scoring; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -Wno-unused-function -Wno-unused-variable \
      -fno-slp-vectorize "$@" -c -o $B/$f.o $ROOT/gabor_color_image_segmentation_amd/csrc/$f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_ab/$name.so $B/abi.o $B/gabor.o $B/kmeans.o $B/scoring.o
ls -la $ROOT/build_ab/$name.so
