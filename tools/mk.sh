#!/bin/bash
# Build libgcs.so in-tree (from any directory) and, with a name, copy it to build_ab/<name>.so for tools/ab.py:  tools/mk.sh [name]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -j4 -C "$ROOT/gabor_color_image_segmentation_amd/csrc" 2>&1 | grep -E "error|Error" -A4 || true
ls -la "$ROOT/gabor_color_image_segmentation_amd/csrc/libgcs.so"
if [ -n "${1:-}" ]; then mkdir -p "$ROOT/build_ab"; cp "$ROOT/gabor_color_image_segmentation_amd/csrc/libgcs.so" "$ROOT/build_ab/$1.so"; ls "$ROOT/build_ab"; fi
