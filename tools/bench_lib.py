#!/usr/bin/env python3
"""bench.py on another build of libgcs.so, same-box comparisons in bench.py's own steady-state conditions:

    python tools/bench_lib.py build_ab/r3.so [bench.py arguments...]

An older build may be given as long as the entry points bench.py uses kept their signatures (the ABI version check is relaxed
to the build's own number, as in tools/ab.py)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
import torch  # noqa: F401  (its HIP runtime first: a library loaded before it would bind /opt/rocm's second copy)
from gabor_color_image_segmentation_amd import _lib
hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
if os.path.exists(hip_rt):
    ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
_lib.LIB_PATH = lib
_lib.ABI_VERSION = ctypes.CDLL(lib).gcs_abi_version()
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench
bench.main()
