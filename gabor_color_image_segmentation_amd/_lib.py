"""ctypes binding of libgcs.so (include/gcs.h). Loader and pointer-passer only.

The product path has NO CPU fallback: if the shared library is missing or a call
fails, this raises. Build it with ``python -c 'import __graft_entry__ as g; g.build()'``
or ``make -C gabor_color_image_segmentation_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GCS_LIB_PATH") or os.path.join(_HERE, "csrc", "libgcs.so")  # override: A/B builds

ABI_VERSION = 18
K_MAX = 16

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t

# name -> (restype, argtypes): every symbol include/gcs.h declares
SIGNATURES = {
    "gcs_abi_version": (_i, []),
    "gcs_last_error": (C.c_char_p, []),
    "gcs_bank_packed_bytes": (_sz, [_i, _i]),
    "gcs_bank_bias_count": (_sz, [_i, _i]),
    "gcs_bank_pack": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "gcs_feature_slab_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "gcs_feature_pass_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "gcs_label_slab_bytes": (_sz, [_i, _i, _i]),
    "gcs_kmeans_parts_per_image": (_sz, [_i, _i, _i]),
    "gcs_kmeans_partial_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "gcs_gabor_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "gcs_gabor_features": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gcs_features_unpack": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "gcs_kmeans_init": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "gcs_features_gather": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gcs_kmeans_assign_accumulate": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gcs_kmeans_assign_raster": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "gcs_kmeans_reduce": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "gcs_kmeans_finalize": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "gcs_kmeans_reduce_finalize": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gcs_download": (_i, [_vp, _vp, _sz, _vp]),
    "gcs_labels_widen": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "gcs_selftest_isqrt": (_i, [C.c_uint, _vp, _vp]),
    "gcs_selftest_native_parts": (_i, [_i, _i, _i]),
    "gcs_device_cu_count": (_i, []),
    "gcs_boundary_scratch_bytes": (_sz, [_i, _i, _i]),
    "gcs_boundary_counts": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "gcs_region_counts": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gcs_boundary_batch_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "gcs_boundary_counts_batch": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gcs_region_counts_batch": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gcs_region_counts_batch_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gcs_region_reduce": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "gcs_score_batch_resident": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp]),
    "gcs_bit_planes_bytes": (_sz, [_i, _i, _i]),
    "gcs_truth_prepare": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gcs_boundary_counts_resident": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gcs_connected_scratch_bytes": (_sz, [_i, _i, _i]),
    "gcs_connected_regions": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
}


class GcsError(RuntimeError):
    pass


_lib = None


def load():
    """Load libgcs.so (once). Raises GcsError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GcsError(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU "
            "fallback; run `python -c 'import __graft_entry__ as g; g.build()'`.")
    # One HIP runtime per process: import torch first so that libgcs.so's NEEDED
    # libamdhip64.so.7 binds (by SONAME) to the runtime torch already loaded, instead of
    # pulling a second copy from /opt/rocm that knows nothing of torch's streams/memory.
    import torch
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    if lib.gcs_abi_version() != ABI_VERSION:
        raise GcsError(f"libgcs ABI {lib.gcs_abi_version()} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gcs_last_error().decode(errors="replace")
        raise GcsError(f"{what} failed (code {rc}): {msg}")
