import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("GCS_DEBUG_ABORT"):
        # diagnostics for a native abort(): a SIGABRT handler that prints the raising thread's native backtrace
        # (tools/dbg/abrt_bt.c; run with -p no:faulthandler) and the objects each cyclic collection frees
        import ctypes
        import gc
        import torch  # noqa: F401  (load the HIP runtime first: whoever installs handlers at load time goes before us)
        ctypes.CDLL(os.path.join(ROOT, "tools", "dbg", "abrt_bt.so")).abrt_bt_install()
        if "gc" in os.environ["GCS_DEBUG_ABORT"]:
            gc.set_debug(gc.DEBUG_COLLECTABLE)
        if "cycle" in os.environ["GCS_DEBUG_ABORT"]:
            # re-create round 4's reference cycle (graph entry -> plan): dead plans then pile up until a cyclic collection,
            # the condition under which the session aborted (profiles/r4_notes.md)
            from gabor_color_image_segmentation_amd import segmenter as _sg
            _orig = _sg.Segmenter._segment_small

            def _cyclic(self, *a, **kw):
                out = _orig(self, *a, **kw)
                for ent in self._graphs.values():
                    ent["_plan"] = self
                return out
            _sg.Segmenter._segment_small = _cyclic


@pytest.fixture(scope="session")
def built():
    """Make sure libgcs.so and the C oracle exist (cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True
