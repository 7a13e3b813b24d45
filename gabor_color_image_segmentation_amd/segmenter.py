"""Host side of the segmenter slot: ``labels = segment(img)``.

Mirrors the only interface the reference has for this path — the positional call at
/root/reference/BSD_metrics/script.py:30 (``labels = slic(img, ...)``): ``img`` is the
(H,W,3) uint8 array from script.py:25, the result is a fresh (H,W) integer array for
metrics.__init__ (/root/reference/BSD_metrics/metrics.py:43-51). Errors are plain Python
exceptions (script.py:19-38 catches nothing).

PyTorch is plumbing only: device memory, the current HIP stream and torch.distributed
(RCCL). All arithmetic runs in libgcs.so (csrc/gcs.hip) through the C ABI
(include/gcs.h). There is no CPU fallback.
"""
from __future__ import annotations

import math
import os

import numpy as np

from . import _lib
from .bank import GaborBank, make_bank

# Feature-slab bytes per group of a per-image batch. Large on purpose: measured in round 4 with the current kernels
# (tools/cache_resident_pass.py, profiles/r4_notes.md "Infinity-Cache-resident groups"): a pass that re-reads a slab small enough
# to stay in the 256 MiB Infinity Cache (8 images, 112 MB) streams it at 3.8 TB/s, 16 images (224 MB) at 4.9 TB/s, the
# 64-image slab (895 MB, from HBM) at 5.9 TB/s - a pass costs ~12 us of launch, prologue and fold however few bytes it reads,
# and what remains is the same 6.4-6.6 TB/s per byte in all three cases (the staging structure, not HBM, sets the rate). 64
# images in cache-sized groups of 16 take 2.45 ms against 2.04 ms in one group.
_SLAB_BUDGET = 16 << 30
# host calls up to this many pixels replay a captured HIP graph of the whole step
_GRAPH_MAX_PIXELS = 1 << 20


class DebugSwitches:
    """The ONE set of measurement / test switches of the host side; none of them changes a result.

    Set from the environment, ``GCS_DEBUG=no_reverse,no_graph,force_collectives,slab_candidates=2`` (any subset), or per
    plan through ``Segmenter.debug``:

    * ``no_reverse``: every Lloyd pass sweeps the slab in the same direction (tools/pass_direction_probe.py);
    * ``no_graph``: small host calls launch eagerly instead of replaying their captured HIP graph;
    * ``force_collectives``: ``lloyd`` takes the multi-rank branch (reduce, all-reduce, finalize) even in a one-rank
      group, so that one GPU can exercise the RCCL path end to end (tests/test_distributed.py, tools/rccl_overhead.py);
    * ``slab_candidates=N``: overrides ``Segmenter(slab_candidates=...)`` (see ``Segmenter._place_slab``)."""

    def __init__(self, spec=""):
        self.no_reverse = self.no_graph = self.force_collectives = False
        self.slab_candidates = None
        for item in filter(None, (x.strip() for x in spec.split(","))):
            name, _, val = item.partition("=")
            if name == "slab_candidates" and val.isdigit():
                self.slab_candidates = int(val)
            elif name in ("no_reverse", "no_graph", "force_collectives") and not val:
                setattr(self, name, True)
            else:
                raise ValueError(f"GCS_DEBUG: unknown switch {item!r}")


_ENV_DEBUG = DebugSwitches(os.environ.get("GCS_DEBUG", ""))



class _CaptureGuard:
    """Process-wide order for HIP graph captures of this module.

    A torch ``CUDAGraph`` destroyed while a stream capture is open throws out of its destructor and ``std::terminate`` ends the
    process (profiles/r4_notes.md). Three rules keep that from happening through this module, whatever threads use it:

    * captures are SERIALISED: one lock is held from the garbage collection in front of a capture to its end, so the cyclic
      collector - a process-global switch - is turned off by the one capturer there is and turned back on (if it was on) when
      that capture has ended, never inside another thread's open capture;
    * graph entries are never dropped where a capture might be open: evicted entries and the entries of a Segmenter that is
      being finalised go through ``retire``, which frees them at once when no capture is open and parks them otherwise
      (drained at the end of the capture, under the lock);
    * a capture that fails is not silent: ``fell_back`` warns once per plan key;
    * a graph whose capture failed is never destroyed: torch 2.10's ``~CUDAGraph`` throws for it too ("The graph should be
      registered to the state", tools/dbg/capture_probe.py) - the few hundred bytes stay in ``_failed`` for the life of the process.

    What a thread that is NOT capturing may do beside an open capture was measured on the box (tools/dbg/capture_probe.py,
    profiles/r5_notes.md): kernel launches, stream / event synchronisation, hipMalloc, pinned allocation and copies are all
    fine; ``hipDeviceSynchronize`` is refused with hipErrorStreamCaptureUnsupported AND invalidates the other thread's
    capture - so no path of this package calls ``torch.cuda.synchronize``: every wait is on a stream or an event. A CALLER's
    thread still can: ``capture`` therefore runs on a helper thread (see there), which confines the damage of an invalidated
    capture to a thread and a stream nobody uses again."""

    def __init__(self):
        import threading
        self._lock = threading.Lock()      # not re-entrant: a plan finalised BY the capturing thread must park its graphs too
        self._parked = []            # retired graph entries waiting for the open capture to end
        self._failed = []            # graphs whose capture was refused or invalidated: kept alive on purpose (see above)
        self._warned = set()

    def capture(self, torch, graph, body, device=None):
        """Capture ``body()`` into ``graph`` (thread_local error mode). Returns True, or False when the capture was refused or
        invalidated (the caller then launches eagerly).

        The capture runs on a HELPER THREAD, and on a stream of its own. Measured on the box (tools/dbg/invalidated_capture_probe.py,
        capture_helper_thread_probe.py, profiles/r6_notes.md): a device-wide synchronize from ANY other thread - a caller's loader, a
        logger - is refused by HIP while a capture is open in whatever mode AND invalidates that capture, and an invalidated capture
        leaves the thread that opened it unable to launch another kernel (hipErrorStreamCaptureInvalidated from every later launch;
        hipStreamEndCapture does not clear it) and its capture stream in the invalidated state for good. Bound to the helper, that
        damage dies with it: the caller's thread and streams stay usable, the shape runs as eager launches, and the next capture
        gets a fresh helper and a fresh stream."""
        import gc
        import threading
        with self._lock:
            gc.collect()             # dead cycles that hold graphs: finalise them NOW, not between capture_begin and capture_end
            was_enabled = gc.isenabled()
            gc.disable()
            result = {}
            stream = torch.cuda.Stream(device=device)

            def run():
                try:
                    if device is not None:
                        torch.cuda.set_device(device)
                    with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                        body()
                    result["ok"] = True
                except RuntimeError as e:          # (GcsError and torch.AcceleratorError are RuntimeErrors)
                    result["err"] = e
                except BaseException as e:         # anything else is the caller's to see
                    result["raise"] = e

            try:
                helper = threading.Thread(target=run, name="gcs-graph-capture")
                helper.start()
                helper.join()
                if "raise" in result:
                    self._failed.append((graph, stream))
                    raise result["raise"]
                if result.get("ok"):
                    return True
                self._failed.append((graph, stream))      # never destroyed, never reused (see the class docstring)
                return False
            finally:
                if was_enabled:
                    gc.enable()
                self._drain()

    def retire(self, entries):
        """Drop graph entries (dicts holding a CUDAGraph and its buffers): now if no capture is open, else after it. ``entries``
        is a LIST the caller gives up: it is emptied here, so that the last references die inside the locked region (or move to
        the parking list) and not later, when the caller's frame unwinds and another thread may have opened a capture."""
        if self._lock.acquire(blocking=False):
            try:
                self._drain()
                entries.clear()      # (the last references: the graphs are destroyed here, under the lock, outside any capture)
            finally:
                self._lock.release()
        else:
            self._parked.extend(e for e in entries if e is not None)      # drained by the capturer, under the lock
            entries.clear()

    def _drain(self):
        while self._parked:
            self._parked.pop()

    def fell_back(self, key, err):
        if key not in self._warned:
            self._warned.add(key)
            import warnings
            warnings.warn(f"gabor_color_image_segmentation_amd: HIP graph capture refused for plan {key} ({err}); this shape runs "
                          "as eager kernel launches for the life of the plan (slower one-image calls, same results)",
                          RuntimeWarning, stacklevel=3)


_CAPTURES = _CaptureGuard()


def _torch():
    import torch
    return torch


def _on_device(fn):
    """Run a HipOps entry point with the plan's device current (a Segmenter on cuda:1 used while cuda:0 is current
    would otherwise launch on the wrong device) after checking where its tensor arguments live."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kw):
        self._check_dev(*[a for a in list(args) + list(kw.values()) if hasattr(a, "data_ptr") and hasattr(a, "device")])
        with self.torch.cuda.device(self.device):
            return fn(self, *args, **kw)
    return wrapper


class HipOps:
    """Thin pointer-passer over the C ABI for one device. Stateless apart from the bank."""

    def __init__(self, bank: GaborBank, device):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.GcsError("no HIP device visible: the segmenter has no CPU fallback")
        self.lib = _lib.load()
        self.torch = torch
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.bank = bank
        ns, no = bank.n_scales, bank.n_orient
        self._bk = (ns, no)                      # the bank as the C ABI describes it
        packed = np.zeros(self.lib.gcs_bank_packed_bytes(ns, no), np.int8)
        bias = np.zeros(self.lib.gcs_bank_bias_count(ns, no), np.int32)
        tapq = np.ascontiguousarray(bank.tapq, np.int16)
        _lib.check(self.lib.gcs_bank_pack(tapq.ctypes.data, ns, no, bank.ksize, packed.ctypes.data,
                                          bias.ctypes.data), "gcs_bank_pack")
        self.packed = torch.from_numpy(packed).to(self.device)
        self.bias = torch.from_numpy(bias).to(self.device)
        self._gabor_ws = None

    # ---- allocation helpers (bytes buffers; layouts are opaque, see gcs.h)
    def empty_bytes(self, n):
        return self.torch.empty(int(n), dtype=self.torch.uint8, device=self.device)

    def feature_slab(self, b, h, w):
        return self.empty_bytes(self.lib.gcs_feature_slab_bytes(b, h, w, *self._bk))

    def label_slab(self, b, h, w):
        return self.empty_bytes(self.lib.gcs_label_slab_bytes(b, h, w))

    def partial_slab(self, b, h, w, k):
        return self.empty_bytes(self.lib.gcs_kmeans_partial_bytes(b, h, w, self.bank.n_features, k))

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def _check_dev(self, *tensors):
        """Launches go through ctypes on the calling thread's CURRENT device: every tensor must live on this plan's
        device, and the calls below run inside ``torch.cuda.device(self.device)`` (see ``_on_device``)."""
        for t in tensors:
            if t.device != self.device:
                raise ValueError(f"tensor on {t.device}, this Segmenter's kernels run on {self.device}")

    # ---- device entry points
    def gabor_scratch(self, b, h, w):
        """A private Gabor workspace for one (batch, shape): what a captured graph must own (see ``gabor_features``)."""
        return self.empty_bytes(self.lib.gcs_gabor_workspace_bytes(b, h, w, self.bank.n_scales))

    @_on_device
    def gabor_features(self, imgs, feats, scratch=None):
        """``scratch``: a workspace the CALLER owns (``gabor_scratch``). Without it the call uses the shared, growing
        workspace of this HipOps - fine for eager launches (torch's allocator hands a replaced block to later work of the
        same stream only), NOT for a captured graph: a replay would keep writing through the raw pointer of a block that
        a later, larger call has meanwhile given back to the allocator."""
        b, h, w, _ = imgs.shape
        need = self.lib.gcs_gabor_workspace_bytes(b, h, w, self.bank.n_scales)
        if scratch is None:
            if self._gabor_ws is None or self._gabor_ws.numel() < need:
                self._gabor_ws = self.empty_bytes(need)
            scratch = self._gabor_ws
        elif scratch.numel() < need or scratch.device != self.device:
            raise ValueError("gabor scratch too small or on another device")
        _lib.check(self.lib.gcs_gabor_features(imgs.data_ptr(), b, h, w, self.packed.data_ptr(),
                                               self.bias.data_ptr(), *self._bk,
                                               self.bank.ksize, self.bank.shift, scratch.data_ptr(),
                                               feats.data_ptr(), self._stream()),
                   "gcs_gabor_features")

    @_on_device
    def features_unpack(self, feats, b, h, w):
        d = self.bank.n_features
        out = self.torch.empty((b, d, h, w), dtype=self.torch.int16, device=self.device)
        _lib.check(self.lib.gcs_features_unpack(feats.data_ptr(), b, h, w, *self._bk, out.data_ptr(),
                                                self._stream()), "gcs_features_unpack")
        return out

    @_on_device
    def kmeans_init(self, feats, b, h, w, k, n_sets, cent):
        _lib.check(self.lib.gcs_kmeans_init(feats.data_ptr(), b, h, w, *self._bk, k, n_sets,
                                            cent.data_ptr(), self._stream()), "gcs_kmeans_init")

    @_on_device
    def assign_accumulate(self, feats, cent, b, h, w, k, n_sets, labels, partials, rows=None, reverse=False):
        """One Lloyd pass. ``labels=None``: the assignment is not stored (every pass but the last); ``partials=None``:
        no sums are accumulated (the last pass)."""
        lo, hi = rows if rows is not None else (0, h)
        _lib.check(self.lib.gcs_kmeans_assign_accumulate(
            feats.data_ptr(), cent.data_ptr(), b, h, w, *self._bk, k, n_sets, lo, hi,
            1 if reverse else 0, None if labels is None else labels.data_ptr(),
            None if partials is None else partials.data_ptr(), self._stream()),
            "gcs_kmeans_assign_accumulate")

    @_on_device
    def assign_raster(self, feats, cent, b, h, w, k, n_sets, out, scratch_labels=None, reverse=False):
        """The last Lloyd pass with the label map written in raster order: ``out`` (B,H,W) int32 or uint8 device tensor."""
        torch = self.torch
        if out.dtype not in (torch.int32, torch.uint8) or tuple(out.shape) != (b, h, w) or not out.is_contiguous():
            raise ValueError("out must be a contiguous (B,H,W) int32 or uint8 tensor")
        _lib.check(self.lib.gcs_kmeans_assign_raster(
            feats.data_ptr(), cent.data_ptr(), b, h, w, *self._bk, k, n_sets, 1 if reverse else 0, out.data_ptr(),
            1 if out.dtype == torch.uint8 else 0, None if scratch_labels is None else scratch_labels.data_ptr(),
            self._stream()), "gcs_kmeans_assign_raster")

    def download(self, dst_pinned, src):
        """Device tensor -> pinned host tensor of the same bytes on the current stream, by SDMA (gcs_download)."""
        self._check_dev(src)
        n = src.numel() * src.element_size()
        if not dst_pinned.is_pinned() or dst_pinned.numel() * dst_pinned.element_size() != n \
                or not src.is_contiguous() or not dst_pinned.is_contiguous():
            raise ValueError("download needs a contiguous device tensor and a pinned host tensor of the same size")
        with self.torch.cuda.device(self.device):
            _lib.check(self.lib.gcs_download(src.data_ptr(), dst_pinned.data_ptr(), n, self._stream()), "gcs_download")

    @_on_device
    def features_gather(self, feats, b, h, w, byx):
        """byx: (n,3) int32 device tensor of (image, row, col); image < 0 -> zero row. -> (n,D) int16."""
        n = byx.shape[0]
        out = self.torch.empty((n, self.bank.n_features), dtype=self.torch.int16, device=self.device)
        _lib.check(self.lib.gcs_features_gather(feats.data_ptr(), b, h, w, *self._bk, n,
                                                byx.data_ptr(), out.data_ptr(), self._stream()),
                   "gcs_features_gather")
        return out

    @_on_device
    def reduce(self, partials, b, h, w, k, n_sets, sums):
        _lib.check(self.lib.gcs_kmeans_reduce(partials.data_ptr(), b, h, w, self.bank.n_features, k,
                                              n_sets, sums.data_ptr(), self._stream()), "gcs_kmeans_reduce")

    @_on_device
    def finalize(self, sums, n_sets, k, cent):
        _lib.check(self.lib.gcs_kmeans_finalize(sums.data_ptr(), n_sets, k, self.bank.n_features,
                                                cent.data_ptr(), self._stream()), "gcs_kmeans_finalize")

    @_on_device
    def reduce_finalize(self, partials, b, h, w, k, n_sets, sums, cent):
        """Single-rank update in one launch (reduce + finalize; no collective in between)."""
        _lib.check(self.lib.gcs_kmeans_reduce_finalize(partials.data_ptr(), b, h, w, self.bank.n_features, k, n_sets,
                                                       sums.data_ptr(), cent.data_ptr(), self._stream()),
                   "gcs_kmeans_reduce_finalize")

    @_on_device
    def connected_regions(self, labels_i32, out):
        """SPEC.md §7 on an int32 (B,H,W) device tensor."""
        b, h, w = labels_i32.shape
        scratch = self.empty_bytes(self.lib.gcs_connected_scratch_bytes(b, h, w))
        _lib.check(self.lib.gcs_connected_regions(labels_i32.data_ptr(), b, h, w, scratch.data_ptr(),
                                                  out.data_ptr(), self._stream()), "gcs_connected_regions")

    @_on_device
    def labels_widen(self, labels, b, h, w, out):
        _lib.check(self.lib.gcs_labels_widen(labels.data_ptr(), b, h, w, out.data_ptr(), self._stream()),
                   "gcs_labels_widen")

    # centroid / sums tensors are ordinary torch tensors so torch.distributed can move them
    def new_centroids(self, n_sets, k):
        return self.torch.zeros((n_sets, k, self.bank.n_features), dtype=self.torch.int16, device=self.device)

    def new_sums(self, n_sets, k):
        return self.torch.zeros((n_sets, k, self.bank.n_features + 1), dtype=self.torch.int64,
                                device=self.device)


def _collective(fn, t, **kw):
    """Run a torch.distributed collective on a small tensor; gloo gets a host copy of device tensors."""
    import torch.distributed as td
    if t.is_cuda and td.get_backend(kw.get("group")) == "gloo":
        tmp = t.cpu()
        fn(tmp, **kw)
        t.copy_(tmp)
    else:
        fn(t, **kw)


def lloyd(ops, feats, b, h, w, k, n_iter, mode, labels, partials, cent, sums, dist_group=None,
          rows=None, init=None, raster=None, debug=_ENV_DEBUG):
    """SPEC.md §4 schedule on one feature slab. ``mode``: 'per_image' or 'global'.

    In 'global' mode with torch.distributed initialised, the init centroids come from
    rank 0 (its image 0 is image 0 of the global batch) and the int64 sums are all-reduced
    (RCCL on GPU, any order: integer sums are exact). One collective per Lloyd pass.
    ``rows=(lo, hi)``: only these rows of every image vote (row-sharded images, halo rows
    excluded); ``init(cent)``: custom centroid initialisation (row-sharded images).
    ``debug``: the plan's ``DebugSwitches`` (``force_collectives``, ``no_reverse``).
    ``raster``: (B,H,W) int32 / uint8 device tensor: the last pass writes the label map there itself (whole images only,
    ops that have ``assign_raster``) instead of filling the label slab for a separate raster kernel.
    """
    n_sets = b if mode == "per_image" else 1
    dist = None
    if mode == "global":
        import torch.distributed as td
        if td.is_available() and td.is_initialized() and (td.get_world_size(dist_group) > 1 or debug.force_collectives):
            dist = td
    if init is not None:
        init(cent)
    else:
        ops.kmeans_init(feats, b, h, w, k, n_sets, cent)
        if dist is not None:
            # RCCL / gloo have no 16-bit integer type: move the centroid bytes
            _collective(dist.broadcast, cent.view(_torch().uint8),
                        src=dist.get_global_rank(dist_group, 0) if dist_group is not None else 0,
                        group=dist_group)
    for t in range(n_iter):
        # alternate the sweep direction: a pass starts where the previous one ended (Infinity Cache reuse); the first
        # pass runs back to front because the Gabor stage, which has just written the slab, finished at its end
        # only the last pass's labels are read, and its sums are not: either output of the kernel is optional
        last = t == n_iter - 1
        rev = not (t & 1) and not debug.no_reverse
        if last and raster is not None:
            ops.assign_raster(feats, cent, b, h, w, k, n_sets, raster, scratch_labels=labels, reverse=rev)
            break
        ops.assign_accumulate(feats, cent, b, h, w, k, n_sets, labels if last else None, None if last else partials, rows,
                              reverse=rev)
        if not last:
            if dist is None:
                ops.reduce_finalize(partials, b, h, w, k, n_sets, sums, cent)
            else:
                ops.reduce(partials, b, h, w, k, n_sets, sums)
                _collective(dist.all_reduce, sums, op=dist.ReduceOp.SUM, group=dist_group)
                ops.finalize(sums, n_sets, k, cent)


def halo_rows(n_levels: int = 2, ksize: int = 13) -> int:
    """Real neighbour rows an interior strip edge needs: the reach (ksize - 1) / 2 of the coarsest level's kernel in
    full-resolution rows (12 for the default 13x13 bank on two levels - the defaults here -, 14 for a 15x15 one)."""
    return ((ksize - 1) // 2) << (n_levels - 1)


def shard_rows(height: int, world: int, rank: int, n_levels: int = 2, ksize: int = 13):
    """Row strip of rank ``rank``: owned global rows [r0, r1) and the strip [s0, s1) that also carries real
    neighbour rows on interior edges (BASELINE config 5).

    ``n_levels`` = pyramid levels of the bank (SPEC.md §2), ``ksize`` its kernel size; the defaults are those of the default
    bank (4 scales on 2 levels, 13x13) - for any other bank use ``Segmenter.shard_rows``, which reads them from its bank. Strip and ownership boundaries are multiples of
    ``2**(n_levels-1)`` so that every strip's pyramid is a window of the global pyramid, and the halo is the reach of
    the coarsest level's kernel: ``halo_rows(n_levels, ksize)`` rows."""
    align = 1 << (n_levels - 1)
    halo = halo_rows(n_levels, ksize)
    r0 = (height * rank) // world // align * align
    r1 = height if rank == world - 1 else (height * (rank + 1)) // world // align * align
    return r0, r1, max(0, r0 - halo), min(height, r1 + halo)


def s1_of(bounds, r):
    """End row (exclusive) of rank r's strip, halo included: what its lower neighbour has to send it."""
    return bounds[r][3]


class Segmenter:
    """Reusable plan: bank on device + cached workspaces. ``__call__`` is the slot."""

    def __init__(self, n_scales=4, n_orient=6, k=8, n_iter=10, ksize=13, f_max=0.4,
                 ratio=math.sqrt(2.0), bandwidth=1.0, connectivity=False, device="cuda:0", ops=None,
                 slab_candidates=1):
        if not (1 <= k <= _lib.K_MAX):
            raise ValueError(f"k must be in 1..{_lib.K_MAX}")
        if n_iter < 1:
            raise ValueError("n_iter must be >= 1")
        self.k, self.n_iter = int(k), int(n_iter)
        self.debug = DebugSwitches(os.environ.get("GCS_DEBUG", ""))     # measurement / test switches (see the class)
        self.connectivity = bool(connectivity)     # SPEC.md §7 post-pass
        self.bank = make_bank(n_scales, n_orient, ksize, f_max, ratio, bandwidth)
        self.ops = ops if ops is not None else HipOps(self.bank, device)
        # feature-slab allocations to time at first use of a large workspace shape (see _place_slab). 1 = take the first
        # one (the library default: no extra memory, no host synchronisation, graph-capturable); bench.py asks for 2.
        self.slab_candidates = int(self.debug.slab_candidates or slab_candidates)
        self.slab_placement_ms = None
        self._ws = {}
        self._host = {}
        self._stream = {}
        self._copy_pair = None
        self._stagers = None
        self._graphs = {}

    def __del__(self):
        # a plan dropped in one thread while another thread captures: its graphs leave through the guard (parked until that
        # capture has ended) instead of being destroyed wherever the last reference happened to die
        try:
            ents = list(self._graphs.values())
            self._graphs.clear()
            _CAPTURES.retire(ents)
        except Exception:            # interpreter shutdown: module globals may be gone
            pass

    @property
    def native(self):
        """True when the ops are HipOps (libgcs.so). The only other ops are the CPU tests' stand-in for the host logic
        (tests/fake_ops.py), which has no slabs, streams or graphs: the pipelined / captured host paths need ``native``."""
        return hasattr(self.ops, "lib")

    # ---- workspaces
    def _workspace(self, g, h, w, mode):
        key = (g, h, w, mode)
        ws = self._ws.get(key)
        if ws is None:
            n_sets = g if mode == "per_image" else 1
            ws = dict(feats=self.ops.feature_slab(g, h, w), labels=self.ops.label_slab(g, h, w),
                      partials=self.ops.partial_slab(g, h, w, self.k),
                      cent=self.ops.new_centroids(n_sets, self.k), sums=self.ops.new_sums(n_sets, self.k))
            self._place_slab(ws, g, h, w, n_sets)
            # keep the eight most recent (batch, shape, mode) triples resident (a data set alternating landscape and
            # portrait batches, each with a remainder batch at its end, beside a global-codebook run, would otherwise
            # rebuild - and re-place - its workspace on every switch; 0.95 GB per 64-image entry)
            if len(self._ws) >= 8:
                self._ws.pop(next(iter(self._ws)))
            self._ws[key] = ws
        return ws

    def _place_slab(self, ws, g, h, w, n_sets):
        """Optionally pick, among ``slab_candidates`` allocations, the feature slab the Lloyd pass streams fastest.

        Measured on MI355X (tools/slab_placement.py, profiles/r2_notes.md): the SAME pass kernel on the SAME bytes runs up
        to 6-7 % faster or slower depending on which 0.9 GB allocation holds them (0.169 vs 0.180 ms, stable per allocation
        for the life of the process; the first large allocation of a process is usually a fast one, which points at how
        contiguously the driver could back it - translation reach - rather than at the addresses, which show no pattern).
        With ``slab_candidates`` > 1 that many allocations are timed once with the pass kernel itself and the fastest is
        kept; the others go back to the allocator. Off by default (1): it costs transient memory, synchronises the host
        at first use of a shape and puts extra launches into a profile. bench.py asks for 2 and reports their times."""
        n_cand = self.slab_candidates
        if not self.native or ws["feats"].numel() < (256 << 20) or n_cand <= 1:
            return
        torch = _torch()
        n_cand = min(n_cand, (24 << 30) // max(1, ws["feats"].numel()))      # at most 24 GB of candidates
        cands = [ws["feats"]]
        for _ in range(n_cand - 1):
            try:
                cands.append(self.ops.feature_slab(g, h, w))
            except torch.OutOfMemoryError:       # a job that fits with one slab must not fail here
                break
        if len(cands) < 2:
            return
        best = [float("inf")] * len(cands)
        with torch.cuda.device(self.ops.device):
            for s in cands:
                s.zero_()                                  # defined bytes: the timing must not depend on stale data
            for rnd in range(3):
                for i, s in enumerate(cands):
                    for rev in (False, True):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        self.ops.assign_accumulate(s, ws["cent"], g, h, w, self.k, n_sets, ws["labels"], ws["partials"],
                                                   reverse=rev)
                        e1.record()
                        e1.synchronize()
                        if rnd and rev:
                            best[i] = min(best[i], e0.elapsed_time(e1))
        keep = min(range(len(cands)), key=lambda i: best[i])
        ws["feats"] = cands[keep]
        del cands                                          # the rejected slabs return to torch's caching allocator
        self.slab_placement_ms = best                      # diagnostics (bench.py prints it)

    def group_size(self, b, h, w, mode):
        if mode == "global":
            return b
        per_image = self.ops.lib.gcs_feature_slab_bytes(1, h, w, self.bank.n_scales, self.bank.n_orient) \
            if self.native else 2 * self.bank.n_features * h * w
        return max(1, min(b, _SLAB_BUDGET // max(1, per_image)))

    # ---- device-resident API (used by bench.py: inputs already in HBM)
    def segment_device(self, imgs, mode="per_image", out=None, dist_group=None, group=None):
        """imgs: (B,H,W,3) uint8 device tensor -> (B,H,W) int32 device tensor."""
        torch = _torch()
        if imgs.dtype != torch.uint8 or imgs.dim() != 4 or imgs.shape[3] != 3:
            raise ValueError("imgs must be a (B,H,W,3) uint8 tensor")
        if mode not in ("per_image", "global"):
            raise ValueError("mode must be 'per_image' or 'global'")
        imgs = imgs.contiguous()
        b, h, w, _ = imgs.shape
        if h < 8 or w < 8:
            raise ValueError("images must be at least 8x8")
        on_gpu = self.native
        if on_gpu and imgs.device != self.ops.device:
            raise ValueError(f"imgs live on {imgs.device}, this Segmenter on {self.ops.device}")
        if out is None:
            out = torch.empty((b, h, w), dtype=torch.int32, device=imgs.device)
        elif out.dtype != torch.int32 or tuple(out.shape) != (b, h, w) or out.device != imgs.device \
                or not out.is_contiguous():
            raise ValueError("out must be a contiguous (B,H,W) int32 tensor on the device of imgs")
        g = group or self.group_size(b, h, w, mode)
        import contextlib
        # launches go through ctypes on the current stream of self.ops.device: make that device current
        with (torch.cuda.device(self.ops.device) if on_gpu else contextlib.nullcontext()):
            for g0 in range(0, b, g):
                n = min(g, b - g0)
                ws = self._workspace(n, h, w, mode) if n == g else self._tail_workspace(n, h, w, mode)
                self.ops.gabor_features(imgs[g0:g0 + n], ws["feats"])
                direct = hasattr(self.ops, "assign_raster")      # the last pass writes the raster map itself
                lloyd(self.ops, ws["feats"], n, h, w, self.k, self.n_iter, mode, ws["labels"], ws["partials"],
                      ws["cent"], ws["sums"], dist_group, raster=out[g0:g0 + n] if direct else None, debug=self.debug)
                if not direct:
                    self.ops.labels_widen(ws["labels"], n, h, w, out[g0:g0 + n])
            if self.connectivity:
                regions = torch.empty_like(out)
                self.ops.connected_regions(out, regions)
                out.copy_(regions)
        return out

    def _tail_workspace(self, n, h, w, mode):
        n_sets = n if mode == "per_image" else 1
        return dict(feats=self.ops.feature_slab(n, h, w), labels=self.ops.label_slab(n, h, w),
                    partials=self.ops.partial_slab(n, h, w, self.k),
                    cent=self.ops.new_centroids(n_sets, self.k), sums=self.ops.new_sums(n_sets, self.k))

    def shard_rows(self, height, world, rank):
        """``shard_rows`` for THIS plan's bank (its pyramid depth and kernel size decide alignment and halo)."""
        return shard_rows(height, world, rank, self.bank.n_levels, self.bank.ksize)

    def segment_rows_sharded_device(self, strip, r0, r1, s0, height, dist_group=None, out=None):
        """Row-sharded images with one global codebook (BASELINE config 5).

        ``strip``: (B, s1-s0, W, 3) uint8 device tensor = global rows [s0, s1) of B images of
        ``height`` rows, of which this rank owns [r0, r1) (see ``shard_rows``). The halo rows
        feed the Gabor stage with real neighbour data (outer image borders still reflect, as
        SPEC.md §3 says), are excluded from every k-means sum, and are dropped from the result.
        Returns the (B, r1-r0, W) int32 labels of the owned rows; identical to the same rows of
        the unsharded result. Needs torch.distributed initialised when the image is split.
        """
        torch = _torch()
        strip = strip.contiguous()
        b, hs, w, _ = strip.shape
        if hs < 8 or w < 8:
            raise ValueError("strips must be at least 8x8 (including halo)")
        if not (s0 <= r0 < r1 <= s0 + hs <= height):
            raise ValueError("inconsistent strip geometry")
        align = 1 << (self.bank.n_levels - 1)
        if s0 % align or r0 % align or (r1 % align and r1 != height) or ((s0 + hs) % align and s0 + hs != height):
            raise ValueError(f"strip boundaries must be multiples of {align} rows (pyramid alignment; use shard_rows)")
        halo = halo_rows(self.bank.n_levels, self.bank.ksize)
        if (r0 - s0 < halo and s0 > 0) or (s0 + hs - r1 < halo and s0 + hs < height):
            raise ValueError(f"interior strip edges need {halo} halo rows (use shard_rows)")
        ws = self._tail_workspace(b, hs, w, "global")
        self.ops.gabor_features(strip, ws["feats"])
        k, dfeat = self.k, self.bank.n_features
        import torch.distributed as td
        use_dist = td.is_available() and td.is_initialized() and td.get_world_size(dist_group) > 1

        def init(cent):
            # SPEC.md §4: p_j = floor((2j+1) P / (2k)) over image 0 of the global batch, global coordinates
            p = height * w
            byx = []
            for j in range(k):
                pj = ((2 * j + 1) * p) // (2 * k)
                y, x = pj // w, pj % w
                byx.append((0, y - s0, x) if r0 <= y < r1 else (-1, 0, 0))
            rows_ = self.ops.features_gather(ws["feats"], b, hs, w,
                                             torch.tensor(byx, dtype=torch.int32, device=strip.device))
            table = rows_.view(torch.int16).to(torch.int32) & 0xFFFF          # uint16 values, zero if not owned
            if use_dist:
                _collective(td.all_reduce, table, op=td.ReduceOp.SUM, group=dist_group)
            cent.copy_(table.to(torch.int16).view(1, k, dfeat))               # wraps back to the uint16 bits

        lloyd(self.ops, ws["feats"], b, hs, w, k, self.n_iter, "global", ws["labels"], ws["partials"],
              ws["cent"], ws["sums"], dist_group, rows=(r0 - s0, r1 - s0), init=init, debug=self.debug)
        full = torch.empty((b, hs, w), dtype=torch.int32, device=strip.device)
        self.ops.labels_widen(ws["labels"], b, hs, w, full)
        res = full[:, r0 - s0:r1 - s0]
        if out is not None:
            out.copy_(res)
            return out
        return res.contiguous()

    def segment_owned_rows_device(self, owned, height, dist_group=None):
        """Row-sharded images where every rank holds ONLY its own rows on its device (BASELINE config 5, option (ii) of
        SURVEY §8e): the halo rows the Gabor stage needs are fetched from the neighbouring ranks device to device
        (``torch.distributed`` point-to-point: RCCL send / recv over xGMI on a multi-GPU node; gloo in the tests), then
        the strip runs through ``segment_rows_sharded_device``.

        ``owned``: (B, r1-r0, W, 3) uint8 device tensor = rows [r0, r1) of B images of ``height`` rows, with
        (r0, r1) = ``shard_rows(height, world, rank, n_levels, ksize)[:2]``. Every rank must own at least the halo
        (``halo_rows(n_levels, ksize)`` rows), so that a halo comes from ONE neighbour. Returns the (B, r1-r0, W) int32 labels."""
        torch = _torch()
        import torch.distributed as td
        if not (td.is_available() and td.is_initialized()):
            raise RuntimeError("segment_owned_rows_device needs torch.distributed (one rank per row strip)")
        world, rank = td.get_world_size(dist_group), td.get_rank(dist_group)
        nl, ks = self.bank.n_levels, self.bank.ksize
        r0, r1, s0, s1 = shard_rows(height, world, rank, nl, ks)
        owned = owned.contiguous()
        b, ho, w, _ = owned.shape
        if ho != r1 - r0:
            raise ValueError(f"rank {rank} owns rows [{r0}, {r1}) of {height}: got {ho} rows")
        halo = halo_rows(nl, ks)
        bounds = [shard_rows(height, world, r, nl, ks) for r in range(world)]
        if world > 1 and min(x[1] - x[0] for x in bounds) < halo:
            raise ValueError(f"every rank must own at least {halo} rows for a single-hop halo exchange")
        up, down = rank - 1, rank + 1
        gloo_host = owned.is_cuda and td.get_backend(dist_group) == "gloo"      # gloo moves host memory only

        def wire(t):
            return t.cpu() if gloo_host else t
        ops, recv_top, recv_bot = [], None, None
        peer = (lambda r: td.get_global_rank(dist_group, r)) if dist_group is not None else (lambda r: r)
        if up >= 0:                                          # my first rows are the upper neighbour's bottom halo
            ops.append(td.P2POp(td.isend, wire(owned[:, :s1_of(bounds, up) - r0].contiguous()), peer(up), dist_group))
            recv_top = torch.empty((b, r0 - s0, w, 3), dtype=torch.uint8, device="cpu" if gloo_host else owned.device)
            ops.append(td.P2POp(td.irecv, recv_top, peer(up), dist_group))
        if down < world:                                     # my last rows are the lower neighbour's top halo
            ops.append(td.P2POp(td.isend, wire(owned[:, bounds[down][2] - r0:].contiguous()), peer(down), dist_group))
            recv_bot = torch.empty((b, s1 - r1, w, 3), dtype=torch.uint8, device="cpu" if gloo_host else owned.device)
            ops.append(td.P2POp(td.irecv, recv_bot, peer(down), dist_group))
        if ops:
            for req in td.batch_isend_irecv(ops):
                req.wait()
        parts = ([recv_top.to(owned.device)] if recv_top is not None else []) + [owned] + \
                ([recv_bot.to(owned.device)] if recv_bot is not None else [])
        strip = torch.cat(parts, dim=1) if len(parts) > 1 else owned
        return self.segment_rows_sharded_device(strip, r0, r1, s0, height, dist_group)

    def features_device(self, imgs):
        """Canonical (B,D,H,W) uint16 features as an int16 tensor (tests / debugging)."""
        imgs = imgs.contiguous()
        b, h, w, _ = imgs.shape
        feats = self.ops.feature_slab(b, h, w)
        self.ops.gabor_features(imgs, feats)
        return self.ops.features_unpack(feats, b, h, w)

    # ---- host API: the slot
    def segment_batch(self, imgs: np.ndarray, mode="per_image", out_dtype=np.int32) -> np.ndarray:
        """(B,H,W,3) uint8 host array -> fresh (B,H,W) host label array (the calling convention of script.py:25,30).

        ``out_dtype``: np.int32 (default, SPEC.md §1) or np.uint8 (a quarter of the bytes; metrics.py:43 casts the map
        with ``.astype('int')`` anyway; not with ``connectivity=True``, whose region ids exceed 255).
        The image upload goes through a cached pinned staging buffer, cut into chunks so that the copy of chunk n+1
        (its own stream) overlaps the Gabor stage of chunk n; the labels are copied straight into a fresh pinned
        buffer that the returned array owns."""
        torch = _torch()
        if isinstance(imgs, (list, tuple)):                        # B equally shaped (H,W,3) images: staged one by one
            imgs = _ImageList(imgs)
        else:
            imgs = np.ascontiguousarray(imgs)
        if imgs.dtype != np.uint8 or imgs.ndim != 4 or imgs.shape[3] != 3:
            raise ValueError("imgs must be a (B,H,W,3) uint8 array")
        if mode not in ("per_image", "global"):
            raise ValueError("mode must be 'per_image' or 'global'")
        out_dtype = np.dtype(out_dtype)
        if out_dtype not in (np.dtype(np.int32), np.dtype(np.uint8)):
            raise ValueError("out_dtype must be int32 or uint8")
        if self.connectivity and out_dtype == np.uint8:
            raise ValueError("connectivity=True needs int32 labels")
        b, h, w, _ = imgs.shape
        if h < 8 or w < 8:
            raise ValueError("images must be at least 8x8")
        dist_on = False
        if mode == "global":
            import torch.distributed as td
            dist_on = td.is_available() and td.is_initialized()
        if not self.native or self.connectivity or dist_on or self.debug.force_collectives \
                or self.group_size(b, h, w, mode) < b:
            dev = torch.from_numpy(np.asarray(imgs)).to(self.ops.device)   # plain path (test stand-ins, post-passes, collectives)
            return self.segment_device(dev, mode).cpu().numpy().astype(out_dtype, copy=False)

        ops, dev = self.ops, self.ops.device
        if b * h * w <= _GRAPH_MAX_PIXELS and not self.debug.no_graph:
            return self._segment_small(imgs, mode, out_dtype)
        st = self._host_state(b, h, w)
        ws = self._workspace(b, h, w, mode)
        per_img = ops.lib.gcs_feature_slab_bytes(1, h, w, self.bank.n_scales, self.bank.n_orient)
        cur = torch.cuda.current_stream(dev)
        n_chunks = min(b, 4)
        bounds = [(b * i) // n_chunks for i in range(n_chunks + 1)]
        pin_np = st["pin_in"].numpy()
        with torch.cuda.device(dev):
            st["copy"].wait_stream(cur)                            # the device input buffer is free again
            # Host memcpy into the pinned buffer, chunk by chunk on two worker threads (numpy releases the GIL): one thread
            # stages a 16-image chunk in 0.2 ms, which alone would pace the uploads and the Gabor launches behind them
            # (4 x 0.26 ms before the last chunk's kernels can start, against 0.44 ms of Gabor work).
            if self._stagers is None:
                from concurrent.futures import ThreadPoolExecutor
                self._stagers = ThreadPoolExecutor(2, thread_name_prefix="gcs-stage")
            staged = [self._stagers.submit(_copy_images, pin_np, imgs, bounds[i], bounds[i + 1]) for i in range(n_chunks)]
            for i in range(n_chunks):
                g0, g1 = bounds[i], bounds[i + 1]
                staged[i].result()
                with torch.cuda.stream(st["copy"]):
                    st["dev_in"][g0:g1].copy_(st["pin_in"][g0:g1], non_blocking=True)
                    st["ev"][i].record(st["copy"])
                cur.wait_event(st["ev"][i])
                ops.gabor_features(st["dev_in"][g0:g1], ws["feats"][g0 * per_img:])
            dev_out = st["dev_out"] if out_dtype == np.uint8 else st["dev_out32"]
            lloyd(ops, ws["feats"], b, h, w, self.k, self.n_iter, mode, ws["labels"], ws["partials"],
                  ws["cent"], ws["sums"], raster=dev_out, debug=self.debug)
            # The result is a FRESH pinned host buffer per call, handed to the caller as the base of the returned
            # array (torch's caching host allocator recycles it once the caller drops the array): the device-to-host
            # copy lands directly in caller-owned memory, with no pageable copy and no first-touch page faults.
            res = torch.empty((b, h, w), dtype=dev_out.dtype, pin_memory=True)
            ops.download(res, dev_out)                             # copy engines, as in segment_stream (not the blit kernel)
            cur.synchronize()
        return res.numpy()

    def segment_stream(self, batches, mode="per_image", out_dtype=np.int32, depth=2):
        """Pipelined host API: a generator over the label arrays of an iterable of equally shaped (B,H,W,3) uint8 host
        batches, in order - ``for labels in seg.segment_stream(loader): ...`` instead of calling ``segment_batch`` per batch
        (the loop of script.py:22-38 over a data set). Results equal ``segment_batch(batch, mode, out_dtype)``.

        Three streams work on ``depth + 1`` buffer slots: while batch n runs through the Gabor stage and the Lloyd passes,
        batch n+1 is copied into pinned memory and uploaded and the labels of batch n-1 are downloaded into a fresh pinned
        array that the caller owns, both by the copy engines (gcs_download: see there). A result is handed out ``depth``
        batches after its input was taken (sooner when the input ends)."""
        out_dtype = np.dtype(out_dtype)
        if out_dtype not in (np.dtype(np.int32), np.dtype(np.uint8)):
            raise ValueError("out_dtype must be int32 or uint8")
        if mode not in ("per_image", "global"):
            raise ValueError("mode must be 'per_image' or 'global'")
        dist_on = False
        if mode == "global":
            import torch.distributed as td
            dist_on = td.is_available() and td.is_initialized()
        if not self.native or self.connectivity or dist_on or self.debug.force_collectives:
            for imgs in batches:                                   # no pipeline for post-passes / collectives / stand-ins
                yield self.segment_batch(imgs, mode, out_dtype)
            return
        pipe = _StreamPipe(self, mode, out_dtype, depth)
        try:
            for imgs in batches:
                yield from pipe.push(imgs)
            yield from pipe.drain()
        finally:
            pipe.close()

    def _segment_small(self, imgs, mode, out_dtype):
        """One image (or a small batch): the slot as script.py:22-30 calls it, once per image. A step is ~25 launches of
        kernels that take microseconds each, so the host's launch calls, not the device, set the latency: the whole step
        (Gabor stage, n_iter Lloyd passes, label raster) is captured ONCE per shape as a HIP graph on static buffers and
        replayed per call - upload, one graph launch, download."""
        torch = _torch()
        b, h, w, _ = imgs.shape
        dev = self.ops.device
        key = (b, h, w, mode, out_dtype.str)
        ent = self._graphs.get(key)
        with torch.cuda.device(dev):
            if ent is None:
                ws = self._tail_workspace(b, h, w, mode)
                dev_in = torch.empty((b, h, w, 3), dtype=torch.uint8, device=dev)
                dev_out = torch.empty((b, h, w), dtype=torch.uint8 if out_dtype == np.uint8 else torch.int32, device=dev)
                pin_in = torch.empty((b, h, w, 3), dtype=torch.uint8, pin_memory=True)
                # the graph bakes raw pointers in: every buffer it touches, the Gabor scratch included, belongs to the
                # entry and lives exactly as long as the graph does (the shared scratch of HipOps is replaced, and its
                # block recycled, whenever a later call needs a larger one)
                scratch = self.ops.gabor_scratch(b, h, w)

                # (the closure must not capture `self`: the entry lives in self._graphs, and a Segmenter inside a reference
                # cycle is freed - with its graphs, streams, pinned and device buffers - only when the cyclic collector
                # happens to run, not when the last user drops it)
                ops, k, n_iter, debug = self.ops, self.k, self.n_iter, self.debug

                def step():
                    ops.gabor_features(dev_in, ws["feats"], scratch=scratch)
                    lloyd(ops, ws["feats"], b, h, w, k, n_iter, mode, ws["labels"], ws["partials"], ws["cent"],
                          ws["sums"], raster=dev_out, debug=debug)
                dev_in.zero_()
                step()                                     # eager once: first-use work (side-stream creation) outside the capture
                # (a STREAM wait - the step joins its side stream back into this one: a device-wide synchronize is refused while
                # any thread of the process captures, and invalidates that thread's capture: _CaptureGuard)
                torch.cuda.current_stream(dev).synchronize()
                graph = torch.cuda.CUDAGraph()
                # capture under the module's guard (_CaptureGuard: serialised, collector off for exactly that long, retired
                # graphs parked meanwhile); a refused capture (e.g. the caller's own capture is open on this thread) is reported
                # once and the shape then runs as eager launches
                if not _CAPTURES.capture(torch, graph, step, dev):
                    _CAPTURES.fell_back(key, "torch.cuda.graph raised RuntimeError")
                    graph = None
                    torch.cuda.current_stream(dev).synchronize()
                ent = dict(graph=graph, step=step, ws=ws, dev_in=dev_in, dev_out=dev_out, pin_in=pin_in, scratch=scratch)
                if len(self._graphs) >= 4:                 # evicted plans leave through the guard: never inside an open capture
                    _CAPTURES.retire([self._graphs.pop(next(iter(self._graphs)))])
                self._graphs[key] = ent
            cur = torch.cuda.current_stream(dev)
            _stage(ent["pin_in"], imgs)
            ent["dev_in"].copy_(ent["pin_in"], non_blocking=True)
            if ent["graph"] is not None:
                ent["graph"].replay()
            else:
                ent["step"]()
            res = torch.empty((b, h, w), dtype=ent["dev_out"].dtype, pin_memory=True)   # caller-owned pinned result
            res.copy_(ent["dev_out"], non_blocking=True)
            cur.synchronize()
        return res.numpy()

    def _copy_streams(self):
        """The upload and the download stream of this Segmenter, created once and shared by every host path and shape. HIP maps
        a process's streams onto a few hardware queues (four by default): with a pair of streams per shape and path, a
        download stream ended up on the compute stream's queue and every label download then held back the next batch's
        kernels (the pipelined rate fell from 4 360 to 3 300 Mpix/s as soon as segment_batch had created its own copy stream;
        profiles/r3_notes.md). Default stream + the Gabor side stream + these two = four."""
        if self._copy_pair is None:
            torch = _torch()
            dev = self.ops.device
            self._copy_pair = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        return self._copy_pair

    def _host_state(self, b, h, w):
        """Pinned staging buffers, device input / uint8 output buffers, copy stream and events for one batch shape."""
        torch = _torch()
        key = (b, h, w)
        st = self._host.get(key)
        if st is None:
            dev = self.ops.device
            st = dict(pin_in=torch.empty((b, h, w, 3), dtype=torch.uint8, pin_memory=True),
                      dev_in=torch.empty((b, h, w, 3), dtype=torch.uint8, device=dev),
                      dev_out=torch.empty((b, h, w), dtype=torch.uint8, device=dev),
                      dev_out32=torch.empty((b, h, w), dtype=torch.int32, device=dev),
                      copy=self._copy_streams()[0], ev=[torch.cuda.Event() for _ in range(4)])
            if len(self._host) >= 4:                               # keep the four most recent shapes (see _workspace)
                self._host.pop(next(iter(self._host)))
            self._host[key] = st
        return st

    def __call__(self, img: np.ndarray) -> np.ndarray:
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("img must be an (H,W,3) uint8 array (skimage.io.imread of an RGB file)")
        return self.segment_batch(img[None])[0]

    def segment_images(self, images, batch=64, out_dtype=np.int32):
        """The loop of script.py:22-38 as a generator: ``for labels in seg.segment_images(loader): ...`` yields, in input
        order, exactly what ``segment(img)`` returns for every (H,W,3) uint8 image of an iterable - any mix of shapes (BSD500
        holds 481x321 and 321x481 images). Images wait in per-shape groups of up to ``batch`` and go through the device
        together (per-image codebooks, so grouping does not change any label): the data-set loop at the batch rate instead of
        one 0.28 ms call per image. Held back at any time: up to two batches per shape in flight (one computing, one
        downloading), the partial group of every shape, and at most ``2 * batch`` finished label maps that wait for an
        EARLIER image of another shape - when that many have piled up, the group (or the pipeline) that holds the oldest
        outstanding image is flushed early as a partial batch, so a rare shape at the start of a long or endless loader
        neither stalls the output nor grows the pinned result memory without bound."""
        batch = int(batch)
        if batch < 1:
            raise ValueError("batch must be >= 1")
        out_dtype = np.dtype(out_dtype)
        if out_dtype not in (np.dtype(np.int32), np.dtype(np.uint8)):
            raise ValueError("out_dtype must be int32 or uint8")
        groups, ready, nxt = {}, {}, 0
        pipes, tags = {}, {}                   # per shape: the pipeline of its FULL batches and the image indices in flight
        piped = self.native and not self.connectivity and not self.debug.force_collectives

        def emit(idx, labels):
            for i, lab in zip(idx, labels):
                ready[i] = lab

        def flush(shape, final):
            idx, imgs = zip(*groups.pop(shape))
            h, w = shape[:2]
            if piped and not final and len(idx) * h * w > _GRAPH_MAX_PIXELS \
                    and self.group_size(len(idx), h, w, "per_image") >= len(idx):
                # full batches of one shape follow each other through the three-stream pipeline (upload, compute and
                # download of consecutive batches overlap); their label maps fall due one push later
                if shape not in pipes:
                    pipes[shape], tags[shape] = _StreamPipe(self, "per_image", out_dtype, 1), []
                tags[shape].append(idx)
                for labels in pipes[shape].push(_ImageList(imgs)):
                    emit(tags[shape].pop(0), labels)
            else:
                emit(idx, self.segment_batch(list(imgs), "per_image", out_dtype))

        try:
            for n, img in enumerate(images):
                img = np.asarray(img)
                if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
                    raise ValueError("every image must be an (H,W,3) uint8 array (skimage.io.imread of an RGB file)")
                groups.setdefault(img.shape, []).append((n, img))
                if len(groups[img.shape]) >= batch:
                    flush(img.shape, False)
                if len(ready) >= 2 * batch and nxt not in ready:
                    # bound the wait: the oldest outstanding image sits in a partial group or in a pipeline
                    held = [sh for sh, g in groups.items() if g and g[0][0] == nxt]
                    if held:
                        flush(held[0], True)
                    else:
                        for shape, pipe in pipes.items():
                            if tags[shape] and tags[shape][0][0] == nxt:
                                for labels in pipe.drain():
                                    emit(tags[shape].pop(0), labels)
                while nxt in ready:
                    yield ready.pop(nxt)
                    nxt += 1
            for shape, pipe in pipes.items():                      # the input has ended
                for labels in pipe.drain():
                    emit(tags[shape].pop(0), labels)
            while groups:                                          # remainders: oldest waiting image first
                flush(min(groups, key=lambda sh: groups[sh][0][0]), True)
            while nxt in ready:
                yield ready.pop(nxt)
                nxt += 1
        finally:
            for pipe in pipes.values():
                pipe.close()


class _StreamPipe:
    """The three-stream pipeline behind Segmenter.segment_stream / segment_images for ONE batch shape: ``push(batch)`` stages
    and uploads the batch, queues its Gabor stage, Lloyd passes and label download, and returns the label arrays that have
    become due (``depth`` pushes later, in order); ``drain()`` returns the rest.

    ``depth + 1`` buffer slots: while batch n computes, batch n+1 is copied into pinned memory and uploaded and the labels of
    batch n-1 are downloaded into a fresh pinned array that the caller owns, both by the copy engines. The download is not a
    ``tensor.copy_`` / hipMemcpyAsync: on this stack that device-to-host copy runs as a blit KERNEL (`__amd_rocclr_copyBuffer`
    in the kernel trace) and, for as long as PCIe takes - 0.75 ms for the 39.5 MB of a 64-image int32 batch - the kernels of
    every other stream crawl (profiles/r3_notes.md); gcs_download hands the same bytes to SDMA, like the uploads."""

    def __init__(self, seg, mode, out_dtype, depth):
        self.seg, self.mode, self.depth = seg, mode, int(depth)
        self.n_slots = max(2, self.depth + 1)
        torch = _torch()
        self.t_dtype = torch.uint8 if np.dtype(out_dtype) == np.uint8 else torch.int32
        self.st = self.ws = self.shape = None
        self.n = 0
        self.pending = []                                          # (download event, result array) in input order

    def _open(self, b, h, w):
        torch = _torch()
        seg, dev, n_slots = self.seg, self.seg.ops.device, self.n_slots
        if h < 8 or w < 8:
            raise ValueError("images must be at least 8x8")
        if seg.group_size(b, h, w, self.mode) < b:
            raise ValueError("batch too large for one feature slab: use smaller batches")
        self.shape = (b, h, w)
        self.ws = seg._workspace(b, h, w, self.mode)
        key = (b, h, w, n_slots, self.t_dtype)
        st = seg._stream.get(key)
        if st is None or st["busy"]:           # pinning 3 x 30 MB costs ms: the state is kept for the next stream of this shape
            st = dict(pin_in=[torch.empty((b, h, w, 3), dtype=torch.uint8, pin_memory=True) for _ in range(n_slots)],
                      dev_in=[torch.empty((b, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(n_slots)],
                      dev_out=[torch.empty((b, h, w), dtype=self.t_dtype, device=dev) for _ in range(n_slots)],
                      up=seg._copy_streams()[0], down=seg._copy_streams()[1],
                      ev_up=[torch.cuda.Event() for _ in range(n_slots)],
                      ev_done=[torch.cuda.Event() for _ in range(n_slots)],
                      ev_down=[torch.cuda.Event() for _ in range(n_slots)], busy=False)
            if key not in seg._stream:                             # (a second live stream of the same shape stays private)
                if len(seg._stream) >= 4:
                    idle = [k for k, v in seg._stream.items() if not v["busy"]]
                    if idle:
                        seg._stream.pop(idle[0])
                if len(seg._stream) < 4:
                    seg._stream[key] = st
        else:                                                      # a stream abandoned half way may still own the slots
            st["up"].synchronize()
            st["down"].synchronize()
            torch.cuda.current_stream(dev).synchronize()
        st["busy"] = True
        self.st = st

    def push(self, imgs):
        torch = _torch()
        seg, ops, dev = self.seg, self.seg.ops, self.seg.ops.device
        if not isinstance(imgs, _ImageList):
            imgs = np.ascontiguousarray(imgs)
        if imgs.dtype != np.uint8 or imgs.ndim != 4 or imgs.shape[3] != 3:
            raise ValueError("every batch must be a (B,H,W,3) uint8 array")
        b, h, w, _ = imgs.shape
        with torch.cuda.device(dev):
            if self.st is None:
                self._open(b, h, w)
            elif (b, h, w) != self.shape:
                raise ValueError(f"batch {self.n} has shape {(b, h, w)}, the stream was opened with {self.shape}")
            st, ws, n, n_slots = self.st, self.ws, self.n, self.n_slots
            cur = torch.cuda.current_stream(dev)
            i = n % n_slots
            if n >= n_slots:
                st["ev_up"][i].synchronize()                       # the slot's previous upload has left the pinned buffer
            _stage(st["pin_in"][i], imgs)                          # host memcpy, while the device works on earlier batches
            with torch.cuda.stream(st["up"]):
                if n >= n_slots:
                    st["up"].wait_event(st["ev_done"][i])          # dev_in[i] is no longer being read
                st["dev_in"][i].copy_(st["pin_in"][i], non_blocking=True)
                st["ev_up"][i].record(st["up"])
            cur.wait_event(st["ev_up"][i])
            if n >= n_slots:
                cur.wait_event(st["ev_down"][i])                   # dev_out[i] has been downloaded
            ops.gabor_features(st["dev_in"][i], ws["feats"])
            lloyd(ops, ws["feats"], b, h, w, seg.k, seg.n_iter, self.mode, ws["labels"], ws["partials"], ws["cent"],
                  ws["sums"], raster=st["dev_out"][i], debug=seg.debug)
            st["ev_done"][i].record(cur)
            land = torch.empty((b, h, w), dtype=self.t_dtype, pin_memory=True)   # caller-owned; torch recycles it once dropped
            with torch.cuda.stream(st["down"]):
                st["down"].wait_event(st["ev_done"][i])
                ops.download(land, st["dev_out"][i])
                st["ev_down"][i].record(st["down"])
            ev = torch.cuda.Event()
            ev.record(st["down"])
            self.pending.append((ev, land.numpy()))
            self.n += 1
            out = []
            while len(self.pending) > self.depth:
                ev0, res0 = self.pending.pop(0)
                ev0.synchronize()
                out.append(res0)
            return out

    def drain(self):
        out = []
        for ev0, res0 in self.pending:
            ev0.synchronize()
            out.append(res0)
        self.pending = []
        return out

    def close(self):
        """Leave the compute stream ordered behind the copy streams and hand the slot buffers back for reuse."""
        if self.st is not None:
            torch = _torch()
            cur = torch.cuda.current_stream(self.seg.ops.device)
            cur.wait_stream(self.st["up"])
            cur.wait_stream(self.st["down"])
            # Results nobody took (an abandoned or failed generator): their pinned landing buffers come from torch's host
            # allocator, which knows nothing about the raw gcs_download still writing them - wait for those copies before
            # the arrays are dropped, or the next pinned allocation could be handed a block with a transfer in flight.
            for ev0, _ in self.pending:
                ev0.synchronize()
            self.pending = []
            self.st["busy"] = False
            self.st = None


class _ImageList:
    """A list of equally shaped (H,W,3) uint8 images seen as a (B,H,W,3) batch without stacking them first."""
    ndim, dtype = 4, np.dtype(np.uint8)

    def __init__(self, items):
        self.items = [np.ascontiguousarray(x) for x in items]
        if not self.items:
            raise ValueError("empty batch")
        first = self.items[0]
        if any(x.dtype != np.uint8 or x.shape != first.shape for x in self.items) or first.ndim != 3:
            raise ValueError("a list batch must hold equally shaped (H,W,3) uint8 images")
        self.shape = (len(self.items),) + first.shape

    def __array__(self, dtype=None, copy=None):
        return np.stack(self.items)


def _copy_images(dst, imgs, g0, g1):
    """dst[g0:g1] = imgs[g0:g1] for an array batch or an _ImageList."""
    if isinstance(imgs, _ImageList):
        for i in range(g0, g1):
            np.copyto(dst[i], imgs.items[i])
    else:
        np.copyto(dst[g0:g1], imgs[g0:g1])


def _stage(pinned, imgs):
    """Host array -> pinned staging tensor with ONE plain memcpy (numpy, GIL released). Not `tensor.copy_`: that runs on
    torch's intra-op pool, which sizes itself by the machine's core count and not by the process's CPU quota - on a
    256-core host inside a 16-core cgroup the pool's threads get throttled and a 30 MB copy takes 10+ ms instead of 0.5."""
    _copy_images(pinned.numpy(), imgs, 0, imgs.shape[0])


_default: dict = {}


def _plan(kw) -> Segmenter:
    key = tuple(sorted(kw.items()))
    if key not in _default:
        _default.clear()
        _default[key] = Segmenter(**kw)
    return _default[key]


def segment(img, **kw) -> np.ndarray:
    """Drop-in for ``slic(img, ...)`` at script.py:30: (H,W,3) uint8 -> (H,W) int32 labels 0..k-1."""
    return _plan(kw)(img)


def segment_images(images, batch=64, **kw):
    """Generator form of the loop at script.py:22-38: yields ``segment(img)`` for every image of an iterable, in order, with
    the images batched per shape behind the scenes (``Segmenter.segment_images``)."""
    out_dtype = kw.pop("out_dtype", np.int32)
    return _plan(kw).segment_images(images, batch=batch, out_dtype=out_dtype)


def segment_batch(imgs, mode="per_image", **kw) -> np.ndarray:
    """(B,H,W,3) uint8 -> (B,H,W) int32; equals stack([segment(i) for i in imgs]) in per_image mode."""
    out_dtype = kw.pop("out_dtype", np.int32)
    return _plan(kw).segment_batch(imgs, mode, out_dtype=out_dtype)
