"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
import os
import numpy as np
import pytest

from oracle import spec_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _synth(b, h, w, seed):
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    return synthetic_batch(b, h, w, seed=seed)


@pytest.mark.parametrize("h,w", [(40, 56), (72, 104), (33, 131), (64, 64), (9, 8)])
def test_features_bit_exact_default_bank(torch_cuda, h, w):
    from gabor_color_image_segmentation_amd import Segmenter
    torch = torch_cuda
    imgs = _synth(2, h, w, seed=11)
    seg = Segmenter()
    got = seg.features_device(torch.from_numpy(imgs).cuda()).cpu().numpy().view(np.uint16)
    tapq, shift = so.bank()
    for b in range(2):
        ref = so.gabor_features(imgs[b], tapq, shift, 6)
        assert got[b].shape == ref.shape
        bad = np.argwhere(got[b] != ref)
        assert bad.size == 0, f"{len(bad)} mismatches, first {bad[:5]}, got {got[b][tuple(bad[0])]} ref {ref[tuple(bad[0])]}"


@pytest.mark.parametrize("ns,no,ks", [(1, 1, 15), (2, 3, 7), (2, 5, 11), (1, 2, 1), (3, 4, 13)])
def test_features_bit_exact_other_banks(torch_cuda, ns, no, ks):
    from gabor_color_image_segmentation_amd import Segmenter
    torch = torch_cuda
    imgs = _synth(1, 50, 70, seed=5)
    seg = Segmenter(n_scales=ns, n_orient=no, ksize=ks)
    got = seg.features_device(torch.from_numpy(imgs).cuda()).cpu().numpy().view(np.uint16)
    tapq, shift = so.bank(ns, no, ks)
    ref = so.gabor_features(imgs[0], tapq, shift, no)
    assert np.array_equal(got[0], ref)


def test_features_extreme_pixels(torch_cuda):
    """All-0, all-255 and checkerboard images exercise the int8 offset / bias path."""
    from gabor_color_image_segmentation_amd import Segmenter
    torch = torch_cuda
    h, w = 40, 72
    imgs = np.zeros((3, h, w, 3), np.uint8)
    imgs[1] = 255
    imgs[2] = ((np.add.outer(np.arange(h), np.arange(w)) & 1) * 255).astype(np.uint8)[..., None]
    seg = Segmenter()
    got = seg.features_device(torch.from_numpy(imgs).cuda()).cpu().numpy().view(np.uint16)
    tapq, shift = so.bank()
    for b in range(3):
        assert np.array_equal(got[b], so.gabor_features(imgs[b], tapq, shift, 6))


@pytest.mark.parametrize("h,w", [(321, 481), (481, 321), (33, 41), (34, 42), (33, 40), (40, 41), (41, 58), (42, 57), (9, 10),
                                 (17, 8), (8, 18), (161, 241), (65, 130), (35, 43)])
@pytest.mark.parametrize("ns", [1, 2, 4])
def test_packed_edge_strips_features_and_labels(torch_cuda, h, w, ns):
    """Round-4 slab layout (csrc/common.h): edges of one or two pixel columns / rows (both sides of every BSD image are
    8k + 1) live in virtual blocks of 16 level-1 parents behind the main 8x8 blocks, for banks of one and two pyramid
    levels. Every combination of packed / unpacked right and bottom edges, both BSD orientations, strips shorter than one
    virtual block and strips that end inside one: features (through gcs_features_unpack) and labels (both codebook modes,
    int32 and uint8 raster maps, a row window) equal the C oracle's, bit for bit."""
    from gabor_color_image_segmentation_amd import Segmenter
    from oracle import c_oracle as co
    torch = torch_cuda
    b = 2 if h * w > 50000 else 3
    imgs = _synth(b, h, w, seed=100 + h + w)
    seg = Segmenter(n_scales=ns, n_orient=3 if ns != 4 else 6, k=5, n_iter=3)
    bank = seg.bank
    d_imgs = torch.from_numpy(imgs).cuda()
    got = seg.features_device(d_imgs).cpu().numpy().view(np.uint16)
    for i in range(b):
        ref = co.gabor_features(imgs[i], bank.tapq, bank.shift, bank.n_orient)
        bad = np.argwhere(got[i] != ref)
        assert bad.size == 0, f"image {i}: {len(bad)} feature mismatches, first (d, y, x) {bad[:4].tolist()}"
    for mode in ("per_image", "global"):
        want = co.segment_batch(imgs, bank.tapq, bank.shift, bank.n_orient, k=5, n_iter=3, mode=mode)
        lab = seg.segment_device(d_imgs, mode=mode).cpu().numpy()
        assert np.array_equal(lab, want), (mode, np.argwhere(lab != want)[:4].tolist())
    assert np.array_equal(seg.segment_batch(imgs, out_dtype=np.uint8), co.segment_batch(
        imgs, bank.tapq, bank.shift, bank.n_orient, k=5, n_iter=3, mode="per_image"))
    # a Lloyd pass with a row window writes a uint8 label for EVERY pixel (rows outside the window do not vote)
    ws = seg._tail_workspace(b, h, w, "global")
    seg.ops.gabor_features(d_imgs, ws["feats"])
    seg.ops.kmeans_init(ws["feats"], b, h, w, 5, 1, ws["cent"])
    lo_, hi_ = h // 4, h - h // 5
    seg.ops.assign_accumulate(ws["feats"], ws["cent"], b, h, w, 5, 1, ws["labels"], ws["partials"], rows=(lo_, hi_))
    seg.ops.reduce(ws["partials"], b, h, w, 5, 1, ws["sums"])
    lab8 = ws["labels"][:b * h * w].view(b, h, w).cpu().numpy()
    cent = ws["cent"].cpu().numpy().view(np.uint16)[0].astype(np.int64)
    feats = np.stack([co.gabor_features(im, bank.tapq, bank.shift, bank.n_orient) for im in imgs]).astype(np.int64)
    x = feats.reshape(b, feats.shape[1], -1).transpose(0, 2, 1)                         # (b, P, D)
    ref_lab = np.stack([so.kmeans_assign(x[i], cent) for i in range(b)]).reshape(b, h, w)
    assert np.array_equal(lab8, ref_lab)
    sums = ws["sums"].cpu().numpy()[0]
    vote = np.zeros((h, w), bool)
    vote[lo_:hi_] = True
    for j in range(5):
        m = (ref_lab == j) & vote[None]
        assert sums[j, -1] == m.sum()
        assert np.array_equal(sums[j, :-1], feats.transpose(0, 2, 3, 1)[m].sum(axis=0))


@pytest.mark.parametrize("k,n_iter", [(8, 10), (3, 4), (16, 3), (1, 2), (5, 1)])
def test_segment_labels_bit_exact(torch_cuda, k, n_iter):
    from gabor_color_image_segmentation_amd import Segmenter
    imgs = _synth(3, 48, 80, seed=21)
    seg = Segmenter(k=k, n_iter=n_iter)
    got = seg.segment_batch(imgs)
    for b in range(3):
        ref = so.segment(imgs[b], k=k, n_iter=n_iter)
        assert got[b].dtype == np.int32 and got[b].shape == ref.shape
        assert np.array_equal(got[b], ref), f"image {b}: {(got[b] != ref).mean():.4%} mismatch"


def test_segment_single_image_matches_batch(torch_cuda):
    from gabor_color_image_segmentation_amd import segment, segment_batch
    imgs = _synth(3, 40, 64, seed=2)
    batch = segment_batch(imgs, n_iter=3)
    for b in range(3):
        assert np.array_equal(segment(imgs[b], n_iter=3), batch[b])


def test_small_host_calls_replay_a_captured_graph_and_equal_the_eager_path(torch_cuda):
    """segment(img) / small segment_batch calls replay one HIP graph per shape (segmenter._segment_small): different images,
    both label dtypes, two alternating shapes and both codebook modes give exactly what the eager device path gives."""
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    seg = Segmenter(n_iter=3)
    for rnd in range(2):
        for (b, h, w), seed in (((1, 321, 481), 5), ((1, 481, 321), 6), ((3, 64, 88), 7)):
            imgs = _synth(b, h, w, seed=seed + 10 * rnd)
            for mode in ("per_image", "global"):
                want = seg.segment_device(torch.from_numpy(imgs).cuda(), mode=mode).cpu().numpy()
                got = seg.segment_batch(imgs, mode=mode)
                assert got.dtype == np.int32 and np.array_equal(got, want), (rnd, b, h, w, mode)
            got8 = seg.segment_batch(imgs, out_dtype=np.uint8)
            assert got8.dtype == np.uint8 and np.array_equal(got8, seg.segment_device(torch.from_numpy(imgs).cuda()).cpu().numpy())
    assert len(seg._graphs) >= 2                               # the graphs were used (and the cache stays bounded)
    assert len(seg._graphs) <= 4


def test_a_replayed_graph_never_writes_through_a_recycled_scratch_block(torch_cuda):
    """ADVICE r3 (use-after-free): the one-image graph used to bake in the pointer of HipOps' SHARED Gabor scratch; a later,
    larger call replaced that scratch, its block went back to torch's allocator, and the next replay wrote the padded
    pyramid planes into whoever owned the block by then. Every graph entry now owns its scratch: after a large batch has
    grown the shared scratch, fill every block the allocator may hand out, replay the small graph, and nothing moved."""
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    seg = Segmenter(n_iter=2)
    one = _synth(1, 96, 128, seed=31)
    want = seg.segment_batch(one)                             # captures the (1, 96, 128) graph
    shared_before = seg.ops._gabor_ws
    big = _synth(6, 200, 300, seed=32)
    seg.segment_device(torch.from_numpy(big).cuda())          # grows (replaces) the shared scratch
    assert seg.ops._gabor_ws is not shared_before or shared_before is None
    del shared_before
    ent = next(iter(seg._graphs.values()))
    assert ent["scratch"].data_ptr() != seg.ops._gabor_ws.data_ptr()
    # occupy whatever the caching allocator has free (the old shared scratch among it) with a known pattern
    canaries = [torch.full((n,), 0x5A, dtype=torch.uint8, device="cuda") for n in (1 << 16, 1 << 18, 1 << 20, 1 << 21, 1 << 22)]
    torch.cuda.synchronize()
    again = seg.segment_batch(one)                            # replay
    torch.cuda.synchronize()
    assert np.array_equal(again, want)
    for c in canaries:
        assert bool((c == 0x5A).all())


def test_a_dropped_segmenter_is_released_at_once(torch_cuda):
    """A plan owns device slabs, pinned buffers, streams, worker threads and captured graphs: dropping the last reference
    must release them by reference counting, not whenever the cyclic collector next runs (the graph entries used to hold a
    closure over the Segmenter: a cycle; a test session then piled up dozens of dead plans)."""
    import gc
    import weakref
    from gabor_color_image_segmentation_amd import Segmenter
    seg = Segmenter(n_iter=2)
    small, batch = _synth(1, 40, 64, seed=3), _synth(20, 250, 330, seed=4)
    seg.segment_batch(small)                                   # graph path
    seg.segment_batch(batch)                                   # chunked upload path
    assert len(list(seg.segment_stream([batch, batch]))) == 2  # three-stream pipeline
    assert len(list(seg.segment_images(list(small) * 3, batch=2))) == 3
    gc.collect()
    gc.disable()
    try:
        ref = weakref.ref(seg)
        del seg
        assert ref() is None, gc.get_referrers(ref())
    finally:
        gc.enable()


def test_capturing_a_plan_survives_dead_graph_cycles(torch_cuda):
    """Round 4's session abort, named: a torch CUDAGraph finalised by the cyclic collector WHILE another capture is open throws
    out of ~CUDAGraph (hipErrorStreamCaptureUnsupported) and std::terminate kills the process; torch 2.10 no longer collects
    before a capture. _segment_small collects first and keeps the collector off while it captures: unreachable cycles that
    hold captured graphs, and a collector set to run at every allocation, must not matter."""
    import gc
    import torch
    from gabor_color_image_segmentation_amd import Segmenter

    def dead_cycle_with_a_graph():
        g = torch.cuda.CUDAGraph()
        x = torch.zeros(64, device="cuda")
        with torch.cuda.graph(g):
            x += 1
        holder = {"graph": g, "x": x}
        holder["self"] = holder                                # unreachable after return, but only the cyclic GC frees it

    gc.collect()
    gc.disable()
    try:
        for _ in range(3):
            dead_cycle_with_a_graph()
    finally:
        gc.enable()
    old = gc.get_threshold()
    gc.set_threshold(1, 1, 1)                                  # a collection at (nearly) every container allocation
    try:
        seg = Segmenter(n_iter=2)
        img = _synth(1, 56, 88, seed=77)
        got = seg.segment_batch(img)                           # eager step, then the capture
    finally:
        gc.set_threshold(*old)
    assert np.array_equal(got[0], so.segment(img[0], n_iter=2))
    assert np.array_equal(seg.segment_batch(img), got)         # replay


def test_two_threads_capture_their_first_plans_at_the_same_time(torch_cuda):
    """the collect / gc.disable() / capture / gc.enable() sequence is process-global state. Two Segmenters
    whose FIRST one-image calls (eager step + capture) overlap on two threads, with dead graph cycles lying around, the
    collector set to run at every allocation, and plans being dropped meanwhile: captures are serialised by the module's guard,
    the collector comes back on only after the capture that turned it off, retired graphs wait for the open capture to end.
    Results == the oracle, the collector is enabled afterwards, nothing aborts."""
    import gc
    import threading
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd import segmenter as sg

    def dead_cycle_with_a_graph():
        g = torch.cuda.CUDAGraph()
        x = torch.zeros(64, device="cuda")
        with torch.cuda.graph(g):
            x += 1
        holder = {"graph": g, "x": x}
        holder["self"] = holder

    gc.collect()
    gc.disable()
    try:
        for _ in range(4):
            dead_cycle_with_a_graph()
    finally:
        gc.enable()
    imgs = {0: _synth(1, 56, 88, seed=81), 1: _synth(1, 64, 72, seed=82)}
    want = {i: so.segment(imgs[i][0], n_iter=2) for i in imgs}
    start = threading.Barrier(2)
    got, errs = {}, []

    def run(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                start.wait()
                for rep in range(3):                            # every round: a fresh plan (eager step + capture), then a replay
                    seg = Segmenter(n_iter=2)
                    a = seg.segment_batch(imgs[i])
                    b = seg.segment_batch(imgs[i])
                    assert np.array_equal(a, b)
                    got[i] = a
                    del seg                                     # dropped while the other thread may be capturing
        except BaseException as e:  # noqa: B036
            errs.append((i, repr(e)))

    old = gc.get_threshold()
    gc.set_threshold(1, 1, 1)
    try:
        threads = [threading.Thread(target=run, args=(i,)) for i in (0, 1)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(300)
    finally:
        gc.set_threshold(*old)
    assert not errs, errs
    assert gc.isenabled()                                       # re-enabled by the capturer that disabled it, and only then
    assert not sg._CAPTURES._parked or sg._CAPTURES.retire([]) is None
    for i in imgs:
        assert np.array_equal(got[i][0], want[i])


def test_a_refused_capture_warns_once_and_runs_eagerly(torch_cuda):
    """A capture that cannot be taken (here: the caller's own capture is open on this thread) used to degrade the shape to eager
    launches silently; now it says so, once per plan key, and the result is the same."""
    import warnings
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd import segmenter as sg
    seg = Segmenter(n_iter=2)
    img = _synth(1, 48, 80, seed=83)
    real = sg._CAPTURES.capture
    sg._CAPTURES.capture = lambda torch_, graph, body, device=None: False    # what a RuntimeError out of torch.cuda.graph turns into
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = seg.segment_batch(img)
            again = seg.segment_batch(img)
    finally:
        sg._CAPTURES.capture = real
    assert sum("graph capture refused" in str(x.message) for x in w) == 1
    assert np.array_equal(got[0], so.segment(img[0], n_iter=2)) and np.array_equal(got, again)


def test_a_capture_invalidated_by_another_threads_device_synchronize(torch_cuda):
    """The REAL failure path of a first call (the monkey-patched form is the test above): while thread A
    captures a new shape's graph, thread B - standing for a caller's loader or logger - calls torch.cuda.synchronize() in a loop,
    which HIP refuses during a capture and which invalidates that capture. tests/checkers/invalidated_capture_child.py checks
    labels == oracle for the first and a second call, at most one warning, a later shape, and a normal process end; it runs in a
    process of its own (two processes on the card) so that an abort would fail this test, not end the session."""
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "checkers", "invalidated_capture_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"child ended with {r.returncode}:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    assert r.stdout.strip().splitlines()[-1].startswith("OK capture"), r.stdout[-2000:]
    print(r.stdout.strip().splitlines()[-1])


def test_segment_stream_equals_segment_batch(torch_cuda):
    """The pipelined host API (three streams, depth + 1 buffer slots): seven batches through segment_stream give, in order,
    exactly what segment_batch gives for each - both label dtypes, both codebook modes, a depth larger than the input."""
    from gabor_color_image_segmentation_amd import Segmenter
    seg = Segmenter(n_iter=3)
    batches = [_synth(6, 72, 104, seed=30 + i) for i in range(7)]
    for mode, dt, depth in (("per_image", np.int32, 2), ("global", np.uint8, 1), ("per_image", np.uint8, 9)):
        want = [seg.segment_batch(x, mode=mode, out_dtype=dt) for x in batches]
        got = list(seg.segment_stream(iter(batches), mode=mode, out_dtype=dt, depth=depth))
        assert len(got) == len(want)
        for g, w_ in zip(got, want):
            assert g.dtype == dt and np.array_equal(g, w_)
    assert list(seg.segment_stream(iter([]))) == []
    with pytest.raises(ValueError):
        list(seg.segment_stream(iter([batches[0], _synth(6, 64, 104, seed=1)])))
    # a stream abandoned half way (its slots still hold work in flight), then the same shape again: the kept slot buffers are
    # drained before reuse; every result is the caller's own pinned array, so results that are KEPT stay valid while later
    # batches reuse the slots
    want = [seg.segment_batch(x) for x in batches]
    gen = seg.segment_stream(iter(batches), depth=2)
    first = next(gen)
    del gen
    assert np.array_equal(first, want[0])
    got = list(seg.segment_stream(iter(batches * 2), depth=1))
    assert len(got) == 14 and all(g.dtype == np.int32 for g in got)
    for g, w_ in zip(got, want * 2):
        assert np.array_equal(g, w_)


def test_segment_images_equals_one_call_per_image(torch_cuda):
    """Segmenter.segment_images: a data-set loop over images of mixed shapes, grouped per shape behind the scenes, yields in
    input order exactly what segment(img) gives for each image (script.py:22-38 calls the slot once per image)."""
    from gabor_color_image_segmentation_amd import Segmenter
    seg = Segmenter(n_iter=3)
    shapes = [(40, 64), (64, 40), (40, 64), (33, 57), (64, 40), (40, 64), (40, 64), (64, 40), (40, 64)]
    imgs = [_synth(1, h, w, seed=50 + i)[0] for i, (h, w) in enumerate(shapes)]
    want = [seg(im) for im in imgs]
    for batch in (1, 2, 3, 64):
        got = list(seg.segment_images(iter(imgs), batch=batch))
        assert len(got) == len(want)
        for g, w_, im in zip(got, want, imgs):
            assert g.dtype == np.int32 and g.shape == im.shape[:2] and np.array_equal(g, w_), batch
    assert list(seg.segment_images([])) == []
    assert [g.dtype for g in seg.segment_images(imgs[:2], out_dtype=np.uint8)] == [np.uint8, np.uint8]
    # batches above the graph-replay size go through the three-stream pipeline, one pipeline per shape: 2 shapes x (2 full
    # batches of 20 + a remainder), interleaved in the input
    big = [_synth(1, *((240, 248) if i % 3 else (248, 240)), seed=200 + i)[0] for i in range(75)]
    want = [seg(im) for im in big[:6]] + [None] * 63 + [seg(im) for im in big[69:]]
    got = list(seg.segment_images(iter(big), batch=20))
    assert len(got) == 75 and all(g.shape == im.shape[:2] for g, im in zip(got, big))
    for g, w_ in zip(got, want):
        assert w_ is None or np.array_equal(g, w_)
    again = list(seg.segment_images(big, batch=20, out_dtype=np.uint8))
    assert all(np.array_equal(a, g) for a, g in zip(again, got))
    with pytest.raises(ValueError):
        list(seg.segment_images([imgs[0][..., 0]]))


def test_download_moves_every_byte(torch_cuda):
    """gcs_download (the label download of segment_stream, SDMA through the pitched copy): sizes below, at and above its
    64 KiB row, with and without a remainder; nothing written past the end; a pageable destination is refused."""
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    ops = Segmenter().ops
    for n in (3, 65536, 65536 * 7, 65536 * 3 + 5, 1_000_003, 64 * 321 * 481 * 4):
        src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
        dst = torch.full((n + 64,), 7, dtype=torch.uint8).pin_memory()
        ops.download(dst[:n], src)
        torch.cuda.synchronize()
        assert torch.equal(dst[:n], src.cpu()) and bool((dst[n:] == 7).all()), n
    with pytest.raises(ValueError):
        ops.download(torch.empty(n, dtype=torch.uint8), src)             # pageable destination


def test_global_codebook_bit_exact(torch_cuda):
    from gabor_color_image_segmentation_amd import Segmenter
    imgs = _synth(4, 40, 64, seed=8)
    got = Segmenter(n_iter=5).segment_batch(imgs, mode="global")
    ref = so.segment_batch(imgs, mode="global", n_iter=5)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("no,ks", [(4, 13), (6, 13), (8, 13), (5, 13), (7, 13), (6, 15), (8, 15), (2, 9)])
@pytest.mark.parametrize("b,h,w", [(1, 81, 121), (3, 72, 104)])
def test_small_call_launch_forms_of_two_level_banks(torch_cuda, no, ks, b, h, w):
    """Round 5: a small call (every tile of both levels fits the resident slots) runs both pre-passes in one launch
    (gabor_pre01_kernel) and the bank's row tiles side by side as blockIdx.y groups of one launch when the 2 * n_orient filters of a
    level fill whole row tiles (n_orient 4 / 6 / 8: two / three / four groups; 8-K-step frames for ksize 15); otherwise the fused
    two-level list of three row tiles (LVL = -2: n_orient 5) or two launches of two (n_orient 7); n_orient 2: one tile, no groups.
    Features and labels == the oracle, packed edge strips (81 = 8 k + 1, 121 = 8 k + 1) included."""
    from gabor_color_image_segmentation_amd import Segmenter
    torch = torch_cuda
    imgs = _synth(b, h, w, seed=500 + no)
    seg = Segmenter(n_scales=4, n_orient=no, ksize=ks, k=5, n_iter=3)
    got = seg.features_device(torch.from_numpy(imgs).cuda()).cpu().numpy().view(np.uint16)
    tapq, shift = so.bank(4, no, ks)
    for i in range(b):
        assert np.array_equal(got[i], so.gabor_features(imgs[i], tapq, shift, no)), (no, ks, i)
    lab = seg.segment_batch(imgs)
    for i in range(b):
        assert np.array_equal(lab[i], so.segment(imgs[i], n_scales=4, n_orient=no, ksize=ks, k=5, n_iter=3)), (no, ks, i)


@pytest.mark.parametrize("ns,no,ks,k", [(1, 1, 15, 3), (1, 4, 9, 5), (2, 5, 11, 8), (3, 8, 15, 16), (2, 13, 7, 4),
                                        (3, 9, 15, 4), (1, 43, 11, 5), (8, 8, 15, 8), (8, 8, 15, 13), (3, 23, 9, 16),
                                        (3, 23, 9, 7), (6, 8, 15, 8), (5, 8, 13, 7), (8, 6, 13, 8), (7, 6, 11, 6)])
def test_segment_small_and_ragged_feature_counts(torch_cuda, ns, no, ks, k):
    """D = 3, 12, 30, 72, 78: every staging-chunk bucket of the narrow MFMA k-means pass, D not a multiple of 8;
    D = 81, 129, 192, 207: every bucket of the wide (208-row) pass, k <= 8 and k > 8. The last four are the
    deep-bank pass's remaining instantiations: 6x8 and 5x8 (three levels, 48 planes on level 0: kmeans_pass_native_kernel
    <3, 3, 6>), 8x6 and 7x6 (four levels of 36 planes: <4, 2, 0>, the two-workgroups-per-CU table form)."""
    from gabor_color_image_segmentation_amd import Segmenter
    imgs = _synth(2, 40, 72, seed=31)
    seg = Segmenter(n_scales=ns, n_orient=no, ksize=ks, k=k, n_iter=4)
    got = seg.segment_batch(imgs)
    for b in range(2):
        assert np.array_equal(got[b], so.segment(imgs[b], n_scales=ns, n_orient=no, ksize=ks, k=k, n_iter=4))


def test_images_narrower_than_a_kmeans_tile(torch_cuda):
    """pitch < 256: a 256-pixel tile spans several rows (the modulo path of the validity test)."""
    from gabor_color_image_segmentation_amd import Segmenter
    imgs = _synth(2, 70, 19, seed=5)
    seg = Segmenter(n_iter=3, k=4)
    got = seg.segment_batch(imgs, mode="global")
    assert np.array_equal(got, so.segment_batch(imgs, mode="global", k=4, n_iter=3))


@pytest.mark.parametrize("b,h,w", [(24, 100, 120), (3, 300, 500), (7, 161, 241)])
def test_global_codebook_batch_list_walk(torch_cuda, b, h, w):
    """One global codebook: the pass walks the tiles of the whole batch as ONE list (stride = all workgroups) and tracks
    the tile index inside its image without a division; a stride crosses image boundaries once or several times
    (49, 599 and 164 tiles per image against 768 workgroups). Forward and reverse sweeps (4 passes) against the C oracle."""
    from oracle import c_oracle as co
    from gabor_color_image_segmentation_amd import Segmenter
    imgs = _synth(b, h, w, seed=77 + b)
    seg = Segmenter(k=5, n_iter=4)
    got = seg.segment_batch(imgs, mode="global")
    ref = co.segment_batch(imgs, seg.bank.tapq, seg.bank.shift, seg.bank.n_orient, k=5, n_iter=4, mode="global")
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("ns,no,b", [(6, 8, 5), (8, 6, 5), (5, 8, 400), (7, 6, 3)])
def test_deep_bank_pass_global_codebook_batches(torch_cuda, ns, no, b):
    """the deep-bank pass's <3, 3, 6> and <4, 2, 0> forms with a batch and ONE global codebook: the batch-list walk,
    reverse sweeps (4 passes) and - 400 small images of 4 tiles - fewer resident slots per image than partial rows (parts_eff 1 < parts 2),
    every label against the C oracle."""
    from oracle import c_oracle as co
    from gabor_color_image_segmentation_amd import Segmenter
    h, w = (56, 88) if b < 10 else (24, 40)
    imgs = _synth(b, h, w, seed=90 + b)
    seg = Segmenter(n_scales=ns, n_orient=no, k=7, n_iter=4)
    got = seg.segment_batch(imgs, mode="global")
    ref = co.segment_batch(imgs, seg.bank.tapq, seg.bank.shift, seg.bank.n_orient, k=7, n_iter=4, mode="global")
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("ns,no", [(4, 6), (8, 8), (3, 23), (6, 8), (5, 8), (8, 6), (7, 6)])
def test_lloyd_pass_with_one_output_only(torch_cuda, ns, no):
    """`labels_dev == NULL` (passes whose assignment nobody reads) and `partials_dev == NULL` (the last pass) give the
    same labels / the same partial sums as the call with both outputs: narrow, wide (8-wave) and generic pass, and every
    instantiation of the deep-bank pass (8x8, 6x8, 5x8, 8x6, 7x6)."""
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    b, h, w = 3, 56, 88
    imgs = torch.from_numpy(_synth(b, h, w, seed=12)).cuda()
    seg = Segmenter(n_scales=ns, n_orient=no, k=6)
    ws = seg._workspace(b, h, w, "global")
    seg.ops.gabor_features(imgs, ws["feats"])
    seg.ops.kmeans_init(ws["feats"], b, h, w, seg.k, 1, ws["cent"])
    lab2, par2 = torch.full_like(ws["labels"], 255), torch.zeros_like(ws["partials"])
    ws["partials"].zero_()                                   # the padding of the chunked partial layout is never written
    seg.ops.assign_accumulate(ws["feats"], ws["cent"], b, h, w, seg.k, 1, ws["labels"], ws["partials"])
    seg.ops.assign_accumulate(ws["feats"], ws["cent"], b, h, w, seg.k, 1, lab2, None)
    seg.ops.assign_accumulate(ws["feats"], ws["cent"], b, h, w, seg.k, 1, None, par2)
    torch.cuda.synchronize()
    assert torch.equal(lab2, ws["labels"])
    assert torch.equal(par2, ws["partials"])


def test_step_replays_from_a_captured_graph(torch_cuda):
    """The C ABI's stream contract (include/gcs.h): a whole step is captured as a HIP graph (gcs_gabor_features keeps a
    captured call on the capturing stream: no side-stream fork); replays on new inputs equal the eager result."""
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    b, h, w = 16, 321, 481
    a = torch.from_numpy(_synth(b, h, w, seed=3)).cuda()
    c = torch.from_numpy(_synth(b, h, w, seed=4)).cuda()
    seg = Segmenter(n_iter=3)
    want_a = seg.segment_device(a, mode="global").clone()       # also builds and places the workspace, outside the capture
    want_c = seg.segment_device(c, mode="global").clone()
    assert not torch.equal(want_a, want_c)
    static_in = a.clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_out = seg.segment_device(static_in, mode="global")
    for src, want in ((c, want_c), (a, want_a), (c, want_c)):
        static_in.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(static_out, want)


def test_two_threads_two_streams_share_the_side_stream(torch_cuda):
    """gcs_gabor_features forks large batches onto ONE library-owned side stream per device with reused fork / join events:
    two host threads driving their own plans on their own streams at the same time must still get their own results."""
    import threading
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    b, h, w = 16, 321, 481
    data = [torch.from_numpy(_synth(b, h, w, seed=20 + i)).cuda() for i in range(2)]
    segs = [Segmenter(n_iter=2) for _ in range(2)]
    want = [segs[i].segment_device(data[i], mode="global").clone() for i in range(2)]
    torch.cuda.synchronize()
    got, errs = [None, None], []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(6):
                    out = segs[i].segment_device(data[i], mode="global")
                got[i] = out.clone()
            st.synchronize()
        except Exception as e:                              # surfaced in the main thread below
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(2):
        assert torch.equal(got[i], want[i])


def test_randomised_shapes_banks_and_codebooks(torch_cuda):
    """40 seeded random cases against the C oracle: image sizes from the 8x8 minimum to a few tiles (odd widths,
    widths below one Gabor / k-means tile, heights that leave waves idle), banks F = 1..30 with every odd ksize,
    k = 1..16, n_iter = 1..6, batch 1..5, both codebook modes, connectivity on and off."""
    from oracle import c_oracle as co
    from gabor_color_image_segmentation_amd import Segmenter
    rng = np.random.default_rng(20261004)
    for case in range(40):
        h = int(rng.integers(8, 150))
        w = int(rng.integers(8, 300)) if case % 3 else int(rng.integers(8, 40))
        b = int(rng.integers(1, 6))
        ns, no = int(rng.integers(1, 6)), int(rng.integers(1, 7))
        ks = int(rng.choice([1, 3, 5, 7, 9, 11, 13, 15]))
        k, n_iter = int(rng.integers(1, 17)), int(rng.integers(1, 7))
        mode = "global" if rng.integers(0, 2) else "per_image"
        conn = bool(rng.integers(0, 4) == 0)
        imgs = _synth(b, h, w, seed=1000 + case)
        if case % 7 == 0:                                    # hard-edged noise instead of smooth synthetic regions
            imgs = rng.integers(0, 256, imgs.shape, dtype=np.uint8)
        seg = Segmenter(n_scales=ns, n_orient=no, ksize=ks, k=k, n_iter=n_iter, connectivity=conn)
        got = seg.segment_batch(imgs, mode=mode)
        ref = co.segment_batch(imgs, seg.bank.tapq, seg.bank.shift, no, k=k, n_iter=n_iter, mode=mode)
        if conn:
            ref = np.stack([so.connected_regions(r) for r in ref])
        assert np.array_equal(got, ref), dict(case=case, h=h, w=w, b=b, bank=(ns, no, ks), k=k, n_iter=n_iter,
                                              mode=mode, conn=conn, wrong=float((got != ref).mean()))


def test_isqrt31_exhaustive(torch_cuda):
    """The 7-instruction exact integer square root of the Gabor epilogue (csrc/gabor.hip isqrt31) on EVERY n of its
    domain and beyond: every n in [0, 2^31) (SPEC.md §3: re^2 + im^2 <= 2 * 32767^2 < 2^31, kept by gcs_bank_pack's tap-sum bound): one kernel, 2.1e9 values."""
    import ctypes as C
    from gabor_color_image_segmentation_amd import _lib
    torch = torch_cuda
    lib = _lib.load()
    bad = torch.full((1,), 123, dtype=torch.int32, device="cuda")
    n_max = 2 ** 31 - 1
    _lib.check(lib.gcs_selftest_isqrt(n_max, bad.data_ptr(), torch.cuda.current_stream().cuda_stream), "selftest")
    assert int(bad.item()) == 0


def test_pyramid_levels_with_odd_sizes_and_deep_banks(torch_cuda):
    """Images whose every pyramid level has an odd size (edge replication on each level), banks of 1..4 levels with one
    or two scales on the last level: canonical features against the NumPy oracle."""
    from gabor_color_image_segmentation_amd import Segmenter
    torch = torch_cuda
    for (h, w), (ns, no) in [((73, 101), (8, 2)), ((41, 57), (7, 3)), ((67, 35), (5, 1)), ((24, 120), (6, 4)),
                             ((9, 9), (8, 1)), ((8, 8), (3, 2))]:
        imgs = _synth(2, h, w, seed=h + w)
        seg = Segmenter(n_scales=ns, n_orient=no)
        got = seg.features_device(torch.from_numpy(imgs).cuda()).cpu().numpy().view(np.uint16)
        tapq, shift = so.bank(ns, no)
        for b in range(2):
            ref = so.gabor_features(imgs[b], tapq, shift, no)
            bad = np.argwhere(got[b] != ref)
            assert bad.size == 0, ((h, w), (ns, no), len(bad), bad[:4])


def test_host_path_variants_agree_with_the_device_path(torch_cuda):
    """segment_batch (host array in, host array out: pinned staging, chunked upload on a second stream, labels copied
    into a fresh pinned buffer) against segment_device, for batch sizes around the chunk count, both codebook modes and
    both output dtypes; results are fresh arrays the caller owns (a later call does not change an earlier result)."""
    from gabor_color_image_segmentation_amd import Segmenter
    torch = torch_cuda
    seg = Segmenter(n_iter=3)
    keep = []
    for b in (1, 2, 3, 5, 9):
        imgs = _synth(b, 40 + b, 72, seed=50 + b)
        for mode in ("per_image", "global"):
            want = seg.segment_device(torch.from_numpy(imgs).cuda(), mode=mode).cpu().numpy()
            got32 = seg.segment_batch(imgs, mode=mode)
            got8 = seg.segment_batch(imgs, mode=mode, out_dtype=np.uint8)
            assert got32.dtype == np.int32 and got8.dtype == np.uint8 and got32.shape == want.shape
            assert np.array_equal(got32, want) and np.array_equal(got8, want)
            keep.append((got32, want))
    for got, want in keep:                                  # earlier results were not overwritten by later calls
        assert np.array_equal(got, want)
    assert np.array_equal(seg(imgs[0]), want[0] if mode == "per_image" else seg.segment_device(
        torch.from_numpy(imgs[:1]).cuda()).cpu().numpy()[0])
    with pytest.raises(ValueError):
        seg.segment_batch(imgs, out_dtype=np.int64)
    with pytest.raises(ValueError):
        Segmenter(connectivity=True).segment_batch(imgs, out_dtype=np.uint8)
    with pytest.raises(ValueError):
        seg.segment_device(torch.from_numpy(imgs))          # host tensor handed to the device API


def test_randomised_shapes_banks_and_codebooks_against_the_c_oracle(torch_cuda):
    """tests/checkers/fuzz_features.py: 60 random cases (tiny / odd / one-row-remainder shapes, batches 1-9, banks of 1-8 scales with
    odd orientation counts, ksize 1-15, k 1-16, both codebook modes, constant extreme images): features and labels equal the C
    oracle's bit for bit; since round 6 half of the cases carry full-contrast patches (flagged tiles of the split slab). (400 further
    cases were run once in round 3, 300 - seed 4, with the packed edge strips - in round 4, 400 - seed 11 - in round 6: 210 of them on
    the split slab, 128 of those with values >= 4096, 96 of those on shapes with packed edge strips; and on the round's final kernels
    (compact level 1 in LDS, swizzled rows, buffer loads) 300 + 600 more - seeds 13 and 14, 131 of the 600 on one-level banks -:
    0 mismatches, profiles/r6_fuzz400.log, r6_fuzz300.log, r6_fuzz600.log.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "checkers", "fuzz_features.py"), "60", "3"], cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "60 cases, 0 bad" in r.stdout
