"""N>1 path on CPU: two gloo ranks, each with its shard of a global batch, exchange only the
int64 centroid sums (plus the init broadcast). Result must equal the unsharded oracle."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, per_rank, n_iter, k, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gabor_color_image_segmentation_amd import Segmenter, make_bank
    from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
    from fake_ops import OracleOps
    imgs = synthetic_shard(rank * per_rank, per_rank, 24, 40, seed=5)
    seg = Segmenter(k=k, n_iter=n_iter, ops=OracleOps(make_bank()))
    out = seg.segment_device(torch.from_numpy(imgs), mode="global").numpy()
    np.save(os.path.join(tmp, f"labels_{rank}.npy"), out)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,per_rank", [(2, 2), (2, 1), (8, 1)])      # 8: the target machine's world size
def test_global_codebook_two_ranks_equals_unsharded_oracle(tmp_path, world, per_rank):
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    from oracle import spec_oracle as so
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, per_rank, 4, 6, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"labels_{r}.npy") for r in range(world)])
    ref = so.segment_batch(synthetic_batch(world * per_rank, 24, 40, seed=5), mode="global", k=6, n_iter=4)
    assert np.array_equal(got, ref)


def _strip_worker(rank, world, port, b, height, width, n_iter, k, tmp, use_gpu):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gabor_color_image_segmentation_amd import Segmenter, make_bank, shard_rows
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(b, height, width, seed=13)
    r0, r1, s0, s1 = shard_rows(height, world, rank)          # defaults = the default bank: 13x13 on two levels, 12 halo rows
    strip = torch.from_numpy(np.ascontiguousarray(imgs[:, s0:s1]))
    if use_gpu:
        seg = Segmenter(k=k, n_iter=n_iter, device="cuda:0")
        strip = strip.cuda()
    else:
        from fake_ops import OracleOps
        seg = Segmenter(k=k, n_iter=n_iter, ops=OracleOps(make_bank()))
    out = seg.segment_rows_sharded_device(strip, r0, r1, s0, height)
    np.save(os.path.join(tmp, f"strip_{rank}.npy"), out.cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_row_sharded_image_equals_unsharded_oracle(tmp_path, world):
    """BASELINE config 5 in miniature: one image split into row strips with the halo the bank's kernel needs (6 rows per
    pyramid level = 12 rows: `halo_rows(2, 13)`, the defaults of `shard_rows`; `Segmenter.shard_rows` reads them from its bank)."""
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    from oracle import spec_oracle as so
    port = 31500 + (os.getpid() % 2000) + world
    b, height, width = (2, 60, 40) if world < 8 else (1, 112, 40)      # eight strips of 14 rows: each at least the 12-row halo
    mp.spawn(_strip_worker, args=(world, port, b, height, width, 4, 5, str(tmp_path), False), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"strip_{r}.npy") for r in range(world)], axis=1)
    ref = so.segment_batch(synthetic_batch(b, height, width, seed=13), mode="global", k=5, n_iter=4)
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("height,width", [(200, 136), (202, 137)])
def test_row_sharded_image_on_gpu_two_ranks(tmp_path, built, height, width):
    """Same, through the HIP kernels: two processes share cuda:0, gloo carries the tiny collectives. 202 x 137: the second
    rank's strip is 114 rows of 137 pixels - a two-row bottom edge and a one-pixel right edge, i.e. the packed edge strips of
    the slab (csrc/common.h) together with a row window (halo rows labelled, not voting)."""
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    from oracle import c_oracle as co, spec_oracle as so
    port = 33500 + (os.getpid() % 2000) + height
    b = 1
    mp.spawn(_strip_worker, args=(2, port, b, height, width, 5, 8, str(tmp_path), True), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / f"strip_{r}.npy") for r in range(2)], axis=1)
    tapq, shift = so.bank()
    ref = co.segment_batch(synthetic_batch(b, height, width, seed=13), tapq, shift, 6, k=8, n_iter=5, mode="global")
    assert np.array_equal(got, ref)


def _big_strip_worker(rank, world, port, height, width, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gabor_color_image_segmentation_amd import Segmenter, shard_rows
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(1, height, width, seed=41)
    r0, r1, s0, s1 = shard_rows(height, world, rank)
    seg = Segmenter(n_iter=4, device="cuda:0")
    out = seg.segment_rows_sharded_device(torch.from_numpy(np.ascontiguousarray(imgs[:, s0:s1])).cuda(), r0, r1, s0, height)
    np.save(os.path.join(tmp, f"big_{rank}.npy"), out.cpu().numpy().astype(np.uint8))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_config5_size_2048_sharded_equals_unsharded_on_gpu(tmp_path, built):
    """BASELINE config 5's real tile size: one 2048x2048 image, 2 row strips with halo == the unsharded GPU result == the
    C oracle's label map of the whole image."""
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    height = width = 2048
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_big_strip_worker, args=(2, port, height, width, str(tmp_path)), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / f"big_{r}.npy") for r in range(2)], axis=1)
    imgs = synthetic_batch(1, height, width, seed=41)
    ref = Segmenter(n_iter=4).segment_batch(imgs, mode="global")
    assert got.shape == ref.shape and np.array_equal(got, ref.astype(np.uint8))
    assert len(np.unique(ref)) > 1
    # and both equal the C oracle at this size (4.2 Mpix: a few seconds with OpenMP)
    from oracle import c_oracle as co, spec_oracle as so
    tapq, shift = so.bank()
    assert np.array_equal(ref, co.segment_batch(imgs, tapq, shift, 6, n_iter=4, mode="global"))


def _rccl_worker(rank, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from gabor_color_image_segmentation_amd.segmenter import _collective
    sums = (torch.arange(8 * 73, dtype=torch.int64, device=dev).view(1, 8, 73) + 1) * ((1 << 55) + 12345)
    keep = sums.clone()
    _collective(dist.all_reduce, sums, op=dist.ReduceOp.SUM)          # what lloyd() does between reduce and finalize
    cent = (torch.arange(8 * 72, dtype=torch.int32, device=dev) * 113).to(torch.int16).view(1, 8, 72)
    keep_c = cent.clone()
    _collective(dist.broadcast, cent.view(torch.uint8), src=0)         # the init-centroid broadcast
    dist.barrier()
    torch.cuda.synchronize()
    ok = bool(torch.equal(sums, keep)) and bool(torch.equal(cent, keep_c))
    # the whole multi-rank branch of lloyd() (reduce kernel -> RCCL all-reduce -> finalize kernel, init broadcast)
    # on this one GPU: kernels on torch's current stream, RCCL on its own stream, ordered by torch's events
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    from oracle import spec_oracle as so
    imgs = synthetic_batch(3, 72, 104, seed=21)
    seg = Segmenter(k=6, n_iter=5, device="cuda:0")
    seg.debug.force_collectives = True
    got = seg.segment_device(torch.from_numpy(imgs).to(dev), mode="global").cpu().numpy()
    ok = ok and bool(np.array_equal(got, so.segment_batch(imgs, mode="global", k=6, n_iter=5)))
    open(os.path.join(tmp, "rccl_ok"), "w").write("1" if ok else "0")
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_carries_the_collective_dtypes(tmp_path, built):
    """The N > 1 GPU path uses exactly two collectives: all_reduce(SUM) of an int64 tensor (values beyond 2^53,
    so a detour through floating point would show) and broadcast of the int16 centroids viewed as bytes. A
    one-rank RCCL group on cuda:0 checks that this RCCL build initialises here and accepts both, then runs the whole
    multi-rank branch of lloyd() through RCCL against the oracle."""
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_rccl_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    assert open(tmp_path / "rccl_ok").read() == "1"


def _global_hip_worker(rank, world, port, per_rank, height, width, n_iter, k, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
    imgs = synthetic_shard(rank * per_rank, per_rank, height, width, seed=0)      # this rank's shard of the global batch
    seg = Segmenter(k=k, n_iter=n_iter, device="cuda:0")
    out = seg.segment_device(torch.from_numpy(imgs).cuda(), mode="global").cpu().numpy()
    np.save(os.path.join(tmp, f"glabels_{rank}.npy"), out)
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,per_rank", [(2, 3), (3, 2)])
def test_global_codebook_hip_ranks_equal_unsharded_c_oracle(tmp_path, built, world, per_rank):
    """BASELINE config 3 in miniature, through the HIP kernels: `world` processes share cuda:0, each owns `per_rank` images
    of one global batch (bench.py's sharding: synthetic_shard), rank 0 broadcasts the init centroids, every Lloyd pass
    all-reduces the int64 sums (gloo carries the collectives here, RCCL on a multi-GPU node). The concatenated label maps
    must equal the C oracle's global-codebook result on the unsharded batch, pixel for pixel."""
    from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
    from oracle import c_oracle as co, spec_oracle as so
    port = 36500 + (os.getpid() % 2000) + world
    height, width, n_iter, k = 136, 200, 6, 8
    mp.spawn(_global_hip_worker, args=(world, port, per_rank, height, width, n_iter, k, str(tmp_path)), nprocs=world,
             join=True)
    got = np.concatenate([np.load(tmp_path / f"glabels_{r}.npy") for r in range(world)])
    tapq, shift = so.bank()
    ref = co.segment_batch(synthetic_shard(0, world * per_rank, height, width, seed=0), tapq, shift, 6, k=k, n_iter=n_iter,
                           mode="global")
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.gpu
def test_bench_two_ranks_as_the_driver_launches_it(tmp_path, built):
    """`python -m torch.distributed.run ... bench.py --gpus 2` exactly as the driver starts the N > 1 runs (one fresh
    subprocess per rank; both ranks land on cuda:0 here and gloo stands in for RCCL): one JSON line from rank 0 with
    n_gpus = 2 and a global batch of 2 x 64 images."""
    import json
    import subprocess
    port = 37500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
           "--warmup", "1", "--spinup-steps", "1"]
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["scaling"] == "weak"
    assert d["value"] > 0 and np.isfinite(d["value"]) and d["steps"] == 2
    assert d["config"]["codebook"] == "global" and "all-reduce" in d["config"]["parallelism"]
    # the N > 1 run diagnoses itself: backend, per-rank times, the cost of one update with its all-reduce
    m = d["multi_gpu"]
    assert m["world_size"] == 2 and m["backend"] == "gloo" and len(m["rank_ms_per_step"]) == 2
    assert m["rank_ms_min"] <= m["rank_ms_max"] and abs(m["rank_ms_max"] - d["ms_per_step"]) < 1e-6
    assert m["collective_samples"] >= 1 and m["collective_ms_per_pass"] > 0 and m["allreduce_bytes"] == 8 * 8 * 73
    assert 0 < m["predicted_efficiency"] <= 1


def _owned_rows_worker(rank, world, port, b, height, width, n_iter, k, tmp, use_gpu):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gabor_color_image_segmentation_amd import Segmenter, make_bank, shard_rows
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(b, height, width, seed=13)
    r0, r1, _, _ = shard_rows(height, world, rank)
    owned = torch.from_numpy(np.ascontiguousarray(imgs[:, r0:r1]))             # this rank never sees a neighbour's rows
    if use_gpu:
        seg = Segmenter(k=k, n_iter=n_iter, device="cuda:0")
        owned = owned.cuda()
    else:
        from fake_ops import OracleOps
        seg = Segmenter(k=k, n_iter=n_iter, ops=OracleOps(make_bank()))
    out = seg.segment_owned_rows_device(owned, height)
    np.save(os.path.join(tmp, f"owned_{rank}.npy"), out.cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_halo_exchange_between_ranks_equals_unsharded_oracle(tmp_path, world):
    """BASELINE config 5 with the cross-rank halo exchange (SURVEY §8e option (ii)): every rank holds only its own rows;
    the 12 halo rows per interior edge travel rank to rank (point-to-point), nothing comes from a host copy of the whole
    image. Result == the unsharded oracle."""
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    from oracle import spec_oracle as so
    port = 38500 + (os.getpid() % 2000) + world
    b, height, width = (2, 72, 40) if world < 8 else (1, 112, 40)      # eight ranks: 14 own rows each, seven interior edges
    mp.spawn(_owned_rows_worker, args=(world, port, b, height, width, 3, 5, str(tmp_path), False), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"owned_{r}.npy") for r in range(world)], axis=1)
    ref = so.segment_batch(synthetic_batch(b, height, width, seed=13), mode="global", k=5, n_iter=3)
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.gpu
def test_halo_exchange_on_gpu_three_ranks(tmp_path, built):
    """Same through the HIP kernels: three processes share cuda:0, halos are exchanged between their device tensors (gloo
    carries them here, RCCL send / recv on a multi-GPU node), against the C oracle."""
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    from oracle import c_oracle as co, spec_oracle as so
    port = 39500 + (os.getpid() % 2000)
    b, height, width = 1, 200, 136
    mp.spawn(_owned_rows_worker, args=(3, port, b, height, width, 5, 8, str(tmp_path), True), nprocs=3, join=True)
    got = np.concatenate([np.load(tmp_path / f"owned_{r}.npy") for r in range(3)], axis=1)
    tapq, shift = so.bank()
    ref = co.segment_batch(synthetic_batch(b, height, width, seed=13), tapq, shift, 6, k=8, n_iter=5, mode="global")
    assert np.array_equal(got, ref)


# ---------------------------------------------------------------------------------------------------------------------
# The target's world size: EIGHT ranks. The GPU box admits at most six processes on its card
# (gpurun's process guard), so eight processes sharing cuda:0 is not something a test may start. Two forms instead:
#   * eight ranks as eight THREADS of one process (tests/threaded_world.py), every rank through the HIP kernels, the collectives
#     on device tensors: config 3 (8 x 2 images) and config 5 with host-delivered halos (8 strips of 256 rows of one 2048 x 2048
#     image);
#   * eight PROCESSES over gloo for the rank-to-rank halo exchange (point-to-point), five of them on cuda:0 (the pytest process
#     itself is the sixth on the card) and three computing their strip with the oracle-backed stand-in of tests/fake_ops.py: every rank takes part in every exchange, and exact
#     integer arithmetic makes the label map independent of who computed which strip.
# No scaling curve is measured by any of this (one GPU): none is claimed.

@pytest.mark.gpu
def test_config3_eight_hip_ranks_equal_unsharded_c_oracle(built):
    """BASELINE config 3 at the target's world size: 8 ranks x 2 images of 136 x 200 (bench.py's sharding), all eight through
    the HIP kernels on cuda:0, init broadcast + one int64 all-reduce per Lloyd pass on DEVICE tensors == the C oracle's
    global-codebook result on the unsharded 16-image batch."""
    from threaded_world import run_threaded
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
    from oracle import c_oracle as co, spec_oracle as so
    world, per_rank, height, width, n_iter, k = 8, 2, 136, 200, 6, 8

    def rank_fn(rank, world):
        assert dist.get_world_size() == 8 and dist.get_backend() == "threaded"
        imgs = synthetic_shard(rank * per_rank, per_rank, height, width, seed=0)
        seg = Segmenter(k=k, n_iter=n_iter, device="cuda:0")
        return seg.segment_device(torch.from_numpy(imgs).cuda(), mode="global").cpu().numpy()

    got = np.concatenate(run_threaded(world, rank_fn))
    tapq, shift = so.bank()
    ref = co.segment_batch(synthetic_shard(0, world * per_rank, height, width, seed=0), tapq, shift, 6, k=k, n_iter=n_iter,
                           mode="global")
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.gpu
def test_config3_at_its_stated_size_eight_hip_ranks_times_64_images(built):
    """BASELINE config 3 AS IT IS STATED: batch 512 of 481 x 321 x 3, eight ranks x 64 images (bench.py's
    sharding: rank r holds images [64 r, 64 (r + 1)) of the seed-0 batch), 4x6 bank, k = 8, n_iter = 10, one global codebook:
    init broadcast + one int64 all-reduce per Lloyd pass on device tensors, all eight ranks through the HIP kernels (threads of
    one process: the box admits six processes per card). EVERY label of the 512 maps == the C oracle's global-codebook result
    on the unsharded 512-image batch (about a minute of OpenMP). No scaling curve is measured by this (one GPU): none is claimed."""
    from threaded_world import run_threaded
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
    from oracle import c_oracle as co, spec_oracle as so
    world, per_rank, height, width = 8, 64, 321, 481

    def rank_fn(rank, world):
        imgs = synthetic_shard(rank * per_rank, per_rank, height, width, seed=0)
        seg = Segmenter(device="cuda:0")                                         # defaults = BASELINE config 3's bank, k, n_iter
        return seg.segment_device(torch.from_numpy(imgs).cuda(), mode="global").cpu().numpy().astype(np.uint8)

    got = np.concatenate(run_threaded(world, rank_fn))
    assert got.shape == (512, height, width)
    # the oracle, image by image (the stacked uint16 feature tensor of 512 images is 11.4 GB; co.segment_batch would hold it twice)
    tapq, shift = so.bank()
    feats = np.empty((world * per_rank, 72, height * width), np.uint16)
    for r in range(world):
        imgs = synthetic_shard(r * per_rank, per_rank, height, width, seed=0)
        for i in range(per_rank):
            feats[r * per_rank + i] = co.gabor_features(imgs[i], tapq, shift, 6).reshape(72, -1)
    ref = co.kmeans(feats, 8, 10)[0].reshape(-1, height, width)
    del feats
    assert np.array_equal(got, ref.astype(np.uint8)) and len(np.unique(ref)) == 8


@pytest.mark.gpu
def test_config5_2048_in_eight_strips_of_256_rows_on_gpu(built):
    """BASELINE config 5 as SURVEY §8e specifies it: ONE 2048 x 2048 image, rank r owns rows [256 r, 256 (r + 1)), 12-row
    halos delivered with the strip (option (i)); eight ranks through the HIP kernels (threads of one process). The stitched
    map == the unsharded GPU result == the C oracle's label map of the whole image."""
    from threaded_world import run_threaded
    from gabor_color_image_segmentation_amd import Segmenter, shard_rows
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    height = width = 2048
    imgs = synthetic_batch(1, height, width, seed=41)

    def rank_fn(rank, world):
        r0, r1, s0, s1 = shard_rows(height, world, rank)
        assert (r0, r1) == (256 * rank, 256 * (rank + 1)) and s0 == max(0, r0 - 12) and s1 == min(height, r1 + 12)
        seg = Segmenter(n_iter=4, device="cuda:0")
        strip = torch.from_numpy(np.ascontiguousarray(imgs[:, s0:s1])).cuda()
        return seg.segment_rows_sharded_device(strip, r0, r1, s0, height).cpu().numpy().astype(np.uint8)

    got = np.concatenate(run_threaded(8, rank_fn), axis=1)
    ref = Segmenter(n_iter=4).segment_batch(imgs, mode="global")
    assert got.shape == ref.shape and np.array_equal(got, ref.astype(np.uint8)) and len(np.unique(ref)) > 1
    from oracle import c_oracle as co, spec_oracle as so
    tapq, shift = so.bank()
    assert np.array_equal(ref, co.segment_batch(imgs, tapq, shift, 6, n_iter=4, mode="global"))


def _mixed_owned_rows_worker(rank, world, port, n_gpu_ranks, height, width, n_iter, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gabor_color_image_segmentation_amd import Segmenter, make_bank, shard_rows
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(1, height, width, seed=41)
    r0, r1, _, _ = shard_rows(height, world, rank)
    owned = torch.from_numpy(np.ascontiguousarray(imgs[:, r0:r1]))             # this rank never sees a neighbour's rows
    if rank < n_gpu_ranks:
        seg = Segmenter(n_iter=n_iter, device="cuda:0")
        owned = owned.cuda()
    else:                                                                      # (the card takes six processes - this session's included -, the world is eight)
        from fake_ops import OracleOps
        seg = Segmenter(n_iter=n_iter, ops=OracleOps(make_bank()))
    out = seg.segment_owned_rows_device(owned, height)
    np.save(os.path.join(tmp, f"mixed_{rank}.npy"), out.cpu().numpy().astype(np.uint8))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_config5_halo_exchange_eight_ranks_2048(tmp_path, built):
    """Option (ii) of SURVEY §8e at the target's world size and config 5's real size: eight PROCESSES each hold only their
    own 256 rows of one 2048 x 2048 image, the 12 halo rows of the seven interior edges travel rank to rank (gloo here, RCCL
    send / recv on a multi-GPU node), then init publication + one all-reduce per pass. Five ranks run the HIP kernels on cuda:0
    (with the session's own process: the card's limit of six), three the oracle-backed stand-in. Stitched map == the unsharded GPU result."""
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    height = width = 2048
    port = 40500 + (os.getpid() % 2000)
    mp.spawn(_mixed_owned_rows_worker, args=(8, port, 5, height, width, 3, str(tmp_path)), nprocs=8, join=True)
    got = np.concatenate([np.load(tmp_path / f"mixed_{r}.npy") for r in range(8)], axis=1)
    ref = Segmenter(n_iter=3).segment_batch(synthetic_batch(1, height, width, seed=41), mode="global")
    assert got.shape == ref.shape and np.array_equal(got, ref.astype(np.uint8)) and len(np.unique(ref)) > 1


def test_eight_threaded_ranks_on_cpu_equal_unsharded_oracle():
    """The thread-backed world itself, on CPU with the oracle-backed stand-in: 8 ranks, batch sharding and row sharding
    (host-delivered halos) == the unsharded oracle. Keeps tests/threaded_world.py honest where no GPU is present."""
    sys.path.insert(0, HERE)
    from threaded_world import run_threaded
    from fake_ops import OracleOps
    from gabor_color_image_segmentation_amd import Segmenter, make_bank, shard_rows
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch, synthetic_shard
    from oracle import spec_oracle as so

    def batch_fn(rank, world):
        seg = Segmenter(k=6, n_iter=4, ops=OracleOps(make_bank()))
        return seg.segment_device(torch.from_numpy(synthetic_shard(rank, 1, 24, 40, seed=5)), mode="global").numpy()

    got = np.concatenate(run_threaded(8, batch_fn))
    assert np.array_equal(got, so.segment_batch(synthetic_batch(8, 24, 40, seed=5), mode="global", k=6, n_iter=4))
    imgs = synthetic_batch(1, 112, 40, seed=13)

    def rows_fn(rank, world):
        r0, r1, s0, s1 = shard_rows(112, world, rank)
        seg = Segmenter(k=5, n_iter=4, ops=OracleOps(make_bank()))
        return seg.segment_rows_sharded_device(torch.from_numpy(np.ascontiguousarray(imgs[:, s0:s1])), r0, r1, s0, 112).numpy()

    got = np.concatenate(run_threaded(8, rows_fn), axis=1)
    assert np.array_equal(got, so.segment_batch(imgs, mode="global", k=5, n_iter=4))
