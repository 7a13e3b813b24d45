#!/usr/bin/env python3
"""Headline benchmark: Mpix/s segmented (Gabor bank + k-means) on synthetic 481x321x3 batches.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
one rank per GPU (RCCL). A "step" is one pass of the hot path (Gabor features + n_iter Lloyd
passes + label widen) over one batch of 64 images per GPU, inputs already resident in HBM.
Weak scaling: every rank owns 64 images of a global 64*N batch; in the default `global`
codebook mode the only collective is the int64 centroid-sum all-reduce per Lloyd pass.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 321, 481            # "481x321" in BASELINE.json is width x height
PER_GPU = 64               # BASELINE configs[1]: batch 64 on one MI355X
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA = 2x bf16 (~2.5 PF), same guide, Matrix cores table


def usable_cores():
    """Cores this process may really use: the affinity mask, cut down to the cgroup CPU quota when there is one (a GPU
    box hands a 1-GPU job a share of a many-core host; oversubscribing that share with one thread per visible core
    makes an OpenMP run crawl)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, q // per)
        except Exception:
            pass
    cores = min(n, quota) if quota else n
    cap = int(os.environ.get("GCS_BENCH_CORES", "0"))
    return (min(cores, cap) if cap > 0 else cores), n, quota


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def _numpy_worker(args):
    """Runs in a spawned process: the NumPy + scipy oracle on ONE image, never touches the GPU."""
    idx, k, n_iter = args
    import time as _t
    from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
    from oracle import spec_oracle
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:  # pragma: no cover
        pass
    img = synthetic_shard(idx, 1, H, W, seed=0)[0]
    t0 = _t.perf_counter()
    spec_oracle.segment(img, k=k, n_iter=n_iter)
    return _t.perf_counter() - t0


def _numpy_import_only(_):
    """Warms a pool worker: imports what _numpy_worker imports, so that the pooled figure times segmentation only."""
    from gabor_color_image_segmentation_amd.synthetic import synthetic_shard  # noqa: F401
    from oracle import spec_oracle  # noqa: F401
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:  # pragma: no cover
        pass
    return 0


def _c_oracle_worker(args):
    """Runs in a spawned process (OpenMP inside): the C restatement of SPEC.md on the WHOLE timed batch, same
    codebook mode, so that every label of the configuration bench.py times can be compared with the GPU result."""
    batch, k, n_iter, mode, threads = args
    os.environ["OMP_WAIT_POLICY"] = "PASSIVE"          # idle OpenMP threads sleep instead of spinning on a CPU share
    os.environ["OMP_NUM_THREADS"] = str(threads)
    import time as _t
    from gabor_color_image_segmentation_amd import make_bank
    from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
    from oracle import c_oracle
    bank = make_bank()
    imgs = synthetic_shard(0, batch, H, W, seed=0)
    c_oracle.set_threads(1)
    t0 = _t.perf_counter()
    c_oracle.segment_batch(imgs[:1], bank.tapq, bank.shift, bank.n_orient, k=k, n_iter=n_iter, mode="per_image")
    single = _t.perf_counter() - t0
    c_oracle.set_threads(threads)
    t0 = _t.perf_counter()
    lab = c_oracle.segment_batch(imgs, bank.tapq, bank.shift, bank.n_orient, k=k, n_iter=n_iter, mode=mode)
    return _t.perf_counter() - t0, single, lab.astype("uint8")


def cpu_baseline(batch, k, n_iter, mode):
    """CPU path timed on the host beside the GPU. The oracle is the checker; here it is also the
    reported baseline, never the product. Spawned workers only (no fork after HIP initialisation).

    * `value`: the C restatement (oracle/gcs_oracle.c, OpenMP on every core this process may use) on the whole timed
      batch in the timed codebook mode; its labels are compared with the GPU's, pixel for pixel, all images.
    * `c_single_thread_mpix_s`: the same code, one thread, one image, running alone.
    * `numpy_single_thread_mpix_s`: the NumPy + scipy.ndimage oracle (oracle/spec_oracle.py), one worker running alone.
    """
    import multiprocessing as mp
    cores, affinity, quota = usable_cores()
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        log(f"cpu baseline: C oracle on {cores} threads (affinity {affinity}, cgroup quota {quota}, cpu_count {os.cpu_count()})")
        wall, c_single, ref = pool.apply_async(_c_oracle_worker, [(batch, k, n_iter, mode, cores)]).get(timeout=600)
        log(f"cpu baseline: C oracle {wall:.1f} s; NumPy oracle, one image")
        np_single = pool.apply_async(_numpy_worker, [(0, k, n_iter)]).get(timeout=300)
    # SURVEY.md §8d (ii), north_star's "reference NumPy/scipy path timed on the node's own host cores (count stated)": the NumPy +
    # scipy oracle in a process pool over images, one single-threaded worker per usable core, one image of the timed batch each
    # (per-image codebooks: images are independent units, script.py:22-38); wall clock from the first worker's start to the last
    # one's end, interpreter start-up excluded
    n_pool = min(cores, batch)
    log(f"cpu baseline: NumPy oracle, pool of {n_pool} single-threaded workers, one image each")
    with ctx.Pool(n_pool) as pool:
        pool.map(_numpy_import_only, range(n_pool))                 # workers up and modules imported before the clock starts
        t0 = time.perf_counter()
        pool.map(_numpy_worker, [(i, k, n_iter) for i in range(n_pool)], chunksize=1)
        np_pool_wall = time.perf_counter() - t0
    return ref, dict(value=round(batch * H * W / wall / 1e6, 3), unit="Mpix/s", cores=cores, kind="port",
                     seconds=round(wall, 2), host_cpu_count=os.cpu_count(), affinity_cores=affinity,
                     cgroup_cpu_quota=quota,
                     c_single_thread_mpix_s=round(H * W / c_single / 1e6, 4),
                     numpy_single_thread_mpix_s=round(H * W / np_single / 1e6, 4),
                     numpy_pool_mpix_s=round(n_pool * H * W / np_pool_wall / 1e6, 4), numpy_pool_workers=n_pool,
                     numpy_pool_note=f"oracle/spec_oracle.segment (NumPy + scipy.ndimage) on {n_pool} images of the timed batch, "
                                     f"one single-threaded worker process per usable core ({n_pool} of os.cpu_count() = {os.cpu_count()}), "
                                     f"per-image codebooks, {np_pool_wall:.1f} s wall",
                     sample=f"all {batch} synthetic {W}x{H}x3 images of the timed batch, same bank/k/n_iter, {mode} "
                            f"codebook: C restatement of SPEC.md with OpenMP on {cores} threads (affinity {affinity}, cgroup quota {quota}, "
                            f"os.cpu_count() = {os.cpu_count()}); single-thread figures: one image, run alone")


class TimedOps:
    """Wraps HipOps: HIP events (on the launch stream) around the two streaming kernels, SAMPLED.

    Each event costs ~5.7 us of stream time (rocprofv3 kernel trace: that is the gap before a kernel that follows a
    record, 0.0 us otherwise). Bracketing every Gabor call and every 11th pass made the timed region 1 % slower than
    `--no-events` (round 4, same box: 4 764 vs 4 817 Mpix/s), so the sampling is sparser now: every GABOR_STRIDE-th Gabor
    call and every PASS_STRIDE-th Lloyd pass (23 against 10 passes per step: the sampled position walks through all ten
    passes, forward and reverse sweeps alike, the LAST pass included - it goes through assign_raster, stores the raster
    label map and accumulates nothing, and its bytes are part of `a_bytes`, so it has to be part of the average too).
    20 timed steps give 5 Gabor samples and 8-9 pass samples."""
    PASS_STRIDE = 23
    GABOR_STRIDE = 4

    def __init__(self, ops, torch):
        self._ops, self._torch = ops, torch
        self.events = {"gabor": [], "assign": []}
        self.enabled = False
        self._n_pass = 0
        self._n_gabor = 0

    def __getattr__(self, name):
        return getattr(self._ops, name)

    def _timed(self, key, fn, *a, **kw):
        if not self.enabled:
            return fn(*a, **kw)
        s = self._torch.cuda.Event(enable_timing=True)
        e = self._torch.cuda.Event(enable_timing=True)
        s.record()
        fn(*a, **kw)
        e.record()
        self.events[key].append((s, e))

    def gabor_features(self, *a, **kw):
        self._n_gabor += 1
        if self._n_gabor % self.GABOR_STRIDE != 1:
            return self._ops.gabor_features(*a, **kw)
        return self._timed("gabor", self._ops.gabor_features, *a, **kw)

    def _pass(self, fn, *a, **kw):
        self._n_pass += 1
        if self._n_pass % self.PASS_STRIDE != 4:
            return fn(*a, **kw)
        return self._timed("assign", fn, *a, **kw)

    def assign_accumulate(self, *a, **kw):
        return self._pass(self._ops.assign_accumulate, *a, **kw)

    def assign_raster(self, *a, **kw):         # the last pass of every step: counted and sampled like the others
        return self._pass(self._ops.assign_raster, *a, **kw)

    # multi-rank update = reduce -> all-reduce -> finalize: one event at the start of `reduce`, one after `finalize`, for the
    # passes that are sampled anyway (the all-reduce in between goes through torch.distributed, not through these ops)
    def reduce(self, *a, **kw):
        self._trio = None
        if self.enabled and self._n_pass % self.PASS_STRIDE == 4:
            self._trio = self._torch.cuda.Event(enable_timing=True)
            self._trio.record()
        return self._ops.reduce(*a, **kw)

    def finalize(self, *a, **kw):
        r = self._ops.finalize(*a, **kw)
        if getattr(self, "_trio", None) is not None:
            e = self._torch.cuda.Event(enable_timing=True)
            e.record()
            self.events.setdefault("update_trio", []).append((self._trio, e))
            self._trio = None
        return r

    def mean_ms(self, key):
        ev = self.events.get(key, [])
        if not ev:
            return float("nan"), 0
        return sum(s.elapsed_time(e) for s, e in ev) / max(1, len(ev)), len(ev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps; the MFMA-heavy Gabor kernel needs ~10 steps (35 ms) to reach its steady clock")
    ap.add_argument("--mode", default="global", choices=["global", "per_image"])
    ap.add_argument("--batch", type=int, default=PER_GPU, help="images per GPU")
    ap.add_argument("--n-iter", type=int, default=10)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--no-other-mode", action="store_true", help="skip timing the other codebook mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--spinup-steps", type=int, default=15, help="device clock spin-up before the warm-up steps (0 = none)")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-array-in / host-array-out timing")
    ap.add_argument("--no-events", action="store_true", help="diagnostic: no per-kernel HIP events in the timed region")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a HIP device")
    dev_index = local_rank % n_dev           # one rank per GPU; the modulo only matters for gloo dry runs
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    B = args.batch
    imgs_np = synthetic_shard(rank * B, B, H, W, seed=0)      # this rank's shard of the global batch
    imgs = torch.from_numpy(imgs_np).to(dev)

    # `value` is measured on the library's own default path: ONE feature-slab allocation, no placement search
    # (Segmenter(slab_candidates=1)). The best-of-two-allocations figure is an extra key (`best_of_two_slabs_mpix_s`).
    seg = Segmenter(k=args.k, n_iter=args.n_iter, device=dev)
    tops = TimedOps(seg.ops, torch)
    seg.ops = tops
    out = torch.empty((B, H, W), dtype=torch.int32, device=dev)

    def step(mode):
        seg.segment_device(imgs, mode=mode, out=out)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    rank_dt = []

    def timed(mode, steps, warmup, events):
        for _ in range(warmup):
            step(mode)
        barrier()
        tops.enabled = events
        tops._n_pass = tops._n_gabor = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            step(mode)
        barrier()
        dt = time.perf_counter() - t0
        tops.enabled = False
        if world > 1:
            t = torch.zeros(world, dtype=torch.float64, device=dev)
            t[rank] = dt
            dist.all_reduce(t, op=dist.ReduceOp.SUM)            # every rank's time; the job's time is the slowest rank's
            rank_dt[:] = [float(x) for x in t.tolist()]
            dt = max(rank_dt)
        return dt

    # Device spin-up (not a bench step, outside every timed region): the GPU leaves its idle clock state only after
    # ~30 ms of matrix work (15 steps = ~50 ms); without it the first timed Gabor launches run ~10 % slower (0.89 vs 0.80 ms) when the
    # caller asks for very few warm-up steps.
    log(f"rank {rank}: spin-up, warm-up and {args.steps} timed steps ({args.mode})")
    for _ in range(args.spinup_steps):       # a fixed count: every rank must issue the same collectives
        step(args.mode)
    torch.cuda.synchronize(dev)
    dt = timed(args.mode, args.steps, args.warmup, events=not args.no_events)
    total_px = world * B * H * W
    value = total_px * args.steps / dt / 1e6
    slab_ms = []                                    # pass times of the two candidate slabs (extra run below)

    # per-kernel device time from the events recorded inside the timed region (rank 0's GPU)
    bank = seg.bank
    D = bank.n_features
    px = B * H * W
    g_ms, g_n = tops.mean_ms("gabor")
    a_ms, a_n = tops.mean_ms("assign")
    # Algorithmic bytes. The slab keeps pyramid level L at 1/4^L of the pixels (SPEC.md §3), so the
    # feature bytes per full-resolution pixel are 2 * sum_L D_L / 4^L (90 for the 4x6 bank) instead of 2D (144).
    # `alg_bytes` is that pyramid-resident figure (the smaller, stricter denominator); `unfused_def_bytes` is
    # SURVEY.md §8d's un-fused definition with uint16 features (3 + 2D, 2D + 1), kept for comparison with round 1.
    lv = [(3 * min(2, bank.n_scales - 2 * L) * bank.n_orient, 4 ** L) for L in range(bank.n_levels)]
    feat_b = 2 * sum(d / q for d, q in lv)
    g_bytes = (3 + feat_b) * px                      # u8 RGB in + u16 pyramid features out
    # Round 6: banks like this one keep the slab SPLIT (include/gcs.h, ABI 18): a pass streams 12 of the 16 bits of every value
    # and the last 4 only for tiles flagged "some value >= 4096" (6 of 10 016 tiles of this batch's first 16 images:
    # profiles/r6_notes.md), so the pass's algorithmic bytes are 3/4 of the feature bytes. The 16-bit figure stays beside it.
    lib = seg.ops.lib
    split = lib.gcs_feature_pass_bytes(1, H, W, bank.n_scales, bank.n_orient) < 0.8 * lib.gcs_feature_slab_bytes(1, H, W, bank.n_scales, bank.n_orient)
    pass_b = feat_b * (0.75 if split else 1.0)
    a_bytes = (pass_b + 4 / args.n_iter) * px        # per Lloyd pass: pyramid features in; the int32 raster label map is stored by the last pass only
    a_bytes_u16 = (feat_b + 4 / args.n_iter) * px    # the rounds 2-5 definition (every value read at 16 bits)
    g_ops = sum(2 * bank.ksize ** 2 * (4 * d // 3) * 3 * px / q for d, q in lv)   # int8 MACs x2: 2 digits x {re,im} x F_L rows
    kernels = {
        "gabor_mfma_kernel": dict(launches=g_n, launches_per_step=1, avg_ms=round(g_ms, 4), alg_bytes=int(g_bytes),
                                  unfused_def_bytes=(3 + 2 * D) * px, alg_ops=int(g_ops),
                                  gbs=round(g_bytes / g_ms / 1e6, 1), tops=round(g_ops / g_ms / 1e9, 1),
                                  hbm_frac=round(g_bytes / g_ms / 1e6 / HBM_PEAK_GBS, 4),
                                  mfma_frac=round(g_ops / g_ms / 1e9 / I8_MFMA_PEAK_TOPS, 4),
                                  note="whole gcs_gabor_features call: pad / pyramid kernels + one MFMA launch per level"),
        "kmeans_pass_mfma_kernel": dict(launches=a_n, launches_per_step=args.n_iter, avg_ms=round(a_ms, 4),
                                     alg_bytes=int(a_bytes), unfused_def_bytes=(2 * D + 1) * px,
                                     gbs=round(a_bytes / a_ms / 1e6, 1),
                                     hbm_frac=round(a_bytes / a_ms / 1e6 / HBM_PEAK_GBS, 4),
                                     alg_bytes_u16_def=int(a_bytes_u16),
                                     hbm_frac_u16_def=round(a_bytes_u16 / a_ms / 1e6 / HBM_PEAK_GBS, 4)),
    }
    if g_ms >= a_ms * args.n_iter:    # dominant = larger share of the step (`launches` = launches bracketed by events)
        kg = kernels["gabor_mfma_kernel"]
        # arithmetic intensity 60 840 op / 93 B = 654 op/B is above the int8 ridge (5 POP/s / 8 TB/s =
        # 625 op/B): the MFMA roof bounds this kernel; the HBM fraction is reported next to it.
        roofline = dict(kernel="gabor_mfma_kernel", bound="mfma", achieved=kg["tops"], peak=I8_MFMA_PEAK_TOPS,
                        unit="TFLOP/s", frac=kg["mfma_frac"], traffic=None, ops="int8 MAC x2 (TOP/s)",
                        hbm_gbs=kg["gbs"], hbm_frac=kg["hbm_frac"])
    else:
        ka = kernels["kmeans_pass_mfma_kernel"]
        roofline = dict(kernel="kmeans_pass_mfma_kernel", bound="hbm", achieved=ka["gbs"], peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=ka["hbm_frac"], traffic=None,
                        alg_bytes_def=("split slab (ABI 18): 1.5*sum_L D_L/4^L B/px (12 of the 16 bits of every value; the last 4 only "
                                       "for flagged tiles, ~0 here) + the label map stored by the last pass (4 B/px int32 / n_iter) = "
                                       "67.9 B/px per launch for the 4x6 bank, n_iter 10; frac_u16_def prices the same launch at the "
                                       "rounds 2-5 definition (90.4 B/px: bytes this pass no longer reads)" if split else
                                       "pyramid-resident: 2*sum_L D_L/4^L B/px read by every pass + the label map stored by the "
                                       "last one (4 B/px int32 / n_iter)"),
                        frac_u16_def=ka["hbm_frac_u16_def"])
        # the same fraction from the committed rocprofv3 --kernel-trace --stats summary of this command (tools/profile_round.sh):
        # the dominant kernel's average over ALL its launches, spin-up included
        try:
            import csv
            rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "kernel_stats_latest.csv"))))
            row = max((r for r in rows if "kmeans_pass_mfma_kernel" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
            roofline["frac_rocprof"] = round(a_bytes / (float(row["AverageNs"]) * 1e-9) / 1e9 / HBM_PEAK_GBS, 4)
            roofline["rocprof_avg_us"] = round(float(row["AverageNs"]) / 1e3, 2)
            roofline["rocprof_source"] = "profiles/kernel_stats_latest.csv (%s launches of %s)" % (row["Calls"], row["Name"].split("(")[0])
        except Exception:
            pass

    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of this same
    # command (tools/profile_round.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md). None if no profile matches.
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")))
        cand = [e for e in prof.get(roofline["kernel"], []) if e["launches"] > 0]
        if roofline["kernel"] == "gabor_mfma_kernel":
            big = max(e["grid_threads"] for e in cand)
            roofline["traffic"] = sum(e["hbm_bytes_corrected"] for e in cand if e["grid_threads"] == big) + \
                sum(e["hbm_bytes_corrected"] for e in prof.get("gabor_plane_kernel", []))
        else:
            roofline["traffic"] = max(cand, key=lambda e: e["grid_threads"])["hbm_bytes_corrected"]
        roofline["traffic_source"] = "profiles/hbm_traffic_latest.json (separate rocprofv3 --pmc run, batch %d)" % PER_GPU
    except Exception:
        pass

    extra = {}
    other = "per_image" if args.mode == "global" else "global"
    if world == 1 and not args.no_other_mode:
        # the other codebook mode (per_image = the reference's own semantics, script.py:22-38), same batch, same clock
        log(f"timing the {other} codebook mode")
        dt2 = timed(other, max(2, args.steps // 2), 2, events=False)
        extra[f"{other}_mpix_s"] = round(px * max(2, args.steps // 2) / dt2 / 1e6, 1)

    if world == 1 and not args.no_other_mode:
        # what a placement search over two slab allocations would give (Segmenter(slab_candidates=2), off in the library):
        # the same step on the better of two candidate slabs; never `value`
        log("timing the best of two slab allocations")
        seg2 = Segmenter(k=args.k, n_iter=args.n_iter, device=dev, slab_candidates=2)
        out2 = torch.empty_like(out)
        for _ in range(3):
            seg2.segment_device(imgs, mode=args.mode, out=out2)
        torch.cuda.synchronize(dev)
        n2 = max(4, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(n2):
            seg2.segment_device(imgs, mode=args.mode, out=out2)
        torch.cuda.synchronize(dev)
        extra["best_of_two_slabs_mpix_s"] = round(px * n2 / (time.perf_counter() - t0) / 1e6, 1)
        slab_ms = list(seg2.slab_placement_ms or [])
        del seg2, out2

    if world == 1 and not args.no_host_path:
        # the slot as the reference calls it (script.py:25,30): host uint8 array in, host label array out, PCIe both
        # ways. Reported beside `value`, never as `value`.
        cores_, _, _ = usable_cores()
        torch.set_num_threads(max(1, min(torch.get_num_threads(), cores_)))     # host-side copies: no more threads than cores
        log(f"host-to-host path ({torch.get_num_threads()} host threads)")
        for key, kw in (("host_to_host_mpix_s", {}), ("host_to_host_u8_mpix_s", dict(out_dtype=np.uint8))):
            for _ in range(2):
                seg.segment_batch(imgs_np, mode=args.mode, **kw)
            ts = []
            for _ in range(9):
                t0 = time.perf_counter()
                seg.segment_batch(imgs_np, mode=args.mode, **kw)
                ts.append(time.perf_counter() - t0)
            extra[key] = round(px / sorted(ts)[4] / 1e6, 1)           # median of nine calls
        # the same slot fed from a loader: segment_stream overlaps batch n+1's staging / upload and batch n-1's download
        # with batch n's compute (three streams, both copies on the copy engines); 24 consecutive batches after 6 that fill
        # the pinned-buffer caches, results handed out in order and dropped; median of three such runs
        log("pipelined host path (segment_stream)")
        rates = {"host_stream_mpix_s": [], "host_stream_u8_mpix_s": []}
        for rep in range(3):
            for key, kw in (("host_stream_mpix_s", {}), ("host_stream_u8_mpix_s", dict(out_dtype=np.uint8))):
                for _ in seg.segment_stream((imgs_np for _ in range(6)), mode=args.mode, **kw):
                    pass
                t0 = time.perf_counter()
                n_b = sum(1 for _ in seg.segment_stream((imgs_np for _ in range(24)), mode=args.mode, **kw))
                rates[key].append(round(px * n_b / (time.perf_counter() - t0) / 1e6, 1))
        for key, v in rates.items():
            extra[key] = sorted(v)[1]                              # median of three runs of 24 batches
            extra[key + "_runs"] = v
        extra["host_stream_batches"] = 24
        # the reference's own loop shape (script.py:22-38: one image per iteration) through the generator that batches per
        # image shape behind the scenes: 9 batches' worth of images, both BSD orientations mixed 2 : 1 (6 + 3 full batches,
        # so that the figure is the steady rate of a long data set, not its two remainder batches), label maps handed
        # out in input order
        log("data-set loop (segment_images)")
        turned = [np.ascontiguousarray(imgs_np[i].transpose(1, 0, 2)) for i in range(min(B, 8))]
        loop_imgs = [imgs_np[i % B] if i % 3 else turned[i % len(turned)] for i in range(9 * B)]
        for _ in seg.segment_images(loop_imgs, batch=B):
            pass
        t0 = time.perf_counter()
        n_i = sum(1 for _ in seg.segment_images(loop_imgs, batch=B))
        extra["segment_images_mpix_s"] = round(n_i * H * W / (time.perf_counter() - t0) / 1e6, 1)
        log("single-image latency")
        one = imgs_np[0]
        seg(one)
        t0 = time.perf_counter()
        for _ in range(20):
            seg(one)
        extra["single_image_latency_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)

    if world == 1 and not args.no_other_mode:
        # BASELINE config 4 (8-scale x 8-orientation bank, D = 192, four pyramid levels) on the same batch: a parity-test
        # configuration, reported beside `value` for the record (never `value`)
        log("timing BASELINE config 4 (8x8 bank)")
        seg4 = Segmenter(n_scales=8, n_orient=8, k=args.k, n_iter=args.n_iter, device=dev)
        for _ in range(5):
            seg4.segment_device(imgs, mode=args.mode, out=out)
        torch.cuda.synchronize(dev)
        n4 = max(10, args.steps // 2)                      # (3 ms per step: five steps were too short a run to be steady)
        t0 = time.perf_counter()
        for _ in range(n4):
            seg4.segment_device(imgs, mode=args.mode, out=out)
        torch.cuda.synchronize(dev)
        extra["config4_8x8_bank_mpix_s"] = round(px * n4 / (time.perf_counter() - t0) / 1e6, 1)
        del seg4

    if world == 1 and not args.no_other_mode:
        # BASELINE config 5's tile: ONE 2048 x 2048 x 3 image on one GPU, unsharded (the row-sharded form over N ranks is
        # tests/test_distributed.py; its cross-rank halo is halo_rows * 2048 * 3 bytes per interior edge and image)
        log("timing BASELINE config 5 (one 2048x2048 tile, unsharded)")
        from gabor_color_image_segmentation_amd import halo_rows
        big = torch.from_numpy(synthetic_shard(0, 1, 2048, 2048, seed=5)).to(dev)
        out5 = torch.empty((1, 2048, 2048), dtype=torch.int32, device=dev)
        for _ in range(3):
            seg.segment_device(big, mode="global", out=out5)
        torch.cuda.synchronize(dev)
        n5 = max(5, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(n5):
            seg.segment_device(big, mode="global", out=out5)
        torch.cuda.synchronize(dev)
        extra["config5_2048_tile_mpix_s"] = round(2048 * 2048 * n5 / (time.perf_counter() - t0) / 1e6, 1)
        extra["config5_halo_bytes_per_edge"] = halo_rows(bank.n_levels, bank.ksize) * 2048 * 3
        del big, out5

    if world == 1 and not args.no_other_mode:
        # the batched GPU scorer on the 24 packed BSD500 val images (device-resident label maps, ground truth
        # from the 500-id pack) and the segment + score loop of examples/bsd_eval.py --val; images per second, never `value`
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import scoring_rate
            log("scoring throughput (24 BSD val images)")
            extra["scoring"] = scoring_rate.measure(seg, reps=3)
        except Exception as e:                     # an extra, never a reason to lose the bench line (e.g. a copy without tests/golden)
            extra["scoring"] = {"skipped": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ref, cpu = cpu_baseline(B, args.k, args.n_iter, args.mode)
        if not args.no_check:
            # parity of the timed configuration itself: every label of the whole batch, timed mode, vs the C oracle
            lab = seg.segment_device(imgs, mode=args.mode).to(torch.uint8).cpu().numpy()
            cpu["labels_match_gpu"] = bool(np.array_equal(lab, ref))
            cpu["labels_compared"] = f"all {B} images, {args.mode} codebook, {lab.size} pixels"

    multi = None
    if world > 1:
        # Self-diagnosis of an N > 1 run: what carried the collectives, how even the ranks were, what one update (reduce ->
        # int64 all-reduce of k (D+1) sums -> finalize) cost on rank 0's stream, and the weak-scaling efficiency that cost
        # predicts from this run's own kernels: a rank's step is the single-rank step plus (n_iter - 1) updates.
        u_ms, u_n = tops.mean_ms("update_trio")
        step_ms = dt / args.steps * 1e3
        local_ms = g_ms + args.n_iter * a_ms                      # Gabor stage + passes, the part that does not grow with N
        multi = dict(world_size=dist.get_world_size(), backend=dist.get_backend(),
                     rank_ms_per_step=[round(x / args.steps * 1e3, 3) for x in rank_dt],
                     rank_ms_min=round(min(rank_dt) / args.steps * 1e3, 3), rank_ms_max=round(max(rank_dt) / args.steps * 1e3, 3),
                     collective_ms_per_pass=None if u_n == 0 else round(u_ms, 4), collective_samples=u_n,
                     allreduce_bytes=8 * args.k * (D + 1),
                     predicted_efficiency=None if u_n == 0 else round(local_ms / (local_ms + (args.n_iter - 1) * u_ms), 4),
                     note="collective_ms_per_pass = HIP-event time from the start of gcs_kmeans_reduce to the end of "
                          "gcs_kmeans_finalize around the all-reduce (sampled passes, rank 0); predicted_efficiency = "
                          "t_local / (t_local + (n_iter-1) * collective) with t_local = Gabor stage + n_iter passes of this run; "
                          "the driver computes the measured efficiency from the per-N values")

    if rank == 0:
        line = dict(metric="Mpix/s segmented (Gabor+k-means), 481x321x3 batch", value=round(value, 1),
                    unit="Mpix/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype="i8", data="synthetic",
                    config=dict(workload=f"batch {B}/GPU synthetic {W}x{H}x3 uint8 (seed 0), 4-scale x "
                                         f"6-orientation Gabor bank ksize {bank.ksize} on a {bank.n_levels}-level octave pyramid, k={args.k}, "
                                         f"n_iter={args.n_iter}",
                                codebook=args.mode, global_batch=world * B,
                                features="uint16 Q7, level L at 1/4^L resolution" + (
                                    "; slab split into low byte / bits 8-11 / bits 12-15, the last read per flagged tile only" if split else ""),
                                parallelism=f"dp{world} (images sharded, int64 centroid all-reduce)"
                                if args.mode == "global" else f"dp{world} (independent images)"),
                    roofline=roofline, cpu_baseline=cpu, kernels=kernels,
                    slab_placement_ms=[round(x, 4) for x in slab_ms],
                    device_cu_count=int(seg.ops.lib.gcs_device_cu_count()), multi_gpu=multi,
                    # whole job against the HBM roof: Gabor + n_iter passes, pyramid-resident bytes per pixel (and the
                    # un-fused uint16 definition of SURVEY.md §8d beside it), per GPU
                    end_to_end=dict(alg_bytes_per_px=round((3 + feat_b) + args.n_iter * pass_b + 4, 1),
                                    gbs_per_gpu=round(((3 + feat_b) + args.n_iter * pass_b + 4) * px * args.steps / dt / 1e9, 1),
                                    hbm_frac=round(((3 + feat_b) + args.n_iter * pass_b + 4) * px * args.steps / dt / 1e9
                                                   / HBM_PEAK_GBS, 4),
                                    alg_bytes_per_px_u16_def=round((3 + feat_b) + args.n_iter * feat_b + 4, 1),
                                    ),
                    **extra)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
