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


def _oracle_worker(args):
    """Runs in a spawned process: oracle only, never touches the GPU."""
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
    lab = spec_oracle.segment(img, k=k, n_iter=n_iter)
    return idx, _t.perf_counter() - t0, lab


def cpu_baseline(k, n_iter):
    """The oracle (NumPy/scipy port of SPEC.md) timed on the host: one worker process per core,
    one image of the same synthetic batch per worker (bounded sample), plus the single-thread rate.
    It is the checker, timed as the reported CPU baseline only. Workers are *spawned* (no fork after
    HIP initialisation) and never import torch.cuda."""
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 16))
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        pool.map(_oracle_worker, [(0, k, 1)] * cores)            # warm the workers (imports, page-in)
        t0 = time.perf_counter()
        res = pool.map(_oracle_worker, [(i, k, n_iter) for i in range(cores)])
        wall = time.perf_counter() - t0
    single = sum(r[1] for r in res) / len(res)
    ref0 = [r[2] for r in res if r[0] == 0][0]
    return ref0, dict(value=round(cores * H * W / wall / 1e6, 4), unit="Mpix/s", cores=cores, kind="port",
                      seconds=round(wall, 2), single_thread_mpix_s=round(H * W / single / 1e6, 5),
                      sample=f"images 0..{cores - 1} of the {PER_GPU} synthetic {W}x{H}x3 images, same bank/k/n_iter, "
                             f"NumPy+scipy.ndimage oracle, one single-threaded worker process per core "
                             f"({cores} of {os.cpu_count()} host cores)")


class TimedOps:
    """Wraps HipOps: HIP events (on the launch stream) around the two streaming kernels.

    Every Gabor launch is bracketed; of the Lloyd passes every PASS_STRIDE-th launch is (stride 11 against 10
    passes per step: the sampled position walks through all ten passes, forward and reverse sweeps alike).
    Each event costs ~5.7 us of stream time (rocprofv3 kernel trace: that is the gap before a kernel that
    follows a record, 0.0 us otherwise); bracketing all 11 launches of a step made the step 2 % slower."""
    PASS_STRIDE = 11

    def __init__(self, ops, torch):
        self._ops, self._torch = ops, torch
        self.events = {"gabor": [], "assign": []}
        self.enabled = False
        self._n_pass = 0

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
        return self._timed("gabor", self._ops.gabor_features, *a, **kw)

    def assign_accumulate(self, *a, **kw):
        self._n_pass += 1
        if self._n_pass % self.PASS_STRIDE != 4:
            return self._ops.assign_accumulate(*a, **kw)
        return self._timed("assign", self._ops.assign_accumulate, *a, **kw)

    def mean_ms(self, key):
        ev = self.events[key]
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
    ap.add_argument("--also-other-mode", action="store_true", help="also time the other codebook mode (extra key)")
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

    def timed(mode, steps, warmup, events):
        for _ in range(warmup):
            step(mode)
        barrier()
        tops.enabled = events
        tops._n_pass = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            step(mode)
        barrier()
        dt = time.perf_counter() - t0
        tops.enabled = False
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # Device spin-up (not a bench step, outside every timed region): the GPU leaves its idle clock state only after
    # ~30 ms of matrix work (15 steps = ~50 ms); without it the first timed Gabor launches run ~10 % slower (0.89 vs 0.80 ms) when the
    # caller asks for very few warm-up steps.
    for _ in range(args.spinup_steps):       # a fixed count: every rank must issue the same collectives
        step(args.mode)
    torch.cuda.synchronize(dev)
    dt = timed(args.mode, args.steps, args.warmup, events=not args.no_events)
    total_px = world * B * H * W
    value = total_px * args.steps / dt / 1e6

    # per-kernel device time from the events recorded inside the timed region (rank 0's GPU)
    D = seg.bank.n_features
    F = seg.bank.n_filters
    px = B * H * W
    g_ms, g_n = tops.mean_ms("gabor")
    a_ms, a_n = tops.mean_ms("assign")
    g_bytes = (3 + 2 * D) * px                       # u8 RGB in + u16 features out (SPEC.md §6)
    a_bytes = (2 * D + 1) * px                       # u16 features in + u8 label out, per Lloyd pass
    g_ops = 2 * seg.bank.ksize ** 2 * (4 * F) * 3 * px   # int8 MACs x2: 2 digits x {re,im} x F rows, 3 channels
    kernels = {
        "gabor_mfma_kernel": dict(launches=g_n, launches_per_step=1, avg_ms=round(g_ms, 4), alg_bytes=g_bytes, alg_ops=g_ops,
                                  gbs=round(g_bytes / g_ms / 1e6, 1), tops=round(g_ops / g_ms / 1e9, 1),
                                  hbm_frac=round(g_bytes / g_ms / 1e6 / HBM_PEAK_GBS, 4),
                                  mfma_frac=round(g_ops / g_ms / 1e9 / I8_MFMA_PEAK_TOPS, 4)),
        "kmeans_pass_mfma_kernel": dict(launches=a_n, launches_per_step=args.n_iter, avg_ms=round(a_ms, 4), alg_bytes=a_bytes,
                                     gbs=round(a_bytes / a_ms / 1e6, 1),
                                     hbm_frac=round(a_bytes / a_ms / 1e6 / HBM_PEAK_GBS, 4)),
    }
    if g_ms >= a_ms * args.n_iter:    # dominant = larger share of the step (`launches` = launches bracketed by events)
        kg = kernels["gabor_mfma_kernel"]
        # arithmetic intensity 2*225*96*3/147 = 881 op/B is above the int8 ridge (5 POP/s / 8 TB/s =
        # 625 op/B): the MFMA roof bounds this kernel; the HBM fraction is reported next to it.
        roofline = dict(kernel="gabor_mfma_kernel", bound="mfma", achieved=kg["tops"], peak=I8_MFMA_PEAK_TOPS,
                        unit="TFLOP/s", frac=kg["mfma_frac"], traffic=None, ops="int8 MAC x2 (TOP/s)",
                        hbm_gbs=kg["gbs"], hbm_frac=kg["hbm_frac"])
    else:
        ka = kernels["kmeans_pass_mfma_kernel"]
        roofline = dict(kernel="kmeans_pass_mfma_kernel", bound="hbm", achieved=ka["gbs"], peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=ka["hbm_frac"], traffic=None)

    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of this same
    # command (tools/profile_round.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md). None if no profile matches.
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")))
        cand = [e for e in prof.get(roofline["kernel"], []) if e["launches"] > 0]
        if roofline["kernel"] == "gabor_mfma_kernel":
            big = max(e["grid_threads"] for e in cand)
            roofline["traffic"] = sum(e["hbm_bytes_corrected"] for e in cand if e["grid_threads"] == big) + \
                sum(e["hbm_bytes_corrected"] for e in prof.get("gabor_pad_kernel", [])
                    if e["grid_threads"] == max(x["grid_threads"] for x in prof["gabor_pad_kernel"]))
        else:
            roofline["traffic"] = max(cand, key=lambda e: e["grid_threads"])["hbm_bytes_corrected"]
        roofline["traffic_source"] = "profiles/hbm_traffic_latest.json (separate rocprofv3 --pmc run, batch %d)" % PER_GPU
    except Exception:
        pass

    extra = {}
    other = "per_image" if args.mode == "global" else "global"
    if world == 1 and args.also_other_mode:
        dt2 = timed(other, max(1, args.steps // 2), 1, events=False)
        extra[f"{other}_mpix_s"] = round(px * max(1, args.steps // 2) / dt2 / 1e6, 1)

    if world == 1 and not args.no_host_path:
        # the slot as the reference calls it: host uint8 array in, host int32 labels out (pageable memory,
        # H2D 29.6 MB + D2H 39.5 MB per batch over PCIe). Reported beside `value`, never as `value`.
        seg.segment_batch(imgs_np, mode=args.mode)
        t0 = time.perf_counter()
        for _ in range(2):
            seg.segment_batch(imgs_np, mode=args.mode)
        extra["host_to_host_mpix_s"] = round(px * 2 / (time.perf_counter() - t0) / 1e6, 1)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ref, cpu = cpu_baseline(args.k, args.n_iter)
        if not args.no_check:
            # parity of the timed configuration itself: per-image labels of image 0 vs the oracle
            lab = seg.segment_device(imgs[:1], mode="per_image").cpu().numpy()[0]
            cpu["labels_match_gpu"] = bool(np.array_equal(lab, ref))

    if rank == 0:
        line = dict(metric="Mpix/s segmented (Gabor+k-means), 481x321x3 batch", value=round(value, 1),
                    unit="Mpix/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype="i8", data="synthetic",
                    config=dict(workload=f"batch {B}/GPU synthetic {W}x{H}x3 uint8 (seed 0), 4-scale x "
                                         f"6-orientation Gabor bank ksize 15, k={args.k}, n_iter={args.n_iter}",
                                codebook=args.mode, global_batch=world * B, features="uint16 Q7",
                                parallelism=f"dp{world} (images sharded, int64 centroid all-reduce)"
                                if args.mode == "global" else f"dp{world} (independent images)"),
                    roofline=roofline, cpu_baseline=cpu, kernels=kernels,
                    # whole job against the HBM roof, un-fused definition of SURVEY.md §8d with the uint16 feature
                    # denominators: (3 + 2D) + n_iter * (2D + 1) bytes per pixel, per GPU
                    end_to_end=dict(alg_bytes_per_px=(3 + 2 * D) + args.n_iter * (2 * D + 1),
                                    gbs_per_gpu=round(((3 + 2 * D) + args.n_iter * (2 * D + 1)) * px * args.steps / dt / 1e9, 1),
                                    hbm_frac=round(((3 + 2 * D) + args.n_iter * (2 * D + 1)) * px * args.steps / dt / 1e9
                                                   / HBM_PEAK_GBS, 4)),
                    **extra)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
