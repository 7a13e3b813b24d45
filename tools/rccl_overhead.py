#!/usr/bin/env python3
"""Cost of the multi-rank branch of lloyd() on ONE GPU: one-rank RCCL group, batch 64, global codebook,
collectives forced on (reduce kernel -> RCCL all-reduce -> finalize kernel per pass) vs the single-rank branch."""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
imgs = torch.from_numpy(synthetic_shard(0, 64, 321, 481)).to(dev)
seg = Segmenter(device=dev)
out = torch.empty((64, 321, 481), dtype=torch.int32, device=dev)
for force in (False, True, False, True):
    seg.debug.force_collectives = force
    for _ in range(10): seg.segment_device(imgs, mode="global", out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): seg.segment_device(imgs, mode="global", out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"force_collectives={force}: {dt*1e3:.3f} ms per step")
dist.destroy_process_group()
