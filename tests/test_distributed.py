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


@pytest.mark.parametrize("world,per_rank", [(2, 2), (2, 1)])
def test_global_codebook_two_ranks_equals_unsharded_oracle(tmp_path, world, per_rank):
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    from oracle import spec_oracle as so
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, per_rank, 4, 6, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"labels_{r}.npy") for r in range(world)])
    ref = so.segment_batch(synthetic_batch(world * per_rank, 24, 40, seed=5), mode="global", k=6, n_iter=4)
    assert np.array_equal(got, ref)
