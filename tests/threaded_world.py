"""N ranks as N THREADS of one process (torch's own thread-backed process group, the one its DTensor tests use).

Why: a GPU box admits at most six processes on its card (gpurun's process guard), and the target machine has eight GPUs. With
this helper all eight ranks of a test run THROUGH THE HIP KERNELS on cuda:0 in one process; `backend == "threaded"` is not
gloo, so `segmenter._collective` hands the DEVICE tensors to the collective (the path RCCL takes on a multi-GPU node) instead
of a host copy. The backend implements the collectives (all_reduce, broadcast, all_gather, barrier), not point-to-point:
the rank-to-rank halo exchange is tested with real processes (tests/test_distributed.py).
Test infrastructure only."""
import threading
import traceback

import torch
import torch.distributed as dist


def run_threaded(world, fn, timeout=900):
    """Run ``fn(rank, world)`` on ``world`` threads, each an initialised rank of one thread-backed group; returns the results
    in rank order. Any rank's exception fails the call (with every traceback)."""
    from torch.testing._internal.distributed.multi_threaded_pg import (ProcessLocalGroup, _install_threaded_pg,
                                                                       _uninstall_threaded_pg)
    _install_threaded_pg()
    torch._C._distributed_c10d._set_thread_isolation_mode(True)
    store = dist.HashStore()
    errs, out = [], {}

    def worker(rank):
        try:
            dist.init_process_group(backend="threaded", rank=rank, world_size=world, store=store)
            out[rank] = fn(rank, world)
        except BaseException as ex:  # noqa: B036  (wake the other ranks, then report)
            errs.append((rank, traceback.format_exc()))
            ProcessLocalGroup.exception_handle(ex)
        finally:
            try:
                dist.destroy_process_group()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(world)]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout)
        hung = [i for i, t in enumerate(threads) if t.is_alive()]
    finally:
        ProcessLocalGroup.reset()
        _uninstall_threaded_pg()
        torch._C._distributed_c10d._set_thread_isolation_mode(False)
    assert not hung, f"ranks {hung} did not finish in {timeout} s"
    assert not errs, "\n".join(f"rank {r}:\n{tb}" for r, tb in errs)
    return [out[r] for r in range(world)]
