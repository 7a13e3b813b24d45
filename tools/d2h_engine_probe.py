"""Which engine carries a device-to-host copy on this stack? Variants of the HIP copy calls on a 39.5 MB buffer, each timed
alone; run under `rocprofv3 --kernel-trace --memory-copy-trace` the kernel trace shows which of them became blit kernels.

    python tools/d2h_engine_probe.py
"""
import ctypes as C
import time

import torch

hip = C.CDLL("libamdhip64.so")
vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
hip.hipMemcpyAsync.argtypes = [vp, vp, sz, i32, vp]
hip.hipMemcpyDtoHAsync.argtypes = [vp, vp, sz, vp]
hip.hipMemcpy2DAsync.argtypes = [vp, sz, vp, sz, sz, sz, i32, vp]
D2H = 2
B, H, W = 64, 321, 481
n = B * H * W * 4
src = torch.zeros(n + (1 << 20), dtype=torch.uint8, device="cuda")
dst = torch.empty(n + (1 << 20), dtype=torch.uint8).pin_memory()
s = torch.cuda.Stream()
marker = torch.zeros(1024, device="cuda")


def run(name, f):
    ts = []
    for _ in range(5):
        with torch.cuda.stream(s):
            marker.add_(1)                      # a kernel in front, as in the pipeline
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = f(s.cuda_stream)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    print("%-44s rc=%d  %.3f ms  (%.1f GB/s)" % (name, rc, sorted(ts)[2] * 1e3, n / sorted(ts)[2] / 1e9), flush=True)


run("hipMemcpyAsync D2H", lambda st: hip.hipMemcpyAsync(dst.data_ptr(), src.data_ptr(), n, D2H, st))
run("hipMemcpyDtoHAsync", lambda st: hip.hipMemcpyDtoHAsync(dst.data_ptr(), src.data_ptr(), n, st))
run("hipMemcpy2DAsync width == pitch", lambda st: hip.hipMemcpy2DAsync(dst.data_ptr(), W * 4, src.data_ptr(), W * 4, W * 4, B * H, D2H, st))
run("hipMemcpy2DAsync src pitch = width + 64", lambda st: hip.hipMemcpy2DAsync(dst.data_ptr(), H * W * 4, src.data_ptr(), H * W * 4 + 64, H * W * 4, B, D2H, st))
run("hipMemcpy2DAsync both pitches = width + 64", lambda st: hip.hipMemcpy2DAsync(dst.data_ptr(), H * W * 4 + 64, src.data_ptr(), H * W * 4 + 64, H * W * 4, B, D2H, st))
run("torch copy_ non_blocking", lambda st: (dst[:n].copy_(src[:n], non_blocking=True), 0)[1])
H2D = 1
run("hipMemcpyAsync H2D (for comparison)", lambda st: hip.hipMemcpyAsync(src.data_ptr(), dst.data_ptr(), n, H2D, st))
