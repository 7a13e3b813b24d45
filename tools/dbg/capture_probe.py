#!/usr/bin/env python3
"""Which host calls does the HIP runtime refuse in thread B while thread A holds an open stream capture in thread_local mode?
(tools/dbg/capture_probe.py; the answers shape segmenter._CaptureGuard: what a non-capturing thread may do beside a capture.)
Every probe runs in thread B on a stream of its own; A's capture is checked afterwards (an illegal call can invalidate it)."""
import threading
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
x = torch.zeros(1 << 16, device=dev)
side = torch.cuda.Stream()
evt = torch.cuda.Event()


def probes():
    s = torch.cuda.Stream()
    y = torch.zeros(1 << 16, device=dev)
    return [
        ("kernel launch on own stream", lambda: (torch.cuda.set_stream(s), y.add_(1))),
        ("own-stream synchronize", lambda: s.synchronize()),
        ("event record + synchronize", lambda: (evt.record(s), evt.synchronize())),
        ("event query", lambda: evt.query()),
        ("hipMalloc (new 64 MB block)", lambda: torch.empty(64 << 20, dtype=torch.uint8, device=dev)),
        ("pinned alloc", lambda: torch.empty(1 << 20, dtype=torch.uint8, pin_memory=True)),
        ("async H2D from pinned", lambda: y.copy_(torch.empty(1 << 16, pin_memory=True), non_blocking=True)),
        ("null-stream synchronize", lambda: torch.cuda.default_stream(dev).synchronize()),
        ("device synchronize", lambda: torch.cuda.synchronize(dev)),
        ("empty_cache (hipFree)", lambda: torch.cuda.empty_cache()),
    ]


for name, fn in probes():
    in_capture, done = threading.Event(), threading.Event()
    res = {}

    def a():
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                x.add_(1)
                in_capture.set()
                done.wait(30)
                x.add_(1)
            g.replay()
            torch.cuda.current_stream().synchronize()
            res["a"] = "capture ok"
        except Exception as e:  # noqa: BLE001
            res["a"] = "capture FAILED: " + repr(e).split("\\n")[0][:90]
            in_capture.set()

    def b():
        in_capture.wait(30)
        try:
            fn()
            res["b"] = "ok"
        except Exception as e:  # noqa: BLE001
            res["b"] = "REFUSED: " + repr(e).split("\\n")[0][:90]
        done.set()

    ta, tb = threading.Thread(target=a), threading.Thread(target=b)
    ta.start(); tb.start(); ta.join(); tb.join()
    print(f"{name:32s} B: {res.get('b')}   | A: {res.get('a')}", flush=True)
