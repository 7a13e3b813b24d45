#!/usr/bin/env python3
"""What state does an INVALIDATED stream capture leave behind, and what brings the thread and its streams back?
(tools/dbg/invalidated_capture_probe.py; round 6)

Thread A opens a capture (torch.cuda.graph, thread_local mode, on torch's capture stream), thread B calls hipDeviceSynchronize,
which HIP refuses and which invalidates A's capture (profiles/r5_capture_probe.txt). A then leaves the `with` block - torch calls
hipStreamEndCapture, which reports the invalidation - and this probe asks, step by step:
  * is the capture stream / the thread's current stream still `capturing`?
  * does a stream synchronize work? a kernel launch? a NEW capture?
  * does a second hipStreamEndCapture (through the runtime directly) or hipThreadExchangeStreamCaptureMode change that?
Every step is wrapped: the probe prints what happened and goes on. The graph object is kept alive (its destructor aborts)."""
import ctypes
import os
import threading

import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
x = torch.zeros(1 << 16, device=dev)
KEEP = []


def status(stream, name):
    st = ctypes.c_int(-1)
    rc = hip.hipStreamIsCapturing(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(st))
    print(f"    hipStreamIsCapturing({name}) rc={rc} status={st.value}  (0 none, 1 active, 2 invalidated)", flush=True)
    return st.value


def attempt(what, fn):
    try:
        r = fn()
        print(f"    {what}: ok {r if r is not None else ''}", flush=True)
        return True
    except Exception as e:  # noqa: BLE001
        print(f"    {what}: FAILED {repr(e).splitlines()[0][:140]}", flush=True)
        return False


def scenario(recover, mode="thread_local"):
    print(f"--- scenario: capture_error_mode = {mode}, recovery = {recover}", flush=True)
    in_capture, done = threading.Event(), threading.Event()
    cap_stream = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    KEEP.append(g)

    def b():
        in_capture.wait(30)
        attempt("B: torch.cuda.synchronize()", lambda: torch.cuda.synchronize(dev))
        done.set()

    tb = threading.Thread(target=b)
    tb.start()
    cur = torch.cuda.current_stream(dev)
    try:
        with torch.cuda.graph(g, stream=cap_stream, capture_error_mode=mode):
            x.add_(1)
            in_capture.set()
            done.wait(30)
            status(cap_stream, "capture stream, inside the with block")
            attempt("A: launch inside the invalidated capture", lambda: x.add_(1))
    except Exception as e:  # noqa: BLE001
        print(f"    A: leaving the capture raised {repr(e).splitlines()[0][:140]}", flush=True)
    tb.join()
    print(f"    hipGetLastError after the failed capture: {hip.hipGetLastError()}", flush=True)
    status(cap_stream, "capture stream")
    status(cur, "current stream")
    if recover == "end_capture_again":
        graph = ctypes.c_void_p()
        rc = hip.hipStreamEndCapture(ctypes.c_void_p(cap_stream.cuda_stream), ctypes.byref(graph))
        print(f"    second hipStreamEndCapture rc={rc} graph={graph.value}; last error {hip.hipGetLastError()}", flush=True)
        status(cap_stream, "capture stream")
    elif recover == "exchange_mode":
        mode = ctypes.c_int(2)          # hipStreamCaptureModeRelaxed
        rc = hip.hipThreadExchangeStreamCaptureMode(ctypes.byref(mode))
        print(f"    hipThreadExchangeStreamCaptureMode(relaxed) rc={rc} previous={mode.value}", flush=True)
    attempt("A: current stream synchronize", lambda: cur.synchronize())
    attempt("A: kernel launch on the current stream + synchronize", lambda: (x.add_(1), cur.synchronize())[1])
    attempt("A: event record / synchronize", lambda: (lambda e: (e.record(cur), e.synchronize()))(torch.cuda.Event()) and None)
    g2 = torch.cuda.CUDAGraph()
    KEEP.append(g2)

    def recapture():
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            x.add_(1)
        g2.replay()
        cur.synchronize()
    attempt("A: a NEW capture on torch's default capture stream + replay", recapture)
    g3 = torch.cuda.CUDAGraph()
    KEEP.append(g3)

    def recapture_same():
        with torch.cuda.graph(g3, stream=cap_stream, capture_error_mode="thread_local"):
            x.add_(1)
        g3.replay()
        cur.synchronize()
    attempt("A: a NEW capture on the SAME capture stream + replay", recapture_same)
    attempt("A: replay of the FIRST graph", lambda: (g.replay(), cur.synchronize())[1])


import sys
if len(sys.argv) > 1:                     # one capture mode per process: an invalidated capture poisons what follows it
    scenario("none", sys.argv[1])
else:
    for r in ("none", "end_capture_again", "exchange_mode"):
        scenario(r)
print("probe done; exiting without destroying the graphs", flush=True)
os._exit(0)
