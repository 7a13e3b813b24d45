#!/usr/bin/env python3
"""Is the damage of an invalidated stream capture bound to the THREAD that captured? (round 6; follows invalidated_capture_probe.py.)

A helper thread H opens a capture; thread B's hipDeviceSynchronize invalidates it; H ends. Then the MAIN thread - which never
captured - launches kernels, synchronizes, and tries a capture of its own (in a second helper thread too)."""
import os
import threading

import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
x = torch.zeros(1 << 16, device=dev)
KEEP = []


def attempt(what, fn):
    try:
        fn()
        print(f"    {what}: ok", flush=True)
        return True
    except Exception as e:  # noqa: BLE001
        print(f"    {what}: FAILED {repr(e).splitlines()[0][:150]}", flush=True)
        return False


def capture_in_helper(disturb):
    in_capture, done, res = threading.Event(), threading.Event(), {}
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    KEEP.extend([g, s])

    def h():
        torch.cuda.set_device(dev)
        try:
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                x.add_(1)
                in_capture.set()
                done.wait(30)
                x.add_(1)
            res["ok"] = True
        except Exception as e:  # noqa: BLE001
            res["ok"] = False
            res["err"] = repr(e).splitlines()[0][:150]
            in_capture.set()

    def b():
        in_capture.wait(30)
        if disturb:
            attempt("B: torch.cuda.synchronize()", lambda: torch.cuda.synchronize(dev))
        done.set()

    th, tb = threading.Thread(target=h), threading.Thread(target=b)
    th.start(); tb.start(); th.join(); tb.join()
    print(f"    helper capture (disturbed={disturb}): {res}", flush=True)
    return g if res.get("ok") else None


print("--- an undisturbed capture in a helper thread, replayed from the main thread", flush=True)
g0 = capture_in_helper(False)
attempt("main: replay of the helper's graph + synchronize", lambda: (g0.replay(), torch.cuda.current_stream().synchronize()))
print("--- a capture in a helper thread, invalidated by B", flush=True)
capture_in_helper(True)
attempt("main: kernel launch + stream synchronize", lambda: (x.add_(1), torch.cuda.current_stream().synchronize()))
attempt("main: torch.cuda.synchronize()", lambda: torch.cuda.synchronize(dev))
attempt("main: replay of the FIRST (good) graph", lambda: (g0.replay(), torch.cuda.current_stream().synchronize()))
print("--- afterwards: another undisturbed capture in a fresh helper thread", flush=True)
g2 = capture_in_helper(False)
if g2 is not None:
    attempt("main: replay of the new graph", lambda: (g2.replay(), torch.cuda.current_stream().synchronize()))
print(f"x[0] = {float(x[0].item())}", flush=True)
print("probe done", flush=True)
os._exit(0)
