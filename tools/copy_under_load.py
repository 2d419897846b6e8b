#!/usr/bin/env python3
"""How long one 33 MB host <-> device copy takes while ANOTHER thread keeps the GPU busy with batched Farneback calls of 8 pairs:
pageable and pinned, upload and download, idle GPU vs loaded GPU.  (Why a render thread's upload phase stretched from 0.7 to 2 - 6 ms
under the submission queue: profiles/r06_experiments.md.)
usage: python tools/copy_under_load.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W, H = 1920, 1080
a, b = synth.flow_pair(W, H)
c = ofxcv.Context(0)
ga = c.to_byte_grayscale(torch.from_numpy(a).cuda()); gb = c.to_byte_grayscale(torch.from_numpy(b).cuda())
flows = [torch.empty((H, W, 2), device="cuda") for _ in range(8)]
stop = threading.Event()
def load():
    while not stop.is_set():
        with torch.cuda.stream(c.stream):
            c.calc_optical_flow_farneback_batch([ga] * 8, [gb] * 8, flows)
        c.synchronize()
host_pg = torch.from_numpy(a.copy()); host_pin = torch.from_numpy(a.copy()).pin_memory()
dev = torch.empty_like(host_pg, device="cuda")
s = torch.cuda.Stream()
def timed(fn, n=40):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            fn()
        s.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return "median %.2f ms, p90 %.2f, max %.2f" % (ts[len(ts) // 2], ts[int(len(ts) * .9)], ts[-1])
legs = [("pageable H2D", lambda: dev.copy_(host_pg, non_blocking=True)), ("pinned   H2D", lambda: dev.copy_(host_pin, non_blocking=True)),
        ("pageable D2H", lambda: host_pg.copy_(dev, non_blocking=True)), ("pinned   D2H", lambda: host_pin.copy_(dev, non_blocking=True))]
for name, fn in legs:
    print("idle GPU   %s 33 MB: %s" % (name, timed(fn)), flush=True)
th = threading.Thread(target=load); th.start(); time.sleep(0.3)
for name, fn in legs:
    print("loaded GPU %s 33 MB: %s" % (name, timed(fn)), flush=True)
# a small kernel on another stream while the batched call runs: how long until it gets its turn
x = torch.zeros(1 << 20, device="cuda")
print("loaded GPU small kernel (4 MB fill) on another stream: %s" % timed(lambda: x.add_(1.0)), flush=True)
stop.set(); th.join()
print("idle GPU   small kernel: %s" % timed(lambda: x.add_(1.0)), flush=True)
c.close()
