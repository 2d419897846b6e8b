#!/usr/bin/env python3
"""How fast can one host thread issue the bench's per-pair call sequence (no GPU wait in the loop)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
P = 3
ctxs = [ofxcv.Context(0) for _ in range(P)]
bufs = []
for c in ctxs:
    a, b = synth.flow_pair(W, H)
    with torch.cuda.stream(c.stream):
        bufs.append(dict(a=torch.from_numpy(a).cuda(), b=torch.from_numpy(b).cuda(), ga=torch.empty((H, W), dtype=torch.uint8, device="cuda"),
                         gb=torch.empty((H, W), dtype=torch.uint8, device="cuda"), flow=torch.empty((H, W, 2), device="cuda"),
                         out=torch.zeros((H, W, 4), device="cuda")))
def step():
    for c, t in zip(ctxs, bufs):
        with torch.cuda.stream(c.stream):
            c.to_byte_grayscale(t["a"], t["ga"]); c.to_byte_grayscale(t["b"], t["gb"])
            c.calc_optical_flow_farneback(t["ga"], t["gb"], t["flow"]); c.flow_to_rgba(t["flow"], t["out"], 1, 2)
for _ in range(10): step()
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%dx%d: host issue %.0f us per pair (%.0f pairs/s issue cap), total %.0f us per pair (%.0f pairs/s)" %
      (W, H, (t1 - t0) / (n * P) * 1e6, n * P / (t1 - t0), (t2 - t0) / (n * P) * 1e6, n * P / (t2 - t0)))
