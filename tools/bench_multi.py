#!/usr/bin/env python3
"""Throughput with P frame pairs in flight on P contexts/streams. usage: bench_multi.py P [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
P = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
W, H = 1920, 1080
a, b = synth.flow_pair(W, H)
ctxs = [ofxcv.Context(0) for _ in range(P)]
bufs = []
for c in ctxs:
    with torch.cuda.stream(c.stream):
        da = torch.from_numpy(a).cuda(); db = torch.from_numpy(b).cuda()
        bufs.append((da, db, torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H, W), dtype=torch.uint8, device="cuda"),
                     torch.empty((H, W, 2), device="cuda"), torch.zeros((H, W, 4), device="cuda")))
def step():
    for c, (da, db, ga, gb, fl, out) in zip(ctxs, bufs):
        with torch.cuda.stream(c.stream):
            c.to_byte_grayscale(da, ga); c.to_byte_grayscale(db, gb)
            c.calc_optical_flow_farneback(ga, gb, fl)
            c.flow_to_rgba(fl, out, 1, 2)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("P=%d: %.1f pairs/s  (%.3f ms per pair)" % (P, P * steps / el, el / steps / P * 1e3))
