#!/usr/bin/env python3
"""How long does the host take to enqueue one Farneback call (launch overhead) vs how long the GPU takes to run it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W, H = 1920, 1080
a, b = synth.flow_pair(W, H)
c = ofxcv.Context(0)
with torch.cuda.stream(c.stream):
    ga = c.to_byte_grayscale(torch.from_numpy(a).cuda()); gb = c.to_byte_grayscale(torch.from_numpy(b).cuda())
    fl = torch.empty((H, W, 2), device="cuda")
    for _ in range(5): c.calc_optical_flow_farneback(ga, gb, fl)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): c.calc_optical_flow_farneback(ga, gb, fl)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("enqueue %.3f ms per call, total %.3f ms per call" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
