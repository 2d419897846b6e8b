#!/usr/bin/env python3
"""Batched Farneback call of n different pairs, synchronised after each call (what the submission queue's leader does): replayed from the
captured hipGraph vs launched eagerly; wall time per call and the host time the enqueue takes (the GPU idles for most of a hipGraphLaunch:
ROCm submits a graph's packets at the end of the call; eager launches start executing at once).  usage: python tools/graph_vs_eager.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W, H = 1920, 1080
prs = []
c = ofxcv.Context(0)
for i in range(8):
    a, b = synth.flow_pair(W, H, seed=1234 + i)
    prs.append((c.to_byte_grayscale(torch.from_numpy(a).cuda()), c.to_byte_grayscale(torch.from_numpy(b).cuda())))
for n in (1, 2, 4, 6, 8):
    flows = [torch.empty((H, W, 2), device="cuda") for _ in range(n)]
    for graph in (1, 0):
        c.set_option("farneback.graph", graph)
        ga = [p[0] for p in prs[:n]]; gb = [p[1] for p in prs[:n]]
        for _ in range(3):
            with torch.cuda.stream(c.stream):
                c.calc_optical_flow_farneback_batch(ga, gb, flows)
            c.synchronize()
        enq = 0.0; k = 0; t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.7:
            t1 = time.perf_counter()
            with torch.cuda.stream(c.stream):
                c.calc_optical_flow_farneback_batch(ga, gb, flows)
            enq += time.perf_counter() - t1
            c.synchronize(); k += 1
        el = time.perf_counter() - t0
        print("%d pairs, %s: %.2f ms per call (%.0f pairs/s), enqueue %.2f ms of it on the host" % (n, "graph replay" if graph else "eager launches", el / k * 1e3, n * k / el, enq / k * 1e3), flush=True)
c.close()
