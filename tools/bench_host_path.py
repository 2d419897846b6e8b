#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of ofxcv_vectorgen_flow_host: f32 RGBA host frames in, flow written into a host RGBA image."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W, H = 1920, 1080
a, b = synth.flow_pair(W, H)
def run(nthreads, n=8):
    ctxs = [ofxcv.Context(0) for _ in range(nthreads)]
    outs = [np.zeros((H, W, 4), np.float32) for _ in range(nthreads)]
    for c, o in zip(ctxs, outs): c.vectorgen_flow_host(a, b, o, 1, 2)
    def work(c, o):
        for _ in range(n): c.vectorgen_flow_host(a, b, o, 1, 2)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(c, o)) for c, o in zip(ctxs, outs)]
    [t.start() for t in th]; [t.join() for t in th]
    el = time.perf_counter() - t0
    for c in ctxs: c.close()
    return nthreads * n / el
for nt in (1, 2, 4):
    print("host path, %d calling threads: %.1f pairs/s" % (nt, run(nt)))

# a default VectorGenerator output frame: forward + backward flow of one reference frame
prev, ref = synth.flow_pair(W, H, seed=11)
_, nxt = synth.flow_pair(W, H, seed=12)
c = ofxcv.Context(0)
out = np.zeros((H, W, 4), np.float32)
for combined in (False, True):
    def frame():
        if combined:
            c.vectorgen_flows_host(ref, nxt, prev, out, 1, 2, 4, 8)
        else:
            c.vectorgen_flow_host(ref, nxt, out, 1, 2); c.vectorgen_flow_host(ref, prev, out, 4, 8)
    frame(); frame()
    t0 = time.perf_counter(); n = 10
    for _ in range(n): frame()
    el = (time.perf_counter() - t0) / n
    print("default output frame (2 flows), %s: %.2f ms = %.0f frames/s" % ("one combined call" if combined else "two calls", el * 1e3, 1 / el))
