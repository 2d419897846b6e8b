#!/usr/bin/env python3
"""ofxcv_vectorgen_flows_host (default output frame = 2 flows) with 1 / 2 / 4 / 8 calling threads, one context each."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W, H = 1920, 1080
ref, nxt = synth.flow_pair(W, H, seed=11)
prev, _ = synth.flow_pair(W, H, seed=12)
for nt in (1, 2, 4, 8):
    ctxs = [ofxcv.Context(0) for _ in range(nt)]
    for kv in filter(None, os.environ.get("BENCH_CTX_OPTIONS", "").split(",")):
        for c in ctxs:
            c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    # every thread its own frames (as a host would hand them)
    frames = [(ref.copy(), nxt.copy(), prev.copy(), np.zeros((H, W, 4), np.float32)) for _ in range(nt)]
    for c, f in zip(ctxs, frames):
        c.vectorgen_flows_host(f[0], f[1], f[2], f[3], 1, 2, 4, 8)
    n = 8
    def work(c, f):
        for _ in range(n):
            c.vectorgen_flows_host(f[0], f[1], f[2], f[3], 1, 2, 4, 8)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(c, f)) for c, f in zip(ctxs, frames)]
    [t.start() for t in th]; [t.join() for t in th]
    el = time.perf_counter() - t0
    print("%d calling threads: %.1f output frames/s = %.0f pairs/s (%.2f ms per call per thread)" % (nt, nt * n / el, 2 * nt * n / el, el / n * 1e3), flush=True)
    for c in ctxs:
        c.close()
