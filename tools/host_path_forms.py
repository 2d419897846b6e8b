#!/usr/bin/env python3
"""ms per output frame of ofxcv_vectorgen_flows_host (one thread, 1920x1080) over the image layouts a host may hand over:
RGBA / RGB, contiguous / padded rows, bottom-up (negative row stride), a partial channel map."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W, H = 1920, 1080
ref, nxt = synth.flow_pair(W, H, seed=11)
prev, _ = synth.flow_pair(W, H, seed=12)
c = ofxcv.Context(0)
def pad(a):
    p = np.zeros((H, W + 16, a.shape[2]), np.float32)
    p[:, :W] = a
    return p[:, :W]
def flip(a):
    return np.ascontiguousarray(a[::-1])[::-1]
cases = {
    "RGBA, contiguous rows": (ref, nxt, prev, (1, 2, 4, 8)),
    "RGB": tuple(np.ascontiguousarray(a[..., :3]) for a in (ref, nxt, prev)) + ((1, 2, 4, 8),),
    "RGBA, padded rows": (pad(ref), pad(nxt), pad(prev), (1, 2, 4, 8)),
    "RGBA, bottom-up": (flip(ref), flip(nxt), flip(prev), (1, 2, 4, 8)),
    "RGBA, forward.u -> R only": (ref, nxt, prev, (1, 0, 0, 0)),
}
want = None
for name, (a, b, p, ch) in cases.items():
    out = np.zeros((H, W, 4), np.float32)
    for _ in range(2):
        c.vectorgen_flows_host(a, b, p, out, *ch)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); c.vectorgen_flows_host(a, b, p, out, *ch); ts.append(time.perf_counter() - t0)
    if want is None: want = out.copy()
    same = np.array_equal(out, want) if ch == (1, 2, 4, 8) else np.array_equal(out[..., 0], want[..., 0])
    print("%-30s %.2f ms per output frame (%s)" % (name, statistics.median(ts) * 1e3, "same pixels" if same else "DIFFERENT"), flush=True)
c.close()
