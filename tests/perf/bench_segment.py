#!/usr/bin/env python3
"""Times pyramid mean-shift filtering at BASELINE config 4 (3840x2160, sp 10, sr 20, maxLevel 2) and the oracle on a crop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
from oracle import binding as oracle
ctx = ofxcv.Context(0)
for (w, h) in [(1920, 1080), (3840, 2160)]:
    fr = np.ascontiguousarray(synth.inpaint_frame(w, h, n_holes=0)[..., :3])
    d = torch.from_numpy(fr).cuda()
    ctx.pyr_mean_shift_filtering(d); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n): ctx.pyr_mean_shift_filtering(d)
    torch.cuda.synchronize()
    g = (time.perf_counter() - t0) / n
    crop = np.ascontiguousarray(fr[:540, :960])
    t0 = time.perf_counter(); oracle.pyr_mean_shift(crop); c = time.perf_counter() - t0
    print("%dx%d: GPU %.1f ms (%.1f Mpx/s); CPU oracle on a 960x540 crop %.2f s (%.2f Mpx/s, 1 thread)" % (w, h, g * 1e3, w * h / g / 1e6, c, 960 * 540 / c / 1e6))
