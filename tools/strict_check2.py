#!/usr/bin/env python3
"""Unrelated frame pair (chaotic flow): OpenCV-order modes vs the FAITHFUL oracle, per level count / iteration count."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
from oracle import binding as oracle
w, h = 320, 240
a, _ = synth.flow_pair(w, h)
c, _ = synth.flow_pair(w, h, seed=77)
ga, gc = oracle.to_byte_grayscale(a), oracle.to_byte_grayscale(c)
for levels, iters in [(0, 1), (0, 2), (0, 4), (0, 8), (0, 15), (1, 15), (3, 15)]:
    ref = oracle.calc_optical_flow_farneback(ga, gc, levels=levels, iterations=iters, blur_mode=oracle.BLUR_FAITHFUL)
    for mode in (2, 1):
        ctx = ofxcv.Context(0)
        ctx.set_option("farneback.opencv_rounding", mode)
        got = ctx.calc_optical_flow_farneback(torch.from_numpy(ga).cuda(), torch.from_numpy(gc).cuda(), levels=levels, iterations=iters).cpu().numpy()
        ctx.close()
        err = np.abs(ref - got)
        bad = err > 1e-4 * np.maximum(1, np.abs(ref))
        ys, xs = np.nonzero(bad.any(axis=2))
        print("levels %d iters %2d mode %d: max err %.3g, outside %d, differing samples %d, first bad %s" % (
            levels, iters, mode, err.max(), bad.sum(), (ref != got).sum(), list(zip(ys[:4], xs[:4]))), flush=True)
