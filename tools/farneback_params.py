#!/usr/bin/env python3
"""Whole-call time of one 1920x1080 Farneback pair (and a batch of 8) over the parameters the VectorGenerator plugin exposes
(levels, iterations, neighborhood = poly_n, sigma): looks for parameter values that fall off the tuned paths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W, H = 1920, 1080
c = ofxcv.Context(0)
prs = [synth.flow_pair(W, H, seed=100 + i) for i in range(8)]
with torch.cuda.stream(c.stream):
    ga = [c.to_byte_grayscale(torch.from_numpy(a).cuda()) for a, _ in prs]
    gb = [c.to_byte_grayscale(torch.from_numpy(b).cuda()) for _, b in prs]
    for (levels, iters, poly_n, sigma) in ((3, 15, 5, 1.1), (3, 15, 7, 1.5), (1, 15, 5, 1.1), (5, 15, 5, 1.1), (6, 15, 5, 1.1), (3, 3, 5, 1.1), (3, 1, 5, 1.1), (3, 30, 5, 1.1), (3, 15, 3, 0.8), (3, 15, 9, 2.0)):
        res = []
        for nb in (1, 8):
            fl = c.calc_optical_flow_farneback_batch(ga[:nb], gb[:nb], None, 0.5, levels, 3, iters, poly_n, sigma, 0)
            for _ in range(2):
                c.calc_optical_flow_farneback_batch(ga[:nb], gb[:nb], fl, 0.5, levels, 3, iters, poly_n, sigma, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                c.calc_optical_flow_farneback_batch(ga[:nb], gb[:nb], fl, 0.5, levels, 3, iters, poly_n, sigma, 0)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 5 / nb * 1e3)
        print("levels %d iterations %2d poly_n %d sigma %.1f: %.3f ms per pair alone, %.3f ms per pair in a batch of 8" % (levels, iters, poly_n, sigma, res[0], res[1]), flush=True)
c.close()
