#!/usr/bin/env python3
"""Whole-call time of Farneback pairs over frame sizes (alone / in a batch of 8), as ns per pixel: looks for sizes that fall off
the tuned paths (widths that are not multiples of 64, odd sizes, small frames)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
c = ofxcv.Context(0)
for (W, H) in ((640, 360), (1280, 720), (1920, 1080), (1921, 1081), (1998, 1080), (2048, 1152), (2048, 858), (3840, 2160), (4096, 2160)):
    prs = [synth.flow_pair(W, H, seed=100 + i) for i in range(2)]
    with torch.cuda.stream(c.stream):
        ga = [c.to_byte_grayscale(torch.from_numpy(prs[i % 2][0]).cuda()) for i in range(8)]
        gb = [c.to_byte_grayscale(torch.from_numpy(prs[i % 2][1]).cuda()) for i in range(8)]
        res = []
        for nb in (1, 8):
            fl = c.calc_optical_flow_farneback_batch(ga[:nb], gb[:nb])
            for _ in range(2):
                c.calc_optical_flow_farneback_batch(ga[:nb], gb[:nb], fl)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                c.calc_optical_flow_farneback_batch(ga[:nb], gb[:nb], fl)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 4 / nb)
    print("%4dx%-4d: %7.3f ms per pair alone (%5.2f ns/px), %7.3f ms per pair in a batch of 8 (%5.2f ns/px)" %
          (W, H, res[0] * 1e3, res[0] * 1e9 / (W * H), res[1] * 1e3, res[1] * 1e9 / (W * H)), flush=True)
c.close()
