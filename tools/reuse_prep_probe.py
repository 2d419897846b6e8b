#!/usr/bin/env python3
"""What keeping a frame's pyramid images and polynomial expansions on the device could save at most: the same call
with and without its preparation work (environment OFXCV_DEBUG_REUSE_PREP=1, read per call: the scratch still holds the expansions of the same frames, results are
unchanged), for the call shapes of a render thread -- one pair, the two pairs of an output frame -- and a batch of 8.
usage: python tools/reuse_prep_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W, H = 1920, 1080
prs = [synth.flow_pair(W, H, seed=100 + i) for i in range(8)]
c = ofxcv.Context(0)
with torch.cuda.stream(c.stream):
    ga = [c.to_byte_grayscale(torch.from_numpy(a).cuda()) for a, _ in prs]
    gb = [c.to_byte_grayscale(torch.from_numpy(b).cuda()) for _, b in prs]
    for n in (1, 2, 8):
        res = {}
        for reuse in (0, 1, 0, 1):
            os.environ["OFXCV_DEBUG_REUSE_PREP"] = "0"
            fl = c.calc_optical_flow_farneback_batch(ga[:n], gb[:n])   # fills the scratch with these frames' expansions
            ref = [f.clone() for f in fl]
            os.environ["OFXCV_DEBUG_REUSE_PREP"] = str(reuse)
            for _ in range(3):
                c.calc_optical_flow_farneback_batch(ga[:n], gb[:n], fl)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 20
            for _ in range(K):
                c.calc_optical_flow_farneback_batch(ga[:n], gb[:n], fl)
            torch.cuda.synchronize()
            res.setdefault(reuse, []).append((time.perf_counter() - t0) / K * 1e6)
            assert all(torch.equal(x, y) for x, y in zip(ref, fl))
        a, b = min(res[0]), min(res[1])
        print("%d pair(s) per call: %8.1f us with the preparation, %8.1f us without = %.1f %% of the call (same flows)" % (n, a, b, 100 * (a - b) / a), flush=True)
c.close()
