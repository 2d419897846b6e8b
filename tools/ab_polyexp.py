#!/usr/bin/env python3
"""A/B of the polyexp kernel variants at 1080p and 540p: equality with variant 0 and time per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openfx_opencv_amd as ofxcv
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6]
for (w, h) in [(1920, 1080), (960, 540), (333, 257)]:
    torch.manual_seed(1)
    img = (torch.rand((h, w), device="cuda") * 255).contiguous()
    base = None
    for v in variants:
        c = ofxcv.Context(0)
        c.set_option("farneback.polyexp_variant", v)
        with torch.cuda.stream(c.stream):
            r = c.farneback_polyexp(img, 5, 1.1).clone()
            torch.cuda.synchronize()
            base = r if base is None else base
            same = bool(torch.equal(base, r))
            for _ in range(5): c.farneback_polyexp(img, 5, 1.1)
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 200
            for _ in range(n): c.farneback_polyexp(img, 5, 1.1)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print("polyexp variant %d %dx%d same=%s %.1f us" % (v, w, h, same, dt * 1e6), flush=True)
        c.close()
