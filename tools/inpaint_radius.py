#!/usr/bin/env python3
"""Render-body time of the inpaint call over the radius parameter (the LDS-window dataflow fill serves radii up to 5, the
barrier-scheduled fill the rest).  usage: python tools/inpaint_radius.py [--size WxH] [radii ...]"""
import argparse, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("radii", nargs="*", type=float, default=[1, 3, 5, 6, 8, 12])
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
fr = synth.inpaint_frame(W, H)
c = ofxcv.Context(0)
for r in args.radii:
    for _ in range(2):
        c.inpaint_render_host(fr, r, 1.0)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        c.inpaint_render_host(fr, r, 1.0)
        ts.append(time.perf_counter() - t0)
    print("%dx%d radius %4.1f: render body %.2f ms (min %.2f), fallbacks %d" % (W, H, r, statistics.median(ts) * 1e3, min(ts) * 1e3, c.inpaint_fallback_count()), flush=True)
c.close()
