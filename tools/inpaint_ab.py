#!/usr/bin/env python3
"""A/B of the pipelined Telea fill: render-body time of ofxcv_inpaint_render_host per option set.
usage: python tools/inpaint_ab.py [--size WxH] "opt=val,opt=val" ..."""
import argparse, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("opts", nargs="*", default=[""])
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
fr = synth.inpaint_frame(W, H)
ref = None
for o in args.opts:
    c = ofxcv.Context(0)
    for kv in filter(None, o.split(",")):
        k, v = kv.split("=")
        c.set_option(k, int(v))
    for _ in range(3):
        out = c.inpaint_render_host(fr)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        out = c.inpaint_render_host(fr)
        ts.append(time.perf_counter() - t0)
    out = np.asarray(out[0] if isinstance(out, tuple) else out)
    same = "" if ref is None else (" identical" if np.array_equal(ref, out) else " DIFFERENT")
    if ref is None:
        ref = out.copy()
    print("%-50s %dx%d render body %.2f ms (min %.2f) fallbacks %d%s" % (o or "(defaults)", W, H, statistics.median(ts) * 1e3, min(ts) * 1e3, c.inpaint_fallback_count() if hasattr(c, "inpaint_fallback_count") else -1, same), flush=True)
    c.close()
