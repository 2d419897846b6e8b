#!/usr/bin/env python3
"""Render-body time of the inpaint call over hole shapes at 1920x1080: the BASELINE frame (12 ellipses), one large blob, many small
holes (dust), thin scratches, a wide border strip -- with the hole pixels and the CPU oracle's time of each."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
from oracle import binding as oracle
W, H = 1920, 1080
base = synth.inpaint_frame(W, H, n_holes=0)
yy, xx = np.mgrid[0:H, 0:W]
rng = np.random.default_rng(7)
frames = {"12 ellipses (BASELINE)": synth.inpaint_frame(W, H)}
f = base.copy(); f[((xx - 960) / 420.0) ** 2 + ((yy - 540) / 300.0) ** 2 <= 1, :3] = 0; frames["one blob 840x600"] = f
f = base.copy()
for _ in range(3000):
    x, y, r = rng.integers(4, W - 4), rng.integers(4, H - 4), rng.integers(1, 4)
    f[max(0, y - r):y + r + 1, max(0, x - r):x + r + 1, :3] = 0
frames["3000 specks of dust"] = f
f = base.copy()
for _ in range(40):
    x0, y0 = rng.integers(0, W), rng.integers(0, H)
    ang = rng.uniform(0, np.pi); L = rng.integers(200, 900)
    t = np.arange(L)
    xs = np.clip((x0 + t * np.cos(ang)).astype(int), 0, W - 1); ys = np.clip((y0 + t * np.sin(ang)).astype(int), 0, H - 1)
    for d in (0, 1):
        f[np.clip(ys + d, 0, H - 1), xs, :3] = 0
frames["40 scratches, 2 px wide"] = f
f = base.copy(); f[:, :160, :3] = 0; frames["a 160-pixel strip along the left edge"] = f
c = ofxcv.Context(0)
for name, fr in frames.items():
    ref = oracle.inpaint_render(fr, 3.0, 1.0)
    t0 = time.perf_counter(); oracle.inpaint_render(fr, 3.0, 1.0); cpu = time.perf_counter() - t0
    out = c.inpaint_render_host(fr, 3.0, 1.0)
    ok = np.array_equal(np.asarray(out), ref)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); c.inpaint_render_host(fr, 3.0, 1.0); ts.append(time.perf_counter() - t0)
    holes = int((oracle.inpaint_mask(fr, 1) > 0).sum())
    print("%-40s %7d hole pixels: %7.2f ms (CPU oracle %7.1f ms), fallbacks %d, %s" % (name, holes, statistics.median(ts) * 1e3, cpu * 1e3, c.inpaint_fallback_count(), "identical" if ok else "DIFFERENT"), flush=True)
c.close()
