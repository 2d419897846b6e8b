#!/usr/bin/env python3
"""Phase stamps of the Telea fill (library hook OFXCV_FILL_TRACE=<file>: eight 64-bit words per fill-order pixel -- shader-clock stamps at
0 start, 1 maps staged + weights, 2 awaited colours there, 3 colours staged, 4 terms in LDS, 5 sums done, 6 pixel stored; 7 = the
largest order number the pixel waited for).  Runs one inpaint call of a shape and prints where a pixel's time goes and what a
hand-off costs.  usage: python tools/fill_trace.py [ellipses|strip|scratches|dust] [--size WxH]"""
import argparse, os, sys, time
ap = argparse.ArgumentParser()
ap.add_argument("shape", nargs="?", default="ellipses")
ap.add_argument("--size", default="1920x1080")
args = ap.parse_args()
TRACE = "/tmp/ofxcv_fill_trace.bin"
os.environ["OFXCV_FILL_TRACE"] = TRACE
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W, H = (int(v) for v in args.size.split("x"))
rng = np.random.default_rng(7)
if args.shape == "ellipses":
    fr = synth.inpaint_frame(W, H)
else:
    fr = synth.inpaint_frame(W, H, n_holes=0)
    if args.shape == "strip":
        fr[:, :160, :3] = 0
    elif args.shape == "dust":
        for _ in range(3000):
            x, y, r = rng.integers(4, W - 4), rng.integers(4, H - 4), rng.integers(1, 4)
            fr[max(0, y - r):y + r + 1, max(0, x - r):x + r + 1, :3] = 0
    else:
        for _ in range(40):
            x0, y0, ang, L = rng.integers(0, W), rng.integers(0, H), rng.uniform(0, np.pi), rng.integers(200, 900)
            t = np.arange(L)
            xs = np.clip((x0 + t * np.cos(ang)).astype(int), 0, W - 1); ys = np.clip((y0 + t * np.sin(ang)).astype(int), 0, H - 1)
            for d in (0, 1):
                fr[np.clip(ys + d, 0, H - 1), xs, :3] = 0
c = ofxcv.Context(0)
c.inpaint_render_host(fr)
t0 = time.perf_counter(); c.inpaint_render_host(fr); wall = time.perf_counter() - t0
tr = np.fromfile(TRACE, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
tr = tr[tr[:, 6] > 0]
n = len(tr)
span = tr[:, 6].max() - tr[:, 0].min()
print("%s %dx%d: %d pixels traced, call %.2f ms, first start -> last store %d ticks" % (args.shape, W, H, n, wall * 1e3, span))
names = ["maps staged + weights (1-0)", "waiting for colours (2-1)", "colours staged (3-2)", "terms (4-3)", "ordered sums (5-4)", "finish + store (6-5)"]
for k, nm in enumerate(names):
    d = tr[:, k + 1] - tr[:, k]
    print("  %-30s median %6d  mean %8.1f  p90 %6d ticks" % (nm, np.median(d), d.mean(), np.percentile(d, 90)))
post = tr[:, 6] - tr[:, 2]
print("  %-30s median %6d  mean %8.1f ticks" % ("after the colours (6-2)", np.median(post), post.mean()))
w = tr[:, 7]
has = (w > 0) & (w <= n)
store_of = np.zeros(n + 1, np.int64)
store_of[1:] = tr[:, 6] if len(tr) == n else 0
# pixels are rows in order-number order only if none were dropped: use the order implied by row index of the raw file instead
raw = np.fromfile(TRACE, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
store = raw[:, 6]
idx = np.nonzero((raw[:, 6] > 0) & (raw[:, 7] > 0))[0]
pred = raw[idx, 7] - 1
ok = store[pred] > 0
hand = raw[idx[ok], 2] - store[pred[ok]]
waited = (raw[idx[ok], 2] - raw[idx[ok], 1]) > 200
print("  hand-off: colours there - predecessor's store stamp, pixels that really waited (%d of %d): median %d  mean %.1f  p90 %d ticks" %
      (waited.sum(), len(hand), np.median(hand[waited]) if waited.any() else 0, hand[waited].mean() if waited.any() else 0, np.percentile(hand[waited], 90) if waited.any() else 0))
c.close()
