#!/usr/bin/env python3
"""Random small / odd frame sizes and parameters, whole Farneback call: direct-window mode vs the oracle's DIRECT evaluation
(bit-identical) and the default OpenCV-order mode vs the FAITHFUL evaluation (every sample within 1e-4; winsize 3, box window)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
from oracle import binding as oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ctx = ofxcv.Context(0)
ctx.set_option("farneback.opencv_rounding", 0)
sctx = ofxcv.Context(0)
bad = 0
cases = [(5, 7), (1, 1), (2, 9), (16, 16), (17, 3), (63, 65), (64, 64), (65, 15), (127, 129)] + \
        [(int(rng.integers(3, 400)), int(rng.integers(3, 300))) for _ in range(30)]
for (w, h) in cases:
    a, b = synth.flow_pair(max(w, 2), max(h, 2), seed=int(rng.integers(1 << 30)))
    a, b = a[:h, :w], b[:h, :w]
    ga, gb = oracle.to_byte_grayscale(np.ascontiguousarray(a)), oracle.to_byte_grayscale(np.ascontiguousarray(b))
    kw = dict(levels=int(rng.integers(0, 5)), iterations=int(rng.integers(1, 6)), winsize=int(rng.choice([3, 3, 3, 5])),
              poly_n=int(rng.choice([5, 5, 7, 3])), poly_sigma=float(rng.choice([1.1, 1.5])))
    flags = int(rng.choice([0, 0, 256]))
    ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_DIRECT, flags=flags, **kw)
    got = ctx.calc_optical_flow_farneback(torch.from_numpy(ga).cuda(), torch.from_numpy(gb).cuda(), flags=flags, **kw).cpu().numpy()
    ok = np.array_equal(ref, got)
    if kw["winsize"] == 3 and flags == 0:
        fref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL, flags=flags, **kw)
        sgot = sctx.calc_optical_flow_farneback(torch.from_numpy(ga).cuda(), torch.from_numpy(gb).cuda(), flags=flags, **kw).cpu().numpy()
        ok = ok and bool((np.abs(sgot - fref) <= 1e-4 * np.maximum(1, np.abs(fref))).all())
    bad += not ok
    print("%4dx%-4d %s flags=%d -> %s" % (w, h, kw, flags, "ok" if ok else "MISMATCH (%d px)" % (ref != got).any(axis=2).sum()), flush=True)
print("mismatching cases:", bad)
