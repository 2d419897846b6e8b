#!/usr/bin/env python3
"""Random masks (blobs, lines, border-touching, nearly-full) and radii: GPU inpaint (both methods) vs the oracle, bit-exact."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from oracle import binding as oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ctx = ofxcv.Context(0)
bad = 0
for case in range(40):
    w, h = int(rng.integers(4, 220)), int(rng.integers(4, 160))
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    mask = np.zeros((h, w), np.uint8)
    kind = case % 5
    if kind == 0:
        mask[rng.random((h, w)) < 0.05] = 255                       # salt
    elif kind == 1:
        for _ in range(4):                                           # rectangles, may touch the border
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            mask[y0:y0 + int(rng.integers(1, 30)), x0:x0 + int(rng.integers(1, 40))] = 255
    elif kind == 2:
        mask[:, ::7] = 255                                           # vertical lines
    elif kind == 3:
        mask[:] = 255
        mask[h // 2, w // 2] = 0                                     # almost everything is a hole
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        mask[(xx - w / 2) ** 2 + (yy - h / 2) ** 2 < (min(w, h) / 3) ** 2] = 255
    radius = float(rng.choice([1, 2, 3, 3, 4, 5, 6, 7, 9, 12, 14]))  # 6 .. 12: the large-window fill; 14: the level schedule
    method = int(rng.integers(0, 2))
    ref, t_ref, f_ref, o_ref = oracle.inpaint(rgb, mask, radius, method, maps=True)
    got, t, order = ctx.inpaint(torch.from_numpy(rgb).cuda(), torch.from_numpy(mask).cuda(), radius, method, maps=True)
    ok = np.array_equal(got.cpu().numpy(), ref) and np.array_equal(order.cpu().numpy(), o_ref) and np.array_equal(t.cpu().numpy(), t_ref)
    bad += not ok
    print("%3dx%-3d kind %d radius %g method %d hole %5d -> %s" % (w, h, kind, radius, method, int((mask > 0).sum()), "ok" if ok else "MISMATCH"), flush=True)
print("mismatching cases:", bad)
