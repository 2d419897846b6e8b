import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from oracle import binding as oracle
rng = np.random.default_rng(0)
ctx = ofxcv.Context(0)
for case in range(40):
    w, h = int(rng.integers(4, 220)), int(rng.integers(4, 160))
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    mask = np.zeros((h, w), np.uint8)
    kind = case % 5
    if kind == 0:
        mask[rng.random((h, w)) < 0.05] = 255
    elif kind == 1:
        for _ in range(4):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            mask[y0:y0 + int(rng.integers(1, 30)), x0:x0 + int(rng.integers(1, 40))] = 255
    elif kind == 2:
        mask[:, ::7] = 255
    elif kind == 3:
        mask[:] = 255
        mask[h // 2, w // 2] = 0
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        mask[(xx - w / 2) ** 2 + (yy - h / 2) ** 2 < (min(w, h) / 3) ** 2] = 255
    radius = float(rng.choice([1, 2, 3, 3, 4, 5, 7]))
    method = int(rng.integers(0, 2))
    ref, t_ref, f_ref, o_ref = oracle.inpaint(rgb, mask, radius, method, maps=True)
    def run(c):
        got, t, order = c.inpaint(torch.from_numpy(rgb).cuda(), torch.from_numpy(mask).cuda(), radius, method, maps=True)
        return (np.array_equal(got.cpu().numpy(), ref), np.array_equal(order.cpu().numpy(), o_ref), np.array_equal(t.cpu().numpy(), t_ref)), t.cpu().numpy()
    r1, t1 = run(ctx)
    if not all(r1):
        fresh = ofxcv.Context(0)
        r2, t2 = run(fresh)
        fresh.close()
        d = np.argwhere(t1 != t_ref)
        print("%3dx%-3d kind %d radius %g method %d: reused ctx (colour, order, t) %s, fresh ctx %s; t diffs %d first %s got %s ref %s" % (
            w, h, kind, radius, method, r1, r2, len(d), d[:3].tolist(), [float(t1[tuple(x)]) for x in d[:3]], [float(t_ref[tuple(x)]) for x in d[:3]]), flush=True)
