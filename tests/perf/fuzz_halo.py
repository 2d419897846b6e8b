#!/usr/bin/env python3
"""Random frame sizes (up to 2100 x 1300), batch sizes, parameters and strip geometries: the overlapped-strip form of the
OpenCV-order window (library default, farneback.fold_carries 4) against the carry pre-pass form (0) -- the flows must be equal
bit for bit (both are within 1e-4 of the faithful oracle at every sample: tests/test_farneback_gpu.py, tests/perf/fuzz_sizes.py).
usage: python tests/perf/fuzz_halo.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for case in range(ncases):
    big = rng.random() < 0.3
    w, h = (int(rng.integers(600, 2100)), int(rng.integers(400, 1300))) if big else (int(rng.integers(3, 500)), int(rng.integers(3, 400)))
    n = int(rng.integers(1, 3 if big else 6))
    kw = dict(levels=int(rng.integers(0, 5)), iterations=int(rng.integers(1, 5)), poly_n=int(rng.choice([5, 5, 7])), pyr_scale=float(rng.choice([0.5, 0.5, 0.7])))
    opts = {}
    if rng.random() < 0.5:
        opts["farneback.halo_geom"] = int(rng.integers(0, 4))
    if rng.random() < 0.3:
        opts["farneback.halo_small"] = int(rng.choice([2, 3, 4, 5, 6]))
    if rng.random() < 0.3:
        opts["farneback.halo_strip"] = int(rng.choice([33, 35, 36, 65, 68, 71, 72]))
    if rng.random() < 0.2:
        opts["farneback.batch_mb"] = int(rng.choice([1, 8, 40]))
    if rng.random() < 0.2:
        opts["farneback.persist"] = 1
    base = rng.integers(0, 256, size=(h + 8, w + 8), dtype=np.uint8)
    blur = (base[:-2, :-2].astype(np.int32) + base[1:-1, 1:-1] + base[2:, 2:]) // 3
    pa = [torch.from_numpy(np.ascontiguousarray(blur[i:i + h, i:i + w]).astype(np.uint8)).cuda() for i in range(n)]
    pb = [torch.from_numpy(np.ascontiguousarray(blur[i + 1:i + 1 + h, i + 2:i + 2 + w]).astype(np.uint8)).cuda() for i in range(n)]
    outs = []
    for fold in (0, 4):
        c = ofxcv.Context(0)
        c.set_option("farneback.fold_carries", fold)
        if fold == 4:
            for k, v in opts.items():
                c.set_option(k, v)
        for _ in range(2):
            fl = c.calc_optical_flow_farneback_batch(pa, pb, **kw)
        outs.append([f.cpu().numpy() for f in fl])
        ab = c.get_option("farneback.persist_aborts")
        c.close()
    ok = all(np.array_equal(x, y) for x, y in zip(*outs)) and ab == 0
    bad += not ok
    print("%4dx%-4d n=%d %s %s -> %s" % (w, h, n, kw, opts, "ok" if ok else "MISMATCH"), flush=True)
print("mismatching cases:", bad)
