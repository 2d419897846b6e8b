#!/usr/bin/env python3
"""Random frame sizes (up to 2100 x 1300), batch sizes, parameters, strip geometries and initial flows: the three evaluations of OpenCV's
running column sums the library has -- the serial column scan (farneback.opencv_rounding 2: the reference form), the overlapped strips
(random geometry hooks) and the column-owning workgroups (forced on every level that is tall enough, both row geometries) -- must give
the same flows bit for bit -- and so must the library's defaults (its own plan of which pairs of a call take which form) -- and no bounded wait may run out.  (All are within 1e-4 of the faithful oracle at every sample:
tests/test_farneback_gpu.py, tests/perf/fuzz_sizes.py.)
usage: python tests/perf/fuzz_halo.py [seed] [cases]"""
import faulthandler, os, sys
faulthandler.enable()
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
    wide = rng.random() < 0.25  # wide, low frames in batches of 6 .. 10 with the library's own plan: some pairs column-owning, the rest in strips
    if wide:
        w, h, n, big = int(rng.integers(1700, 2100)), int(rng.integers(64, 200)), int(rng.integers(6, 11)), True
    kw = dict(levels=int(rng.integers(0, 5)), iterations=int(rng.integers(1, 6)), poly_n=int(rng.choice([5, 5, 7])), pyr_scale=float(rng.choice([0.5, 0.5, 0.7])))
    init = rng.random() < 0.25
    if init:
        kw["flags"] = ofxcv.OPTFLOW_USE_INITIAL_FLOW
    halo = {"farneback.col": 0}
    # farneback.halo_geom (test hook): low nibble the form, bits 4..6 the small form's wavefronts, bits 8.. the rows of a tall strip
    hg = 0
    if rng.random() < 0.5:
        hg |= int(rng.integers(0, 4))
    if rng.random() < 0.3:
        hg |= int(rng.choice([2, 3, 4, 5, 6])) << 4
    if rng.random() < 0.3:
        hg |= int(rng.choice([33, 35, 36, 65, 68, 71, 72])) << 8
    if hg:
        halo["farneback.halo_geom"] = hg
    if rng.random() < 0.2:
        halo["farneback.batch_mb"] = int(rng.choice([1, 8, 40]))
    col = {"farneback.col_min": 1, "farneback.col_ring": int(rng.integers(0, 2))}
    base = rng.integers(0, 256, size=(h + 24, w + 24), dtype=np.uint8)
    blur = (base[:-2, :-2].astype(np.int32) + base[1:-1, 1:-1] + base[2:, 2:]) // 3
    pa = [torch.from_numpy(np.ascontiguousarray(blur[i:i + h, i:i + w]).astype(np.uint8)).cuda() for i in range(n)]
    pb = [torch.from_numpy(np.ascontiguousarray(blur[i + 1:i + 1 + h, i + 2:i + 2 + w]).astype(np.uint8)).cuda() for i in range(n)]
    inits = [rng.normal(0, 2, size=(h, w, 2)).astype(np.float32) for _ in range(n)]
    print("case %d: %dx%d n=%d %s %s %s" % (case, w, h, n, kw, halo, col), flush=True)
    outs, aborts = [], 0
    for opts in ({"farneback.opencv_rounding": 2}, halo, col, {}):
        c = ofxcv.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        for _ in range(2):
            fl = [torch.from_numpy(i0.copy()).cuda() for i0 in inits] if init else None
            fl = c.calc_optical_flow_farneback_batch(pa, pb, fl, **kw)
        outs.append([f.cpu().numpy() for f in fl])
        aborts += c.get_option("farneback.col_aborts")
        c.close()
    ok = all(all(np.array_equal(x, y) for x, y in zip(outs[0], o)) for o in outs[1:]) and aborts == 0
    bad += not ok
    print("%4dx%-4d n=%d %s %s %s -> %s" % (w, h, n, kw, halo, col, "ok" if ok else "MISMATCH"), flush=True)
print("mismatching cases:", bad)
