#!/usr/bin/env python3
"""Column-owning form (iterate_col_kernel) against the overlapped-strip form: bit-identity of the flows on a few sizes, batches and
iteration counts, and the abort word.  usage: python tools/col_check.py [ring ...]     (ring: 1 = R1 from the LDS ring, 0 = every gather from memory)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth

extra = []
argv = sys.argv[1:]
if "--opts" in argv:  # further options of the tested context: --opts k=v,k=v
    i = argv.index("--opts")
    extra = [kv.split("=") for kv in argv[i + 1].split(",") if kv]
    del argv[i:i + 2]
geoms = [int(v) for v in argv] or [1, 0]
cases = [(333, 257, 2, dict()), (640, 480, 1, dict()), (125, 70, 3, dict(levels=1)), (640, 480, 2, dict(iterations=4)), (640, 480, 1, dict(iterations=1)),
         (200, 150, 1, dict(iterations=2, levels=0)), (1920, 1080, 2, dict()), (61, 131, 1, dict(levels=0)), (60, 64, 1, dict(levels=0)), (59, 300, 1, dict(levels=1))]
bad = 0
for w, h, n, kw in cases:
    prs = [synth.flow_pair(w, h, seed=50 + i) for i in range(n)]
    ref_ctx = ofxcv.Context(0)
    ref_ctx.set_option("farneback.col", 0)
    ga = [ref_ctx.to_byte_grayscale(torch.from_numpy(a).cuda()) for a, _ in prs]
    gb = [ref_ctx.to_byte_grayscale(torch.from_numpy(b).cuda()) for _, b in prs]
    ref = [f.cpu().numpy() for f in ref_ctx.calc_optical_flow_farneback_batch(ga, gb, **kw)]
    ref_ctx.close()
    for g in geoms:
        c = ofxcv.Context(0)
        c.set_option("farneback.col", 1)
        c.set_option("farneback.col_min", 1)
        c.set_option("farneback.col_ring", g)
        for k_, v_ in extra:
            c.set_option(k_, int(v_))
        got = [f.cpu().numpy() for f in c.calc_optical_flow_farneback_batch(ga, gb, **kw)]
        ab = c.get_option("farneback.col_aborts")
        same = [bool(np.array_equal(x, y)) for x, y in zip(ref, got)]
        mx = max(float(np.nanmax(np.abs(x - y))) for x, y in zip(ref, got))
        nan = any(bool(np.isnan(y).any()) for y in got)
        ok = all(same) and not ab
        bad += not ok
        print("%4dx%-4d batch %d %-28s geom %d: %s  max diff %.3g  nan %s  aborts %d" % (w, h, n, kw, g, "IDENTICAL" if all(same) else "DIFFERENT", mx, nan, ab), flush=True)
        c.close()
print("col_check:", "ok" if not bad else "%d cases differ" % bad)
sys.exit(1 if bad else 0)
