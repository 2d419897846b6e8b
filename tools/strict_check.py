#!/usr/bin/env python3
"""Strip-parallel OpenCV-order window (farneback.opencv_rounding=1) against the FAITHFUL oracle and against the serial
column scan (=2), at several sizes; prints max error, fraction outside 1e-4 and fraction of bit-identical samples."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
from oracle import binding as oracle

sizes = [(64, 48), (160, 120), (333, 257), (640, 480), (1920, 1080)]
if len(sys.argv) > 1:
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for w, h in sizes:
    a, b = synth.flow_pair(w, h)
    ga, gb = oracle.to_byte_grayscale(a), oracle.to_byte_grayscale(b)
    ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL)
    out = {}
    for mode, rows in [(0, 0), (2, 0), (1, 0), (1, 2), (1, 4), (1, 8)]:
        ctx = ofxcv.Context(0)
        ctx.set_option("farneback.opencv_rounding", mode)
        ctx.set_option("farneback.strict_rows", rows)
        for kv in filter(None, os.environ.get("BENCH_CTX_OPTIONS", "").split(",")):
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        got = ctx.calc_optical_flow_farneback(torch.from_numpy(ga).cuda(), torch.from_numpy(gb).cuda()).cpu().numpy()
        err = np.abs(ref - got)
        bad = err > 1e-4 * np.maximum(1, np.abs(ref))
        out[(mode, rows)] = got
        print("%dx%d mode %d rows %d: max err %.3g outside-1e-4 %.3g bit-identical-to-faithful %.6f  identical-to-serial %s" % (
            w, h, mode, rows, err.max(), bad.mean(), (got == ref).mean(),
            "%.6f" % (got == out[(2, 0)]).mean() if (2, 0) in out else "-"), flush=True)
        ctx.close()
