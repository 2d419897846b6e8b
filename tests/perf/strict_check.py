#!/usr/bin/env python3
"""OpenCV-order window variants against the FAITHFUL oracle: folded carries (default), the two-kernel form, the serial scan."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    for name, opts in [("serial", {"farneback.opencv_rounding": 2}), ("two-kernel", {"farneback.fold_carries": 0}), ("folded-1", {"farneback.fold_carries": 1}), ("folded-2", {"farneback.fold_carries": 2}), ("row pairs", {"farneback.strict_variant": 2})]:
        ctx = ofxcv.Context(0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        for rep in range(2):
            got = ctx.calc_optical_flow_farneback(torch.from_numpy(ga).cuda(), torch.from_numpy(gb).cuda()).cpu().numpy()
        err = np.abs(ref - got)
        bad = err > 1e-4 * np.maximum(1, np.abs(ref))
        print("%dx%d %-10s: max err %.3g outside-1e-4 %.3g bit-identical-to-faithful %.6f finite %s" % (
            w, h, name, np.nanmax(err), bad.mean(), (got == ref).mean(), np.isfinite(got).all()), flush=True)
        ctx.close()
