#!/usr/bin/env python3
"""Phase times of the inpaint render body (OFXCV_TRACE_INPAINT=1 prints them on stderr), serial and component-parallel front march."""
import os, sys, time
os.environ["OFXCV_TRACE_INPAINT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
for (w, h) in ((640, 480), (1920, 1080)):
    fr = synth.inpaint_frame(w, h)
    for par in (1, 0):
        c = ofxcv.Context(0)
        c.set_option("inpaint.parallel_march", par)
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); c.inpaint_render_host(fr); ts.append(time.perf_counter() - t0)
        print("%dx%d parallel_march %d: render_host median %.2f ms" % (w, h, par, np.median(ts[1:]) * 1e3), file=sys.stderr, flush=True)
        c.close()
