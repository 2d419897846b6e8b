#!/usr/bin/env python3
"""Times the inpaint render() body (host image in/out) and the oracle on the same frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
from oracle import binding as oracle
ctx = ofxcv.Context(0)
for kv in filter(None, os.environ.get('BENCH_CTX_OPTIONS', '').split(',')):
    ctx.set_option(kv.split('=')[0], int(kv.split('=')[1]))
for (w, h) in [(640, 480), (1920, 1080)]:
    fr = synth.inpaint_frame(w, h)
    ctx.inpaint_render_host(fr)
    t0 = time.perf_counter(); n = 5
    for _ in range(n): ctx.inpaint_render_host(fr)
    g = (time.perf_counter() - t0) / n
    t0 = time.perf_counter(); oracle.inpaint_render(fr); c = time.perf_counter() - t0
    d = torch.from_numpy(fr).cuda(); m = ctx.inpaint_mask(d, 1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): ctx.inpaint_telea(d, m); torch.cuda.synchronize()
    tt = (time.perf_counter() - t0) / n
    hole = int((m > 0).sum())
    print("%dx%d hole px %d: render_host %.2f ms, telea(device images) %.2f ms (%.2f Mpx_hole/s), cpu oracle %.1f ms" % (w, h, hole, g * 1e3, tt * 1e3, hole / tt / 1e6, c * 1e3))
