#!/usr/bin/env python3
"""Times individual Farneback stages (stage-level C ABI) with HIP events on the context stream.
usage: python tools/bench_stage.py [W H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
ctx = ofxcv.Context(0)
for kv in filter(None, os.environ.get("BENCH_CTX_OPTIONS", "").split(",")):
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))

def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(ctx.stream)
    for _ in range(n): fn()
    e1.record(ctx.stream); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

with torch.cuda.stream(ctx.stream):
    a, b = synth.flow_pair(W, H)
    ga = ctx.to_byte_grayscale(torch.from_numpy(a).cuda()); gb = ctx.to_byte_grayscale(torch.from_numpy(b).cuda())
    lw, lh, sg, ks = ofxcv.farneback_level_geom(W, H, 0.5, 0)
    I0 = ctx.farneback_pyr_image(ga, lw, lh, sg, ks); I1 = ctx.farneback_pyr_image(gb, lw, lh, sg, ks)
    R0 = ctx.farneback_polyexp(I0); R1 = ctx.farneback_polyexp(I1)
    flow = torch.zeros((H, W, 2), device="cuda"); flow[..., 0] = 2.5; flow[..., 1] = -1.25
    M = ctx.farneback_update_matrices(R0, R1, flow)
    px = W * H
    Mo = torch.zeros_like(M); fl = torch.zeros_like(flow)
    if os.environ.get("ONLY_ITER"):
        for _ in range(int(os.environ["ONLY_ITER"])): ctx.farneback_update_flow_blur(R0, R1, M, W, 3, True, fl, Mo)
        torch.cuda.synchronize(); sys.exit(0)
    t = timeit(lambda: ctx.farneback_update_flow_blur(R0, R1, M, W, 3, True, fl, Mo))
    print("iterate(update)   %8.1f us  %6.0f GB/s (80 B/px)" % (t, 80 * px / t / 1e3))
    t = timeit(lambda: ctx.farneback_update_flow_blur(R0, R1, M, W, 3, False, fl))
    print("iterate(final)    %8.1f us  %6.0f GB/s (28 B/px)" % (t, 28 * px / t / 1e3))
    t = timeit(lambda: ctx.farneback_update_matrices(R0, R1, flow))
    print("update_matrices   %8.1f us  %6.0f GB/s (68 B/px)" % (t, 68 * px / t / 1e3))
    t = timeit(lambda: ctx.farneback_polyexp(I0))
    print("polyexp           %8.1f us  %6.0f GB/s (24 B/px)" % (t, 24 * px / t / 1e3))
    t = timeit(lambda: ctx.farneback_pyr_image(ga, lw, lh, sg, ks))
    print("pyr_image k=0     %8.1f us" % t)
    for k in (1, 2, 3):
        g = ofxcv.farneback_level_geom(W, H, 0.5, k)
        t = timeit(lambda: ctx.farneback_pyr_image(ga, *g))
        print("pyr_image k=%d     %8.1f us" % (k, t))
