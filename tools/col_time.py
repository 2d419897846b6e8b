#!/usr/bin/env python3
"""Average duration of the dominant launch -- iterate_col_kernel (iterate, iterate), 8 pairs at 1920x1080 -- by HIP events (ofxcv_profile_enable), a few
contexts one after the other in ONE process, optionally with the scratch placed at different offsets (OFXCV_SCRATCH_SKEW) or with other options.
usage: python tools/col_time.py [--skews 0,4096,...] [--opts k=v,...] [--reps 3] [--size WxH] [--pairs 8]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--skews", default="0")
ap.add_argument("--opts", default="")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--calls", type=int, default=12)
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--pairs", type=int, default=8)
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
c0 = ofxcv.Context(0)
prs = []
for i in range(args.pairs):
    a, b = synth.flow_pair(W, H, seed=1234 + i)
    prs.append((c0.to_byte_grayscale(torch.from_numpy(a).cuda()), c0.to_byte_grayscale(torch.from_numpy(b).cuda())))
ga = [p[0] for p in prs]; gb = [p[1] for p in prs]
flows = [torch.empty((H, W, 2), device="cuda") for _ in range(args.pairs)]
for rep in range(args.reps):
    for skew in [int(v, 0) for v in args.skews.split(",")]:
        os.environ["OFXCV_SCRATCH_SKEW"] = str(skew)
        c = ofxcv.Context(0)
        for kv in filter(None, args.opts.split(",")):
            c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        with torch.cuda.stream(c.stream):
            for _ in range(2):
                c.calc_optical_flow_farneback_batch(ga, gb, flows)
        c.synchronize()
        c.profile_enable(1)
        with torch.cuda.stream(c.stream):
            for _ in range(args.calls):
                c.calc_optical_flow_farneback_batch(ga, gb, flows)
        c.synchronize()
        ms, n = c.profile_read()
        c.profile_enable(0)
        t0 = time.perf_counter()
        with torch.cuda.stream(c.stream):
            for _ in range(args.calls):
                c.calc_optical_flow_farneback_batch(ga, gb, flows)
        c.synchronize()
        el = time.perf_counter() - t0
        print("rep %d skew %8d opts %s: dominant launch %.1f us (%d launches), %.0f pairs/s, aborts %d" % (rep, skew, args.opts or "-", ms / max(1, n) * 1e3, n, args.pairs * args.calls / el, c.get_option("farneback.col_aborts")), flush=True)
        c.close()
