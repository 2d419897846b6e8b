#!/usr/bin/env python3
"""N batched Farneback calls with the given options, one in flight (for rocprofv3 --kernel-trace + tools/trace_by_grid.py).
usage: trace_call.py [--size WxH] [--batch N] [--calls N] "opt=val,opt=val" """
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--calls", type=int, default=20)
ap.add_argument("opts", nargs="?", default="")
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
prs = [synth.flow_pair(W, H, seed=100 + i) for i in range(args.batch)]
c = ofxcv.Context(0)
for kv in filter(None, args.opts.split(",")):
    k, v = kv.split("=")
    c.set_option(k, int(v))
with torch.cuda.stream(c.stream):
    ga = [c.to_byte_grayscale(torch.from_numpy(a).cuda()) for a, _ in prs]
    gb = [c.to_byte_grayscale(torch.from_numpy(b).cuda()) for _, b in prs]
    fl = c.calc_optical_flow_farneback_batch(ga, gb)
    for _ in range(3):
        c.calc_optical_flow_farneback_batch(ga, gb, fl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.calls):
        c.calc_optical_flow_farneback_batch(ga, gb, fl)
    torch.cuda.synchronize()
    call = (time.perf_counter() - t0) / args.calls
print("%s %dx%d batch %d: call %.1f us = %.1f pairs/s" % (args.opts or "(defaults)", W, H, args.batch, call * 1e6, args.batch / call))
c.close()
