#!/usr/bin/env python3
"""A/B of the level-0 iteration kernel: per option set, the average launch time of the iteration kernel and of the carry
kernel (HIP event pairs of the library's measurement hook) for a batched call, plus the whole-call time.
usage: python tools/ab_iter.py [--size WxH] [--batch N] "opt=val,opt=val" "opt=val" ...   ("" = defaults)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--calls", type=int, default=6)
ap.add_argument("opts", nargs="*", default=[""])
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
prs = [synth.flow_pair(W, H, seed=100 + i) for i in range(args.batch)]
for o in args.opts:
    c = ofxcv.Context(0)
    for kv in filter(None, o.split(",")):
        k, v = kv.split("=")
        c.set_option(k, int(v))
    with torch.cuda.stream(c.stream):
        ga = [c.to_byte_grayscale(torch.from_numpy(a).cuda()) for a, _ in prs]
        gb = [c.to_byte_grayscale(torch.from_numpy(b).cuda()) for _, b in prs]
        fl = c.calc_optical_flow_farneback_batch(ga, gb)
        res = []
        for which in (1, 2):
            c.profile_enable(which)
            for _ in range(args.calls):
                c.calc_optical_flow_farneback_batch(ga, gb, fl)
            torch.cuda.synchronize()
            ms, n = c.profile_read()
            c.profile_enable(0)
            res.append(ms * 1e3 / max(1, n))
        for _ in range(3):
            c.calc_optical_flow_farneback_batch(ga, gb, fl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.calls):
            c.calc_optical_flow_farneback_batch(ga, gb, fl)
        torch.cuda.synchronize()
        call = (time.perf_counter() - t0) / args.calls
    print("%-50s %dx%d batch %d: iteration %7.1f us  carries %6.1f us  call %8.1f us = %7.1f pairs/s" %
          (o or "(defaults)", W, H, args.batch, res[0], res[1], call * 1e6, args.batch / call), flush=True)
    c.close()
