#!/usr/bin/env python3
"""K timed-workload steps and nothing else (for counter passes): per step 2 x B gray LUTs + ONE batched Farneback call of B pairs with F7 inside.
usage: python tools/run_batch_calls.py [--size WxH] [--batch B] [--calls K] ["opt=val,..."]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--calls", type=int, default=2)
ap.add_argument("opts", nargs="?", default="")
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
c = ofxcv.Context(0)
for kv in filter(None, args.opts.split(",")):
    k, v = kv.split("=")
    c.set_option(k, int(v))
prs = [synth.flow_pair(W, H, seed=1234 + i) for i in range(args.batch)]
with torch.cuda.stream(c.stream):
    a = [torch.from_numpy(x).cuda() for x, _ in prs]
    b = [torch.from_numpy(y).cuda() for _, y in prs]
    ga = [torch.empty((H, W), dtype=torch.uint8, device="cuda") for _ in prs]
    gb = [torch.empty((H, W), dtype=torch.uint8, device="cuda") for _ in prs]
    fl = [torch.empty((H, W, 2), dtype=torch.float32, device="cuda") for _ in prs]
    out = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in prs]
    for _ in range(args.calls):
        for x, y, gx, gy in zip(a, b, ga, gb):
            c.to_byte_grayscale(x, gx)
            c.to_byte_grayscale(y, gy)
        c.calc_optical_flow_farneback_batch_rgba(ga, gb, fl, out, [1] * args.batch, [2] * args.batch)
torch.cuda.synchronize()
print("pairs processed: %d" % (args.calls * args.batch))
c.close()
