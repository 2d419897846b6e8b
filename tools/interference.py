#!/usr/bin/env python3
"""What the side work of other render threads costs a batched Farneback call of 8 pairs running beside it: one thread loops the batched
call (synchronising after each, like the submission queue's leader), a second thread issues ONE kind of side work at a paced rate --
the gray LUT of a 1080p f32 frame, 2 MB device-to-device copies (named frames taken from / put into the cache), 33 MB pageable uploads,
33 MB downloads -- and the batched call's rate is reported for each.  usage: python tools/interference.py [--rate 600] [--pairs 8]"""
import argparse, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--rate", type=float, default=600.0, help="side operations per second")
ap.add_argument("--pairs", type=int, default=8)
ap.add_argument("--seconds", type=float, default=1.0)
args = ap.parse_args()
W, H = 1920, 1080
a, b = synth.flow_pair(W, H)
c = ofxcv.Context(0); c2 = ofxcv.Context(0)
da = torch.from_numpy(a).cuda()
ga = c.to_byte_grayscale(da); gb = c.to_byte_grayscale(torch.from_numpy(b).cuda())
flows = [torch.empty((H, W, 2), device="cuda") for _ in range(args.pairs)]
g2 = torch.empty_like(ga); gout = torch.empty_like(ga)
host = torch.from_numpy(a.copy()); dev = torch.empty_like(da); hout = torch.empty_like(host)
s2 = torch.cuda.Stream()
def lut():
    with torch.cuda.stream(c2.stream):
        c2.to_byte_grayscale(da, out=gout)
    c2.synchronize()
def d2d():
    with torch.cuda.stream(s2):
        g2.copy_(ga, non_blocking=True)
    s2.synchronize()
def h2d():
    with torch.cuda.stream(s2):
        dev.copy_(host, non_blocking=True)
    s2.synchronize()
def d2h():
    with torch.cuda.stream(s2):
        hout.copy_(dev, non_blocking=True)
    s2.synchronize()
def allk():
    h2d(); lut(); d2d(); d2d(); d2d(); d2h()
side = {"none": None, "gray LUT": lut, "2 MB d2d copy": d2d, "33 MB pageable upload": h2d, "33 MB download": d2h, "all of a call (upload, LUT, 3 d2d, download)": allk}
for name, fn in side.items():
    stop = threading.Event(); n_side = [0]
    def interferer():
        t_next = time.perf_counter()
        while not stop.is_set():
            fn(); n_side[0] += 1
            t_next += 1.0 / args.rate
            d = t_next - time.perf_counter()
            if d > 0: time.sleep(d)
            else: t_next = time.perf_counter()
    th = threading.Thread(target=interferer) if fn else None
    if th: th.start()
    for _ in range(3):
        with torch.cuda.stream(c.stream):
            c.calc_optical_flow_farneback_batch([ga] * args.pairs, [gb] * args.pairs, flows)
        c.synchronize()
    n = 0; t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.seconds:
        with torch.cuda.stream(c.stream):
            c.calc_optical_flow_farneback_batch([ga] * args.pairs, [gb] * args.pairs, flows)
        c.synchronize(); n += 1
    el = time.perf_counter() - t0
    stop.set()
    if th: th.join()
    print("side work: %-48s %6.0f/s | batched call of %d pairs: %.2f ms = %.0f pairs/s" % (name, n_side[0] / el, args.pairs, el / n * 1e3, args.pairs * n / el), flush=True)
