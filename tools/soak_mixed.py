#!/usr/bin/env python3
"""Soak: inpaint calls (fill launches of up to 192 resident workgroups) and host-image Farneback calls (named frames) from several
threads at once on one device; every result is compared with the one the same call gives alone.
usage: python tools/soak_mixed.py [--seconds 10] [--inpaint-threads 2] [--flow-threads 2]"""
import argparse, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=10)
ap.add_argument("--inpaint-threads", type=int, default=2)
ap.add_argument("--flow-threads", type=int, default=2)
args = ap.parse_args()
W, H = 1280, 720
fr = synth.inpaint_frame(W, H)
seq = [synth.flow_pair(W, H, seed=500 + i)[0] for i in range(6)]
c0 = ofxcv.Context(0)
c0.host_cache_clear()
want_ip = np.asarray(c0.inpaint_render_host(fr, 3.0, 1.0)).copy()
want_fl = {}
for t in range(1, 5):
    o = np.zeros((H, W, 4), np.float32)
    c0.vectorgen_flows_host(seq[t], seq[t + 1], seq[t - 1], o, 1, 2, 4, 8)
    want_fl[t] = o
stop = threading.Event()
bad, counts, fallbacks = [], {"inpaint": 0, "flow": 0}, []
def ip_work(k):
    c = ofxcv.Context(0)
    while not stop.is_set():
        out = np.asarray(c.inpaint_render_host(fr, 3.0, 1.0))
        if not np.array_equal(out, want_ip): bad.append(("inpaint", k))
        counts["inpaint"] += 1
    fallbacks.append(c.inpaint_fallback_count())
    c.close()
def fl_work(k):
    c = ofxcv.Context(0)
    o = np.zeros((H, W, 4), np.float32)
    t = 1 + k
    while not stop.is_set():
        tt = 1 + t % 4
        c.vectorgen_flows_host(seq[tt], seq[tt + 1], seq[tt - 1], o, 1, 2, 4, 8, keys=("m%d" % tt, "m%d" % (tt + 1), "m%d" % (tt - 1)))
        if not np.array_equal(o, want_fl[tt]): bad.append(("flow", k, tt))
        counts["flow"] += 1
        t += 1
    c.close()
th = [threading.Thread(target=ip_work, args=(k,)) for k in range(args.inpaint_threads)] + [threading.Thread(target=fl_work, args=(k,)) for k in range(args.flow_threads)]
[t.start() for t in th]
time.sleep(args.seconds)
stop.set()
[t.join() for t in th]
print("%s calls, %d wrong results, inpaint fall-backs %s" % (counts, len(bad), fallbacks))
sys.exit(1 if bad else 0)
