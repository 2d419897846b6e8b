#!/usr/bin/env python3
"""Soak of the named-frame cache (ofxcv_vectorgen_flows_host_keyed): N threads, one context each, render random output frames of one
sequence with named and unnamed frames mixed, a cache small enough to evict all the time; every output frame is compared with the
unnamed call's.  usage: python tools/soak_named_frames.py [--threads 8] [--seconds 10] [--cache-mb 1]"""
import argparse, os, sys, threading, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--seconds", type=float, default=10)
ap.add_argument("--cache-mb", type=int, default=1)
ap.add_argument("--devices", type=int, default=1, help="logical devices (OFXCV_VIRTUAL_DEVICES over the physical ones); thread k renders on device k % N")
ap.add_argument("--opts", default="", help="context options of the render threads: k=v,k=v")
args = ap.parse_args()
if args.devices > 1:
    os.environ["OFXCV_VIRTUAL_DEVICES"] = str(args.devices)
w, h, n = 384, 216, 12
seq = [synth.flow_pair(w, h, seed=300 + i)[0] for i in range(n + 2)]
plain = ofxcv.Context(0)
plain.host_cache_clear()
want = {}
for t in range(1, n + 1):
    o = np.zeros((h, w, 4), np.float32)
    plain.vectorgen_flows_host(seq[t], seq[t + 1], seq[t - 1], o, 1, 2, 4, 8)
    want[t] = o
bad, calls, stats = [], [0] * args.threads, []
stop = threading.Event()
def work(k):
    rng = random.Random(k)
    c = ofxcv.Context(k % args.devices)
    for kv in filter(None, args.opts.split(",")):
        c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    c.set_option("host.cache_mb", args.cache_mb)
    c.set_option("host.split", rng.choice((0, 1, 2)))
    o = np.zeros((h, w, 4), np.float32)
    while not stop.is_set():
        t = rng.randint(1, n)
        keys = tuple(("q%d" % u) if rng.random() < 0.8 else None for u in (t, t + 1, t - 1))
        o[...] = -1
        c.vectorgen_flows_host(seq[t], seq[t + 1], seq[t - 1], o, 1, 2, 4, 8, keys=keys)
        if not np.array_equal(o, want[t]):
            bad.append((k, t, keys))
        calls[k] += 1
    stats.append((c.host_cache_hits(), c.host_cache_misses()))
    c.close()
th = [threading.Thread(target=work, args=(k,)) for k in range(args.threads)]
[t.start() for t in th]
time.sleep(args.seconds)
stop.set()
[t.join() for t in th]
print("%d threads, %d calls, %d wrong frames; hits / misses per thread %s; cache now %s" % (args.threads, sum(calls), len(bad), stats, plain.host_cache_stats()))
sys.exit(1 if bad else 0)
