#!/usr/bin/env python3
"""Where a render thread's time goes when several threads render named frames at once (the device's submission queue, vectorgen.hip):
per host-image call the phases the library stamps with OFXCV_HOST_TRACE=1 -- upload + conversion enqueued, frames complete, waiting in the
queue, the batched call, wake-up, download -- the sizes of the batched calls and the fraction of the wall time one was running.
usage: python tools/host_queue_trace.py [--threads 4] [--seconds 2] [--timeline 30]      (BENCH_CTX_OPTIONS=opt=val,...)"""
import argparse, os, sys, time, threading, ctypes as C
os.environ["OFXCV_HOST_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=4)
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--timeline", type=float, default=0, help="ms of the middle of the run to print call by call")
ap.add_argument("--unnamed", action="store_true")
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
nt = args.threads
shot = synth.sequence(W, H, 4)          # consecutive frames of one shot
seq = [shot[synth.pingpong(k, 4)].copy() for k in range(30)]   # 30 buffers = 5 periods of the forth-and-back run: the wrap-around is seamless
ctxs = [ofxcv.Context(0) for _ in range(nt)]
for kv in filter(None, os.environ.get("BENCH_CTX_OPTIONS", "").split(",")):
    for c in ctxs:
        c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
outs = [np.zeros((H, W, 4), np.float32) for _ in range(nt)]
for c, o in zip(ctxs, outs):
    c.vectorgen_flows_host(seq[1], seq[2], seq[0], o, 1, 2, 4, 8)
ctxs[0].host_cache_clear()
lib = ofxcv.lib()
lib.ofxcv_debug_host_trace(None, C.c_int(0))
stop = threading.Event()
counts = [0] * nt
next_frame = [0]
frame_lock = threading.Lock()
def work(i):
    n = len(seq)
    while not stop.is_set():
        with frame_lock:            # a host hands out the output frames of a sequence in order, whichever render thread is free
            t = next_frame[0]
            next_frame[0] += 1
        keys = None if args.unnamed else ("f%d" % t, "f%d" % (t + 1), "f%d" % (t - 1))
        ctxs[i].vectorgen_flows_host(seq[t % n], seq[(t + 1) % n], seq[(t - 1) % n], outs[i], 1, 2, 4, 8, keys=keys)
        counts[i] += 1
th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
t0 = time.perf_counter()
[t.start() for t in th]
time.sleep(args.seconds)
stop.set()
[t.join() for t in th]
el = time.perf_counter() - t0
buf = (C.c_double * (16 * 100000))()
m = lib.ofxcv_debug_host_trace(buf, C.c_int(len(buf)))
tr = np.array(buf[:m]).reshape(-1, 16)
print("%d threads: %.0f pairs/s, %d calls traced" % (nt, 2 * sum(counts) / el, len(tr)))
tr = tr[len(tr) // 5:]   # steady state
co = tr[tr[:, 7] > 0]
def ms(x): return "%.2f" % (np.mean(x) / 1e3)
print("per call (ms, mean): entry->frames enqueued %s" % ms(tr[:, 1] - tr[:, 0]))
if len(co):
    print("  coalesced calls %d: ready event recorded +%s, queued +%s (until its call was being enqueued), rider released +%s after that, then waiting for the call and downloading +%s; whole call %s" % (
        len(co), ms(co[:, 2] - co[:, 1]), ms(co[:, 3] - co[:, 2]), ms(co[:, 5] - co[:, 3]), ms(co[:, 6] - co[:, 5]), ms(co[:, 6] - co[:, 0])))
    led = co[(co[:, 8] > 0) & (co[:, 4] > 0)]
    print("  leaders (%d): enqueueing the call %s ms (event waits, gather, launches, F7), enqueue -> complete %s ms" % (len(led), ms(led[:, 15] - led[:, 3]), ms(led[:, 4] - led[:, 3])))
    sizes = led[:, 7].astype(int)
    span = (co[:, 6].max() - co[:, 0].min())
    print("  batched calls %d: pairs histogram %s; one was in flight %.0f %% of the time; mean enqueue -> complete by pairs: %s" % (
        len(led), {int(k): int((sizes == k).sum()) for k in sorted(set(sizes))}, 100 * (led[:, 4] - led[:, 3]).sum() / span,
        {int(k): round(float(np.mean((led[:, 4] - led[:, 3])[sizes == k])) / 1e3, 2) for k in sorted(set(sizes))}))
own = tr[tr[:, 7] == 0]
if len(own):
    print("  calls that ran their own Farneback call: %d, whole call %s ms" % (len(own), ms(own[:, 6] - own[:, 0])))
if args.timeline:
    mid = np.median(tr[:, 0])
    w = tr[(tr[:, 0] >= mid) & (tr[:, 0] < mid + args.timeline * 1e3)]
    for r in sorted(w.tolist()):
        print("  %8.2f entry | lookup +%.2f cached +%.2f uploads(%d) +%.2f pending(%d) +%.2f | +%.2f enq |" % ((r[0] - mid) / 1e3, (r[9] - r[0]) / 1e3, (r[10] - r[9]) / 1e3, r[13], (r[11] - r[10]) / 1e3, r[14], (r[1] - r[11]) / 1e3, (r[1] - r[0]) / 1e3), end="")
        print(" +%.2f ready | batch %8.2f .. %8.2f (%d pairs%s) | woke +%.2f | done %8.2f" % ((r[2] - r[1]) / 1e3, (r[3] - mid) / 1e3, (r[4] - mid) / 1e3, r[7], ", led" if r[8] else "", (r[5] - r[4]) / 1e3, (r[6] - mid) / 1e3))
        continue
        print("  %8.2f entry | +%.2f enq | +%.2f ready | batch %8.2f .. %8.2f (%d pairs%s) | woke +%.2f | done %8.2f" % (
            (r[0] - mid) / 1e3, (r[1] - r[0]) / 1e3, (r[2] - r[1]) / 1e3, (r[3] - mid) / 1e3, (r[4] - mid) / 1e3, r[7], ", led" if r[8] else "", (r[5] - r[4]) / 1e3, (r[6] - mid) / 1e3))
for c in ctxs:
    c.close()
