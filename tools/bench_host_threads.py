#!/usr/bin/env python3
"""ofxcv_vectorgen_flows_host (default output frame = 2 flows, one batched call) with 1 / 2 / 4 / 8 calling threads, one
context each, every thread rendering for --seconds.  --devices N spreads the contexts round-robin over N devices
(in-process multi-GPU: what an OFX host with eRenderFullySafe render threads does); more devices than the box has are LOGICAL devices over
the physical ones (OFXCV_VIRTUAL_DEVICES, set here): per-device locks (OFXCV_LOCK_PER_DEVICE=1) and per-device caches of named frames run as
on an N-GPU node.  With --sequence the frames go to devices the way the VectorGenerator plugin sends them: blocks of --block consecutive
frame times per device (--block 0: by calling thread, the round-4 behaviour).
usage: python tools/bench_host_threads.py [--devices N] [--seconds S] [--threads 1,2,4,8]     (BENCH_CTX_OPTIONS=opt=val,...)"""
import argparse, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--devices", type=int, default=1)
ap.add_argument("--seconds", type=float, default=1.5)
ap.add_argument("--threads", default="1,2,4,8")
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--block", type=int, default=16, help="frames per device block in --sequence mode (0 = device by calling thread)")
ap.add_argument("--sequence", type=int, default=0, help="N > 0: the threads render the output frames of ONE sequence of N distinct frames in order "
                "(a free thread takes the next frame, as a host hands them out), every frame named (ofxcv_vectorgen_flows_host_keyed): what an OFX host that provides "
                "kOfxImagePropUniqueIdentifier gets during playback")
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
shot = synth.sequence(W, H, 4)          # consecutive frames of one shot
prev, ref, nxt = shot[0], shot[1], shot[2]
if args.devices > torch.cuda.device_count():
    os.environ["OFXCV_VIRTUAL_DEVICES"] = str(args.devices)
ndev = max(1, min(args.devices, ofxcv.lib().ofxcv_device_count()))
for nt in [int(v) for v in args.threads.split(",")]:
    ctxs = [ofxcv.Context(i % ndev) for i in range(nt)]
    by_block = args.sequence and args.block > 0 and ndev > 1
    per_dev = [[ofxcv.Context(d) for d in range(ndev)] for _ in range(nt)] if by_block else None  # a render thread's context per device
    for kv in filter(None, os.environ.get("BENCH_CTX_OPTIONS", "").split(",")):
        for c in ctxs:
            c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    # every thread its own frames (as a host would hand them)
    frames = [(ref.copy(), nxt.copy(), prev.copy(), np.zeros((H, W, 4), np.float32)) for _ in range(nt)]
    for c, f in zip(ctxs, frames):
        for _ in range(2):
            c.vectorgen_flows_host(f[0], f[1], f[2], f[3], 1, 2, 4, 8)
    counts = [0] * nt
    stop = threading.Event()
    if args.sequence:
        nseq = max(6, args.sequence // 6 * 6)                                # whole periods of the forth-and-back run: the wrap-around is seamless
        seq = [shot[synth.pingpong(k, 4)].copy() for k in range(nseq)]        # N buffers (distinct names; the pixels repeat: generation is slow)
        args.sequence = nseq
        ctxs[0].host_cache_clear()
    next_frame = [0]
    frame_lock = threading.Lock()
    def work(i):
        c, f = ctxs[i], frames[i]
        while not stop.is_set():
            if args.sequence:
                n = args.sequence
                with frame_lock:    # a host hands out the output frames of a sequence in order, whichever render thread is free
                    t = next_frame[0]
                    next_frame[0] += 1
                a, b, p = t % n, (t + 1) % n, (t - 1) % n
                # names by position in an endless sequence (the N buffers come round again under new names): every output frame
                # has ONE frame the device has not seen, as in playback
                cc = per_dev[i][(t // args.block) % ndev] if by_block else c
                cc.vectorgen_flows_host(seq[a], seq[b], seq[p], f[3], 1, 2, 4, 8, keys=("f%d" % t, "f%d" % (t + 1), "f%d" % (t - 1)))
            else:
                c.vectorgen_flows_host(f[0], f[1], f[2], f[3], 1, 2, 4, 8)
            counts[i] += 1
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
    [t.start() for t in th]
    time.sleep(args.seconds)
    stop.set()
    [t.join() for t in th]
    el = time.perf_counter() - t0
    n = sum(counts)
    print("%d devices, %d calling threads: %.1f output frames/s = %.0f pairs/s (%.2f ms per call per thread; zero-copy calls %s)" %
          (ndev, nt, n / el, 2 * n / el, el * nt / max(1, n) * 1e3, [c.host_zero_copy_calls() for c in ctxs]) +
          ("  named frames found on the device / uploaded: %d / %d" % (sum(c.host_cache_hits() for c in ctxs), sum(c.host_cache_misses() for c in ctxs)) if args.sequence else ""), flush=True)
    allc = ctxs + ([c for row in per_dev for c in row] if per_dev else [])
    if args.sequence:
        print("    named frames over all contexts: found %d / uploaded %d" % (sum(c.host_cache_hits() for c in allc), sum(c.host_cache_misses() for c in allc)), flush=True)
    co = [c.host_coalesce_stats() for c in allc]
    if sum(x[0] for x in co):
        print("    coalesced calls %d of %d, mean pairs of the batched call they rode in %.2f" % (sum(x[0] for x in co), n, sum(x[2] for x in co) / sum(x[0] for x in co)), flush=True)
    lh = [c.lock_hold() for c in allc]
    print("    runtime lock: %.1f us held per Farneback call (%d holds)" % (sum(a for a, _ in lh) / 1e3 / max(1, sum(b for _, b in lh)), sum(b for _, b in lh)), flush=True)
    for c in allc:
        c.close()
