#!/usr/bin/env python3
"""Do the copies and the kernels of several calling threads overlap on one device?  Per thread count: iterations/s of
(a) copies only (3 pinned H2D of a 33 MB frame + 1 D2H), (b) compute only (3 gray LUTs + one batched Farneback call of 2 pairs
on resident frames), (c) both in sequence per thread (what a host call does)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W, H = 1920, 1080
a, b = synth.flow_pair(W, H, seed=11)
NT = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4").split(",")]
keep = []
for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("copy", "compute", "both", "pcopy", "pboth")):  # p...: pageable host buffers
    for nt in NT:
        ctxs = [ofxcv.Context(0) for _ in range(nt)]
        st = []
        for c in ctxs:  # (tensors on torch's default stream: they outlive the contexts' streams)
            pageable = mode.startswith("p")
            pins = [torch.from_numpy(a).clone() if pageable else torch.from_numpy(a).clone().pin_memory() for _ in range(3)]
            devs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
            for d, p in zip(devs, pins):
                d.copy_(p)
            out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
            pout = torch.empty((H, W, 4), dtype=torch.float32) if pageable else torch.empty((H, W, 4), dtype=torch.float32).pin_memory()
            g = [torch.empty((H, W), dtype=torch.uint8, device="cuda") for _ in range(3)]
            fl = [torch.empty((H, W, 2), dtype=torch.float32, device="cuda") for _ in range(2)]
            st.append((pins, devs, out, pout, g, fl))
        torch.cuda.synchronize()
        counts = [0] * nt
        stop = threading.Event()
        def work(i):
            c = ctxs[i]
            pins, devs, out, pout, g, fl = st[i]
            with torch.cuda.stream(c.stream):
                while not stop.is_set():
                    if mode in ("copy", "both", "pcopy", "pboth"):
                        for d, p in zip(devs, pins):
                            d.copy_(p, non_blocking=True)
                    if mode in ("compute", "both", "pboth"):
                        for d, gg in zip(devs, g):
                            c.to_byte_grayscale(d, gg)
                        c.calc_optical_flow_farneback_batch([g[0], g[0]], [g[1], g[2]], fl)
                        c.flow_to_rgba(fl[0], out, 1, 2)
                    if mode in ("copy", "both", "pcopy", "pboth"):
                        pout.copy_(out, non_blocking=True)
                    c.stream.synchronize()
                    counts[i] += 1
        th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        time.sleep(1.0)
        stop.set()
        [t.join() for t in th]
        el = time.perf_counter() - t0
        print("%-8s %d threads: %.1f iterations/s (%.2f ms per iteration per thread)" % (mode, nt, sum(counts) / el, el * nt / max(1, sum(counts)) * 1e3), flush=True)
        keep.append((ctxs, st))  # (not closed: torch's pinned-memory allocator still holds events on the contexts' streams)
