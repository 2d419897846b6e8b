#!/usr/bin/env python3
"""BASELINE config 5 shape: 3840x2160 pairs, 8 in flight on one GPU; parity of one pair vs the oracle + throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
from oracle import binding as oracle
W, H, P = 3840, 2160, int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctxs = [ofxcv.Context(0) for _ in range(P)]
bufs = []
for i, c in enumerate(ctxs):
    a, b = synth.flow_pair(W, H, seed=1234 + i)
    with torch.cuda.stream(c.stream):
        ga = c.to_byte_grayscale(torch.from_numpy(a).cuda()); gb = c.to_byte_grayscale(torch.from_numpy(b).cuda())
        bufs.append((ga, gb, torch.empty((H, W, 2), device="cuda")))
def step():
    for c, (ga, gb, fl) in zip(ctxs, bufs):
        with torch.cuda.stream(c.stream):
            c.calc_optical_flow_farneback(ga, gb, fl)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
for _ in range(n): step()
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("4K: %d pairs in flight: %.1f pairs/s (%.2f ms per pair), mem %.1f GB" % (P, P * n / el, el / n / P * 1e3, torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9))
ga, gb, fl = bufs[0]
# the library default is OpenCV's own summation order (since round 2): against the FAITHFUL oracle, every sample within 1e-4
t0 = time.perf_counter(); ref = oracle.calc_optical_flow_farneback(ga.cpu().numpy(), gb.cpu().numpy(), blur_mode=oracle.BLUR_FAITHFUL); print("oracle %.1f s" % (time.perf_counter() - t0))
got = fl.cpu().numpy()
err = np.abs(ref - got)
print("default mode vs FAITHFUL oracle: outside 1e-4: %d samples, max |err| %.3g, bit-identical %.6f" % (int((err > 1e-4 * np.maximum(1, np.abs(ref))).sum()), err.max(), (ref == got).mean()))
# ... and the opt-in direct-window mode against the oracle's DIRECT evaluation: bit for bit
d = ofxcv.Context(0)
d.set_option("farneback.opencv_rounding", 0)
got = d.calc_optical_flow_farneback(ga, gb).cpu().numpy()
ref = oracle.calc_optical_flow_farneback(ga.cpu().numpy(), gb.cpu().numpy(), blur_mode=oracle.BLUR_DIRECT)
print("direct-window mode bit-identical to the DIRECT oracle:", np.array_equal(ref, got))
