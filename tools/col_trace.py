#!/usr/bin/env python3
"""Phase timeline of one workgroup of iterate_col_kernel (option farneback.col_trace): per wavefront and round the shader-clock
stamps at the phase boundaries, averaged over the rounds.  usage: python tools/col_trace.py [--size WxH] [--batch N]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--nw", type=int, default=8)
ap.add_argument("--rw", type=int, default=4)
ap.add_argument("--opts", default="", help="further options of the traced context: k=v,k=v")
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
prs = [synth.flow_pair(W, H, seed=100 + i) for i in range(args.batch)]
c = ofxcv.Context(0)
c.set_option("farneback.col_trace", 1)
for kv in filter(None, args.opts.split(",")):
    k_, v_ = kv.split("=")
    c.set_option(k_, int(v_))
ga = [c.to_byte_grayscale(torch.from_numpy(a).cuda()) for a, _ in prs]
gb = [c.to_byte_grayscale(torch.from_numpy(b).cuda()) for _, b in prs]
for _ in range(3):
    c.calc_optical_flow_farneback_batch(ga, gb)
torch.cuda.synchronize()
NW, RW = args.nw, args.rw
rounds = (H + 2 + NW * RW - 1) // (NW * RW)
n = rounds * NW * 16
buf = (C.c_ulonglong * n)()
rc = ofxcv.lib().ofxcv_debug_col_trace(c._h, buf, C.c_int(n))
assert rc == 0, rc
t = np.array(buf, dtype=np.uint64).reshape(rounds, NW, 16)[:, :, :10].astype(np.int64)
t[:, :, 3] = np.where(t[:, :, 3] < t[:, :, 2], t[:, :, 2], t[:, :, 3])  # stamp 3 exists in the ring form only
t0 = t[0, 0, 0]
names = ["d rows loaded+summed", "chain 1 wait", "ring fill issue + wait (0 without the ring: stamp 3 unset)", "step 1 rows (M')", "boundary 1 wait + sums", "chain 2 wait", "solves 2", "update 2 (M'')", "boundary 2 + stores"]
print("clock: s_memtime ticks (100 MHz constant clock on gfx9: 1 tick = 10 ns)" )
print("launch span of the traced workgroup: %d ticks" % (t[-1, :, 9].max() - t0))
dur = np.diff(t, axis=2)  # [rounds][NW][9]
mid = dur[1:-1]           # full rounds only
print("per phase, mean over rounds 1..%d and wavefronts (ticks), and share of a round:" % (rounds - 2))
tot = mid.sum(axis=2).mean()
for k, nm in enumerate(names):
    print("  %-26s %8.1f  %5.1f %%   (min %d max %d)" % (nm, mid[:, :, k].mean(), 100 * mid[:, :, k].mean() / tot, mid[:, :, k].min(), mid[:, :, k].max()))
print("  %-26s %8.1f" % ("round", tot))
print("round start skew between wavefronts (ticks after wavefront 0), round %d:" % (rounds // 2), (t[rounds // 2, :, 0] - t[rounds // 2, 0, 0]).tolist())
print("per-wavefront timeline of round %d (ticks since its start):" % (rounds // 2))
for u in range(NW):
    print("  wave %2d:" % u, (t[rounds // 2, u, :] - t[rounds // 2, u, 0]).tolist())
