import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
from oracle import binding as oracle
w, h = 320, 240
a, b = synth.flow_pair(w, h)
c, _ = synth.flow_pair(w, h, seed=77)
ga, gb, gc = (oracle.to_byte_grayscale(x) for x in (a, b, c))
fwd = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL)
bwd = oracle.calc_optical_flow_farneback(ga, gc, blur_mode=oracle.BLUR_FAITHFUL)
def rep(name, got, ref):
    err = np.abs(got - ref); bad = err > 1e-4 * np.maximum(1, np.abs(ref))
    print("%-40s max err %.3g outside %d differing %d" % (name, err.max(), bad.sum(), (got != ref).sum()), flush=True)
for opts in [{}, {"farneback.graph": 0}, {"farneback.prep_stream": 0}]:
    ctx = ofxcv.Context(0)
    for k, v in opts.items(): ctx.set_option(k, v)
    out = np.zeros((h, w, 4), np.float32)
    ctx.vectorgen_flows_host(a, b, c, out, 1, 2, 4, 8)
    rep("flows_host fwd %s" % opts, out[..., :2], fwd); rep("flows_host bwd %s" % opts, out[..., 2:], bwd)
    out = np.zeros((h, w, 4), np.float32)
    ctx.vectorgen_flow_host(a, c, out, 4, 8)
    rep("flow_host bwd alone %s" % opts, out[..., 2:], bwd)
    f1 = ctx.calc_optical_flow_farneback(torch.from_numpy(ga).cuda(), torch.from_numpy(gb).cuda()).cpu().numpy()
    f2 = ctx.calc_optical_flow_farneback(torch.from_numpy(ga).cuda(), torch.from_numpy(gc).cuda()).cpu().numpy()
    rep("device fwd", f1, fwd); rep("device bwd (same ctx, after fwd)", f2, bwd)
    ctx.close()
