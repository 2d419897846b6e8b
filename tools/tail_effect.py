#!/usr/bin/env python3
"""Level-0 fused-iteration launch time as a function of the number of workgroups (frame height): is there a tail effect?
Run under rocprofv3 --kernel-trace and read tools/trace_by_grid.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openfx_opencv_amd as ofxcv
from openfx_opencv_amd import synth
W = 1920
for H in [int(v) for v in sys.argv[1:]] or [1036, 1050, 1064, 1080, 1092, 1106]:
    a, b = synth.flow_pair(W, H)
    c = ofxcv.Context(0)
    with torch.cuda.stream(c.stream):
        ga, gb = c.to_byte_grayscale(torch.from_numpy(a).cuda()), c.to_byte_grayscale(torch.from_numpy(b).cuda())
        for _ in range(6):
            c.calc_optical_flow_farneback(ga, gb)
    torch.cuda.synchronize()
    c.close()
