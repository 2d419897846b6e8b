#!/usr/bin/env python3
"""Launches the level-0 fused iteration kernel N times on random planes (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openfx_opencv_amd as ofxcv
W, H, N = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = ofxcv.Context(0)
with torch.cuda.stream(ctx.stream):
    torch.manual_seed(0)
    R0 = torch.randn((5, H, W), device="cuda") * 20; R1 = torch.randn((5, H, W), device="cuda") * 20
    flow = torch.randn((H, W, 2), device="cuda") * 2
    M = ctx.farneback_update_matrices(R0, R1, flow)
    Mo = torch.zeros_like(M); fl = torch.zeros_like(flow)
    for _ in range(N):
        ctx.farneback_update_flow_blur(R0, R1, M, W, 3, True, fl, Mo)
    torch.cuda.synchronize()
