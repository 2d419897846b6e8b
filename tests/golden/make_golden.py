#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the CPU oracle on seeded synthetic inputs.

The reference (NatronGitHub/openfx-opencv) holds no test data, so these are NOT reference outputs: they freeze the
oracle's own results ('golden vectors: oracle outputs on seeded synthetic inputs') so that a
change of the oracle, of the synthetic generators or of the GPU kernels shows up as a diff against committed data.
Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as oracle  # noqa: E402
from openfx_opencv_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    # Farneback: 96x72 pair (two pyramid levels), default plugin parameters
    a, b = synth.flow_pair(96, 72, seed=1234)
    ga, gb = oracle.to_byte_grayscale(a), oracle.to_byte_grayscale(b)
    np.savez_compressed(os.path.join(OUT, "farneback_96x72.npz"), gray_a=ga, gray_b=gb,
                        flow_faithful=oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL),
                        flow_direct=oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_DIRECT))
    # sRGB gray LUT: all 65536 table entries + a ramp image
    ramp = np.linspace(-0.1, 1.2, 64 * 48 * 3, dtype=np.float32).reshape(48, 64, 3)
    np.savez_compressed(os.path.join(OUT, "srgb_lut.npz"), lut=oracle.srgb_lut(), ramp=ramp, ramp_gray=oracle.to_byte_grayscale(ramp))
    # Telea: 96x72 frame with 4 holes, radius 3, dilation 1
    fr = synth.inpaint_frame(96, 72, seed=1234, hole_seed=42, n_holes=4)
    mask = oracle.inpaint_mask(fr, 1)
    out, t, f, order = oracle.inpaint_telea(np.ascontiguousarray(fr[..., :3]), mask, 3.0, maps=True)
    np.savez_compressed(os.path.join(OUT, "inpaint_96x72.npz"), frame=fr, mask=mask, out=out, t=t, order=order,
                        render=oracle.inpaint_render(fr, 3.0, 1.0))
    # mean-shift: 96x72, sp 10, sr 20, two pyramid levels
    img = np.ascontiguousarray(synth.inpaint_frame(96, 72, seed=77, n_holes=0)[..., :3])
    np.savez_compressed(os.path.join(OUT, "meanshift_96x72.npz"), img=img, out=oracle.pyr_mean_shift(img, 10.0, 20.0, 2))
    for f_ in sorted(os.listdir(OUT)):
        if f_.endswith(".npz"):
            print("%-24s %7d bytes" % (f_, os.path.getsize(os.path.join(OUT, f_))))


if __name__ == "__main__":
    main()
