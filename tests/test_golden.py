"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle on seeded inputs): the oracle
and the synthetic generators must still reproduce them (CPU), and the HIP path must match them through the C ABI (GPU)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_oracle_reproduces_golden_vectors(oracle):
    from openfx_opencv_amd import synth
    g = _load("farneback_96x72.npz")
    a, b = synth.flow_pair(96, 72, seed=1234)
    assert np.array_equal(oracle.to_byte_grayscale(a), g["gray_a"]) and np.array_equal(oracle.to_byte_grayscale(b), g["gray_b"])
    assert np.array_equal(oracle.calc_optical_flow_farneback(g["gray_a"], g["gray_b"], blur_mode=oracle.BLUR_FAITHFUL), g["flow_faithful"])
    assert np.array_equal(oracle.calc_optical_flow_farneback(g["gray_a"], g["gray_b"], blur_mode=oracle.BLUR_DIRECT), g["flow_direct"])
    s = _load("srgb_lut.npz")
    assert np.array_equal(oracle.srgb_lut(), s["lut"]) and np.array_equal(oracle.to_byte_grayscale(s["ramp"]), s["ramp_gray"])
    i = _load("inpaint_96x72.npz")
    assert np.array_equal(synth.inpaint_frame(96, 72, seed=1234, hole_seed=42, n_holes=4), i["frame"])
    assert np.array_equal(oracle.inpaint_mask(i["frame"], 1), i["mask"])
    out, t, f, order = oracle.inpaint_telea(np.ascontiguousarray(i["frame"][..., :3]), i["mask"], 3.0, maps=True)
    assert np.array_equal(out, i["out"]) and np.array_equal(t, i["t"]) and np.array_equal(order, i["order"])
    assert np.array_equal(oracle.inpaint_render(i["frame"], 3.0, 1.0), i["render"])
    m = _load("meanshift_96x72.npz")
    assert np.array_equal(oracle.pyr_mean_shift(m["img"], 10.0, 20.0, 2), m["out"])


@pytest.mark.gpu
def test_hip_path_matches_golden_vectors(gpu_ctx, direct_ctx, ofxcv):
    g = _load("farneback_96x72.npz")
    # default mode (OpenCV-order window): every sample within 1e-4 of the faithful evaluation
    got_s = gpu_ctx.calc_optical_flow_farneback(_dev(g["gray_a"]), _dev(g["gray_b"])).cpu().numpy()
    assert (np.abs(got_s - g["flow_faithful"]) <= 1e-4 * np.maximum(1, np.abs(g["flow_faithful"]))).all()
    # direct-window mode: identical to the oracle's direct evaluation, within 1e-4 of the faithful one at all but a few samples
    got = direct_ctx.calc_optical_flow_farneback(_dev(g["gray_a"]), _dev(g["gray_b"])).cpu().numpy()
    assert np.array_equal(got, g["flow_direct"])
    err = np.abs(got - g["flow_faithful"])
    assert (err <= 1e-4 * np.maximum(1, np.abs(g["flow_faithful"]))).mean() > 0.998
    s = _load("srgb_lut.npz")
    assert np.array_equal(gpu_ctx.to_byte_grayscale(_dev(s["ramp"])).cpu().numpy(), s["ramp_gray"])
    i = _load("inpaint_96x72.npz")
    assert np.array_equal(gpu_ctx.inpaint_mask(_dev(i["frame"]), 1).cpu().numpy(), i["mask"])
    dst, t, order = gpu_ctx.inpaint_telea(_dev(np.ascontiguousarray(i["frame"][..., :3])), _dev(i["mask"]), 3.0, maps=True)
    assert np.array_equal(dst.cpu().numpy(), i["out"]) and np.array_equal(t.cpu().numpy(), i["t"]) and np.array_equal(order.cpu().numpy(), i["order"])
    assert np.array_equal(gpu_ctx.inpaint_render_host(i["frame"], 3.0, 1.0), i["render"])
    m = _load("meanshift_96x72.npz")
    assert np.array_equal(gpu_ctx.pyr_mean_shift_filtering(_dev(m["img"]), 10.0, 20.0, 2).cpu().numpy(), m["out"])
