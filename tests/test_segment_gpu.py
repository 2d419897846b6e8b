"""Parity of the pyramid mean-shift filtering kernels (C ABI) against the CPU oracle: all-integer, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _img(w, h, seed=1234):
    from openfx_opencv_amd import synth
    return synth.inpaint_frame(w, h, seed=seed, n_holes=0)


@pytest.mark.parametrize("w,h,level,cn", [(64, 48, 0, 3), (160, 120, 2, 3), (161, 119, 2, 4), (97, 83, 1, 3), (200, 150, 3, 4)])
def test_mean_shift_bit_exact(oracle, gpu_ctx, w, h, level, cn):
    fr = _img(w, h)
    rgb = np.ascontiguousarray(fr[..., :3])
    ref = oracle.pyr_mean_shift(rgb, 10, 20, level)
    src = rgb if cn == 3 else fr
    got = gpu_ctx.pyr_mean_shift_filtering(_dev(src), 10, 20, level).cpu().numpy()
    assert np.array_equal(got[..., :3], ref), "differs at %d pixels" % (got[..., :3] != ref).any(axis=2).sum()
    if cn == 4:
        assert np.array_equal(got[..., 3], fr[..., 3])


def test_mean_shift_other_parameters_and_edges(oracle, gpu_ctx):
    fr = _img(120, 90, seed=7)[..., :3]
    for sp, sr, lv, it, eps in [(4, 12, 1, 3, 0.0), (1, 200, 0, 5, 1.0), (15, 3, 2, 100, 1.0), (10, 20, 2, 1, 5.0)]:
        ref = oracle.pyr_mean_shift(fr, sp, sr, lv, it, eps)
        got = gpu_ctx.pyr_mean_shift_filtering(_dev(fr), sp, sr, lv, it, eps).cpu().numpy()
        assert np.array_equal(got, ref), (sp, sr, lv, it, eps)
    const = np.full((33, 47, 3), (10, 120, 201), np.uint8)           # constant image -> identity
    assert np.array_equal(gpu_ctx.pyr_mean_shift_filtering(_dev(const)).cpu().numpy(), const)
    two = const.copy()
    two[:, 24:] = (200, 30, 90)                                        # two tones further apart than sr -> identity at level 0
    assert np.array_equal(gpu_ctx.pyr_mean_shift_filtering(_dev(two), 10, 20, 0).cpu().numpy(), two)
    tiny = _img(3, 2)[..., :3]                                         # smaller than the pyramid kernels
    assert np.array_equal(gpu_ctx.pyr_mean_shift_filtering(_dev(tiny), 10, 20, 2).cpu().numpy(), oracle.pyr_mean_shift(tiny, 10, 20, 2))


def test_segment_render_host(oracle, gpu_ctx):
    fr = _img(320, 240)
    got = gpu_ctx.segment_render_host(fr, 10, 20, 2)
    assert np.array_equal(got[..., :3], oracle.pyr_mean_shift(np.ascontiguousarray(fr[..., :3]), 10, 20, 2))
    assert (got[..., 3] == 255).all()


def test_mean_shift_4k_properties(oracle, gpu_ctx):
    """BASELINE config 4 size (3840x2160): the filter is deterministic, and a 256x256 crop filtered on its own by the
    oracle agrees with the full-frame result away from the crop border (the filter is local: sp * iterations pixels of
    reach) except where a window mean lands on a rounding tie -- cvRound(sum * (1./count)) is not translation
    invariant in the last bit, so a handful of pixels may differ."""
    import torch
    fr = _img(3840, 2160)
    d = _dev(np.ascontiguousarray(fr[..., :3]))
    full = gpu_ctx.pyr_mean_shift_filtering(d, 10, 20, 0)
    assert torch.equal(full, gpu_ctx.pyr_mean_shift_filtering(d, 10, 20, 0))
    y0, x0 = 1000, 2000
    crop = np.ascontiguousarray(fr[y0:y0 + 256, x0:x0 + 256, :3])
    ref = oracle.pyr_mean_shift(crop, 10, 20, 0)
    got = full[y0:y0 + 256, x0:x0 + 256].cpu().numpy()
    same = (got[64:-64, 64:-64] == ref[64:-64, 64:-64]).all(axis=2)
    assert same.mean() > 0.999, same.mean()


def test_mean_shift_4k_max_level_2_full_frame(oracle, gpu_ctx):
    """BASELINE config 4 exactly: 3840x2160, sp 10, sr 20, maxLevel 2 -- the whole frame against the oracle (about 10 s of CPU),
    every pixel identical (all-integer arithmetic)."""
    fr = np.ascontiguousarray(_img(3840, 2160)[..., :3])
    got = gpu_ctx.pyr_mean_shift_filtering(_dev(fr), 10, 20, 2).cpu().numpy()
    ref = oracle.pyr_mean_shift(fr, 10, 20, 2)
    assert np.array_equal(got, ref), "differing pixels: %d" % (got != ref).any(axis=2).sum()
