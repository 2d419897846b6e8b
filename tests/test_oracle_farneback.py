"""Known-answer tests that pin the CPU oracle's Farneback restatement without OpenCV.

The reference holds no tests or golden vectors and its arithmetic lives in un-vendored OpenCV, so the
oracle is 'parity unpinned'; these analytic checks are what anchors it.  CPU only.
"""
import numpy as np
import pytest


def test_cv_round_is_half_to_even(oracle):
    assert [oracle.cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 17.5)] == [0, 2, 2, 0, -2, 2, 18]


def test_gaussian_kernel_fixed_and_exp(oracle):
    assert np.array_equal(oracle.gaussian_kernel(3, 0.0), np.array([0.25, 0.5, 0.25], np.float32))
    k = oracle.gaussian_kernel(9, 1.5)
    x = np.arange(9) - 4.0
    ref = np.exp(-x * x / (2 * 1.5 * 1.5))
    ref /= ref.sum()
    assert np.allclose(k, ref, rtol=1e-6, atol=0) and abs(k.sum() - 1) < 1e-6 and np.array_equal(k, k[::-1])


def test_level_geometry_1080p(oracle):
    # levels=3 -> 4 resolutions; blur taps 3,3,9,19 for k=0..3 (sigma 0, .5, 1.5, 3.5)
    assert oracle.farneback_num_levels(1920, 1080, 0.5, 3) == 3
    assert oracle.farneback_num_levels(1920, 1080, 0.5, 10) == 5       # 1080/2^5 = 33.75 >= 32, /2^6 < 32
    assert oracle.farneback_num_levels(64, 48, 0.5, 3) == 0
    geo = [oracle.farneback_level_geom(1920, 1080, 0.5, k) for k in range(4)]
    assert geo == [(1920, 1080, 0.0, 3), (960, 540, 0.5, 3), (480, 270, 1.5, 9), (240, 135, 3.5, 19)]
    assert oracle.farneback_level_geom(100, 75, 0.5, 1)[:2] == (50, 38)  # cvRound(37.5) = 38 (half to even)


def test_resize_half_is_2x2_mean_and_identity_is_copy(oracle):
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 255, size=(20, 36)).astype(np.float32)
    half = oracle.resize_linear(a, 18, 10)
    ref = ((a[0::2, 0::2] * np.float32(0.5) + a[0::2, 1::2] * np.float32(0.5)) * np.float32(0.5)
           + (a[1::2, 0::2] * np.float32(0.5) + a[1::2, 1::2] * np.float32(0.5)) * np.float32(0.5))
    assert np.array_equal(half, ref)
    assert np.array_equal(oracle.resize_linear(a, 36, 20), a)
    q = oracle.resize_linear(a[:, :32], 8, 5)        # /4: mean of the two central samples per axis
    assert np.allclose(q[1, 2], a[5:7, 9:11].mean(), rtol=1e-6)


def test_flow_prolongation_rule(oracle):
    # upsampling x2: fx = dx/2 - 0.25 -> samples 0, 0 (.25), 0 (.75), 1 (.25) ...; borders clamp with zero fraction
    src = np.arange(6, dtype=np.float32).reshape(1, 6, 1).repeat(2, axis=0).repeat(2, axis=2)
    up = oracle.resize_linear(src, 12, 4)
    assert np.allclose(up[0, :, 0], [0, 0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3.25, 3.75, 4.25, 4.75, 5.0])


def test_gaussian_blur_constant_and_reflect101(oracle):
    c = np.full((9, 13), 77.0, np.float32)
    for ks, sg in ((3, 0.0), (9, 1.5), (19, 3.5)):
        assert np.allclose(oracle.gaussian_blur(c, ks, sg), 77.0, rtol=1e-6)
    # BORDER_REFLECT_101: a linear ramp stays a ramp in the interior and is mirrored (not replicated) at the ends
    ramp = np.tile(np.arange(16, dtype=np.float32), (4, 1))
    b = oracle.gaussian_blur(ramp, 3, 0.0)
    assert np.allclose(b[:, 1:-1], ramp[:, 1:-1]) and np.allclose(b[:, 0], 0.5) and np.allclose(b[:, -1], 14.5)


def test_polyexp_recovers_quadratic_exactly(oracle):
    # I = a + b x + c y + d x^2 + e y^2 + f xy  ->  R = [c, b, e, d, f] at interior pixels (border: replicated rows/cols)
    h, w = 40, 48
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    a, b, c, d, e, f = 3.0, 0.7, -1.3, 0.05, -0.02, 0.03
    I = (a + b * x + c * y + d * x * x + e * y * y + f * x * y).astype(np.float32)
    R = oracle.polyexp(I, 5, 1.1)
    xi, yi = x[8:-8, 8:-8], y[8:-8, 8:-8]
    inner = R[8:-8, 8:-8]
    assert np.allclose(inner[..., 0], c + 2 * e * yi + f * xi, atol=2e-3)   # d/dy
    assert np.allclose(inner[..., 1], b + 2 * d * xi + f * yi, atol=2e-3)   # d/dx
    assert np.allclose(inner[..., 2], e, atol=2e-4)                          # yy
    assert np.allclose(inner[..., 3], d, atol=2e-4)                          # xx
    assert np.allclose(inner[..., 4], f, atol=2e-4)                          # xy


def test_polyexp_prepare_matches_direct_inverse(oracle):
    g, xg, xxg, ig = oracle.polyexp_prepare(5, 1.1)
    n = 5
    G = np.zeros((6, 6))
    gd = g.astype(np.float64)
    for yy in range(-n, n + 1):
        for xx in range(-n, n + 1):
            wgt = gd[yy + n] * gd[xx + n]
            G[0, 0] += wgt; G[1, 1] += wgt * xx * xx; G[3, 3] += wgt * xx ** 4; G[5, 5] += wgt * xx * xx * yy * yy
    G[2, 2] = G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = G[1, 1]
    G[4, 4] = G[3, 3]
    G[3, 4] = G[4, 3] = G[5, 5]
    inv = np.linalg.inv(G)
    assert np.allclose(ig, [inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5]], rtol=1e-5)
    assert abs(g.sum() - 1) < 1e-6 and np.allclose(xg, np.arange(-n, n + 1) * g) and np.allclose(xxg, np.arange(-n, n + 1) ** 2 * g)


def test_update_matrices_border_scale_table(oracle):
    # flow 0 and R1 == R0 in the interior: r2 = r3 = 0; M depends on r4, r5, r6 scaled by the border table product
    h = w = 12
    R = np.zeros((h, w, 5), np.float32)
    R[..., 2] = 2.0   # r4
    R[..., 3] = 3.0   # r5
    M = oracle.update_matrices(R, R, np.zeros((h, w, 2), np.float32))
    tab = np.ones(12, np.float32)
    tab[:5] = [0.14, 0.14, 0.4472, 0.4472, 0.4472]
    tab[-5:] = tab[:5][::-1]
    for (yy, xx) in [(5, 5), (6, 6), (0, 5), (2, 6), (5, 1), (0, 0), (3, 4)]:
        s = np.float32(tab[xx]) * np.float32(tab[yy])
        if (xx, yy) == (5, 5) or (xx, yy) == (6, 6):
            s = np.float32(1.0)
        r4, r5 = np.float32(2.0) * s, np.float32(3.0) * s
        assert M[yy, xx, 0] == r4 * r4 and M[yy, xx, 2] == r5 * r5 and M[yy, xx, 1] == 0 and M[yy, xx, 3] == 0
    # last row / column sample R1 out of range -> the "else" branch (r6 halves, r2 = R0[0]/2)
    R2 = R.copy()
    R2[..., 0] = 4.0
    M2 = oracle.update_matrices(R2, R2, np.zeros((h, w, 2), np.float32))
    assert M2[5, 5, 3] == 0.0 and M2[5, w - 1, 3] != 0.0 and M2[h - 1, 5, 3] != 0.0


def test_blur_modes_agree_and_solve_is_regularised(oracle):
    rng = np.random.default_rng(3)
    h, w = 33, 47
    R0 = rng.normal(0, 20, size=(h, w, 5)).astype(np.float32)
    R1 = rng.normal(0, 20, size=(h, w, 5)).astype(np.float32)
    M = oracle.update_matrices(R0, R1, rng.normal(0, 1, size=(h, w, 2)).astype(np.float32))
    fa, Ma = oracle.update_flow_blur(R0, R1, M, 3, True, oracle.BLUR_FAITHFUL)
    fd, Md = oracle.update_flow_blur(R0, R1, M, 3, True, oracle.BLUR_DIRECT)
    assert (np.abs(fa - fd) <= 1e-4 * np.maximum(1, np.abs(fa))).mean() > 0.99   # random M: a few ill-conditioned solves
    # M == 0 -> det = 1e-3 regulariser only -> flow == 0
    f0, _ = oracle.update_flow_blur(R0, R1, np.zeros_like(M), 3, False, oracle.BLUR_DIRECT)
    assert np.array_equal(f0, np.zeros_like(f0))
    # single-pixel check of the direct evaluation against numpy
    yy, xx = 10, 20
    win = M[yy - 1:yy + 2, xx - 1:xx + 2].astype(np.float64).sum(axis=(0, 1)) / 9.0
    idet = 1.0 / (win[0] * win[2] - win[1] * win[1] + 1e-3)
    ref = np.array([(win[0] * win[4] - win[1] * win[3]) * idet, (win[2] * win[3] - win[1] * win[4]) * idet])
    f1, _ = oracle.update_flow_blur(R0, R1, M, 3, False, oracle.BLUR_DIRECT)
    assert np.allclose(f1[yy, xx], ref, rtol=1e-5)


def test_farneback_recovers_translation(oracle):
    """A smooth texture shifted by (+1.5, -0.75) px: interior flow close to that vector (SURVEY.md KAT 4)."""
    from openfx_opencv_amd import synth
    h, w = 120, 160
    big = synth.texture(w + 16, h + 16, seed=5)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    a = synth._bilinear(big, xx + 8, yy + 8)
    b = synth._bilinear(big, xx + 8 - 1.5, yy + 8 + 0.75)
    ga, gb = (np.rint(a * 255)).astype(np.uint8), (np.rint(b * 255)).astype(np.uint8)
    for mode in (oracle.BLUR_FAITHFUL, oracle.BLUR_DIRECT):
        flow = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=mode)
        inner = flow[24:-24, 24:-24]
        assert abs(np.median(inner[..., 0]) - 1.5) < 0.1 and abs(np.median(inner[..., 1]) + 0.75) < 0.1


def test_faithful_and_direct_evaluations_differ_only_by_reference_rounding_noise(oracle):
    """OpenCV's box window rounds each vertical row difference to f32 before adding it to its f64 running sum.
    The oracle's DIRECT evaluation (what the HIP kernels compute) omits that rounding; the two agree within
    1e-4*max(1,|flow|) on all but a small fraction of samples (ill-conditioned pixels amplify the f32 noise)."""
    from openfx_opencv_amd import synth
    a, b = synth.flow_pair(320, 240)
    ga, gb = oracle.to_byte_grayscale(a), oracle.to_byte_grayscale(b)
    f0 = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL)
    f1 = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_DIRECT)
    err = np.abs(f0 - f1)
    bad = err > 1e-4 * np.maximum(1, np.abs(f0))
    assert bad.mean() < 2e-3
    assert np.median(err) < 1e-5


def test_srgb_lut_bytes_round_trip(oracle):
    lut = oracle.srgb_lut()
    assert lut.shape == (65536,) and lut[0] == 0 and lut.max() == 0xff00
    # every byte value survives decode -> encode (the table is patched so that it does)
    b = np.arange(256, dtype=np.float32) / np.float32(255)
    lin = np.where(b < 0.04045, b / 12.92, ((b + 0.055) / 1.055) ** 2.4).astype(np.float32)
    img = np.stack([lin, lin, lin], axis=-1)[None]
    got = oracle.to_byte_grayscale(img)[0]
    assert np.abs(got.astype(int) - np.arange(256)).max() <= 1
    # luma weights: pure green is brighter than pure red is brighter than pure blue
    px = np.array([[[1, 0, 0, 1], [0, 1, 0, 1], [0, 0, 1, 1], [1, 1, 1, 1], [0, 0, 0, 1]]], np.float32)
    g = oracle.to_byte_grayscale(px)[0]
    assert g[1] > g[0] > g[2] and g[3] == 255 and g[4] == 0


def test_flow_to_rgba_leaves_unmapped_channels(oracle):
    flow = np.arange(2 * 3 * 2, dtype=np.float32).reshape(2, 3, 2)
    dst = np.full((2, 3, 4), -1.0, np.float32)
    oracle.flow_to_rgba(flow, dst, [1, 0, 0, 0], [0, 1, 0, 0], 0.5, 0.25)
    assert np.array_equal(dst[..., 0], flow[..., 0] / 0.5) and np.array_equal(dst[..., 1], flow[..., 1] / 0.25)
    assert (dst[..., 2:] == -1.0).all()


# ---- OPTFLOW_USE_INITIAL_FLOW / OPTFLOW_FARNEBACK_GAUSSIAN ----

def test_resize_area_integer_factor_is_the_cell_mean(oracle):
    rng = np.random.default_rng(5)
    a = rng.random((48, 64, 2), dtype=np.float32)
    for f in (2, 4, 8):
        r = oracle.resize_area(a, 64 // f, 48 // f)
        ref = a.reshape(48 // f, f, 64 // f, f, 2).astype(np.float64).mean(axis=(1, 3))
        assert np.abs(r - ref).max() < 1e-6
    assert np.array_equal(oracle.resize_area(a, 64, 48), a)  # equal sizes: a copy


def test_resize_area_fractional_factor_weights_sum_to_one(oracle):
    c = np.full((37, 53), 3.25, np.float32)
    for (dw, dh) in [(7, 5), (20, 9), (52, 36)]:
        assert np.abs(oracle.resize_area(c, dw, dh) - 3.25).max() < 1e-5
    # a horizontal ramp keeps its mean and stays monotone
    ramp = np.tile(np.arange(53, dtype=np.float32), (37, 1))
    r = oracle.resize_area(ramp, 7, 5)
    assert abs(float(r.mean()) - 26.0) < 1e-3 and np.all(np.diff(r[0]) > 0)


def test_gaussian_window_taps_are_normalised(oracle):
    """A constant M field passes the Gaussian window unchanged: the flow is the solve of the constants."""
    h, w = 24, 32
    M = np.empty((h, w, 5), np.float32)
    M[:] = np.array([2.0, 0.5, 3.0, 1.0, -2.0], np.float32)
    R = np.zeros((h, w, 5), np.float32)
    for ws in (3, 5, 9):
        flow, _ = oracle.update_flow_gaussian(R, R, np.zeros((h, w, 2), np.float32), M, ws, False)
        idet = 1.0 / (2.0 * 3.0 - 0.25 + 1e-3)
        assert np.allclose(flow[..., 0], (2.0 * -2.0 - 0.5 * 1.0) * idet, rtol=1e-5)
        assert np.allclose(flow[..., 1], (3.0 * 1.0 - 0.5 * -2.0) * idet, rtol=1e-5)


def test_gaussian_flag_and_initial_flow_recover_the_translation(oracle):
    from openfx_opencv_amd import synth
    a, b = synth.flow_pair(160, 120)
    ga, gb = oracle.to_byte_grayscale(a), oracle.to_byte_grayscale(b)
    u, v = synth.known_flow(160, 120)
    truth = np.stack([u, v], axis=-1)[24:-24, 24:-24]
    box = oracle.calc_optical_flow_farneback(ga, gb)
    gau = oracle.calc_optical_flow_farneback(ga, gb, winsize=5, flags=oracle.OPTFLOW_FARNEBACK_GAUSSIAN)
    err = lambda f: float(np.abs(f[24:-24, 24:-24] - truth).mean())
    assert err(gau) < 0.5 and err(box) < 0.5
    # two iterations on one level from the converged flow stay there; from zero they do not get close
    warm = oracle.calc_optical_flow_farneback(ga, gb, levels=0, iterations=2, flags=oracle.OPTFLOW_USE_INITIAL_FLOW, initial_flow=box)
    cold = oracle.calc_optical_flow_farneback(ga, gb, levels=0, iterations=2)
    assert err(warm) < 0.5 < err(cold)
    with pytest.raises(ValueError):
        oracle.calc_optical_flow_farneback(ga, gb, flags=1)


def test_resize_generations_are_the_same_2x2_mean_up_to_association(oracle):
    """cv::resize(INTER_LINEAR) at an exact 2x reduction is INTER_AREA's block mean; the three known associations of its float
    additions agree with the exact mean to one ulp, generation 0 IS the bilinear form, and other scales are untouched."""
    rng = np.random.default_rng(5)
    src = rng.uniform(0, 255, size=(48, 64)).astype(np.float32)
    exact = (src.astype(np.float64).reshape(24, 2, 32, 2).sum(axis=(1, 3)) * 0.25)
    outs = []
    try:
        for gen in (0, 1, 2):
            oracle.set_resize_generation(gen)
            out = oracle.resize_linear(src, 32, 24)
            assert np.abs(out - exact).max() <= np.spacing(np.float32(255.0))
            outs.append(out)
            other = oracle.resize_linear(src, 16, 12)   # 4x: stays bilinear in every generation
            if gen == 0:
                other0 = other
            assert np.array_equal(other, other0)
    finally:
        oracle.set_resize_generation(0)
    a, b, c = src[0::2, 0::2], src[0::2, 1::2], src[1::2, 0::2]
    d = src[1::2, 1::2]
    assert np.array_equal(outs[0], ((a + b) + (c + d)) * np.float32(0.25))
    assert np.array_equal(outs[1], (((a + b) + c) + d) * np.float32(0.25))
    assert np.array_equal(outs[2], ((a + c) + (b + d)) * np.float32(0.25))
    assert not np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[0], outs[2])
