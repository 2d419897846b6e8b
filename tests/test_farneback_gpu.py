"""Parity of the HIP Farneback path (through the C ABI) against the CPU oracle.

The library's default evaluates the box window in OpenCV's order (running column sums with every row difference rounded
to f32, strip-parallel): every sample must be within 1e-4 (relative) of the FAITHFUL oracle -- the test_opencv_order_* tests
and tests/test_golden.py.  The direct-window mode (`direct_ctx`, option farneback.opencv_rounding = 0) is the faster opt-in
path; for it, two comparisons per case:
  * vs the oracle's DIRECT box-window evaluation (same IEEE operation order as the kernels):
    every stage must agree to the last bit;
  * vs the oracle's FAITHFUL restatement of OpenCV's running sums: |a-b| <= 1e-4*max(1,|b|)
    (the north_star tolerance).  OpenCV rounds each vertical row difference to f32 before adding it
    to its f64 running sum; that rounding noise is the reference's own, is not reproduced on the
    GPU, and at a few ill-conditioned pixels is amplified past 1e-4 -- the oracle's own two
    evaluations differ there by the same amount (tests/test_oracle_farneback.py measures it).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4           # north_star: 1e-4 relative float tolerance


def outlier_bound(w, h):
    """Samples the DIRECT-WINDOW mode (option farneback.opencv_rounding = 0: each 3x3 window summed on its own, the fast
    opt-in path) may have outside REL_TOL of the FAITHFUL oracle: the reference's own f32 rounding noise of its running
    column sums, which that mode does not reproduce, amplified at ill-conditioned pixels.  It grows with the column height:
    measured 1.6e-6 at 640x480, 6.1e-5 at 1920x1080, 6.5e-4 at 3840x2160; the toy sizes are a handful of samples (7 of 6144
    at 64x48).  The DEFAULT mode (OpenCV order) has NO such allowance: every sample within REL_TOL (tests below)."""
    if h > 1080:
        return 1e-3
    return 2e-4 if w * h >= 640 * 480 else 1.5e-3


def _gray_pair(oracle, w, h, seed=1234):
    from openfx_opencv_amd import synth
    a, b = synth.flow_pair(w, h, seed)
    return oracle.to_byte_grayscale(a), oracle.to_byte_grayscale(b)


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("w,h", [(64, 48), (160, 120), (100, 75), (333, 257)])
def test_pyr_image_bit_exact(oracle, ofxcv, direct_ctx, w, h):
    ga, _ = _gray_pair(oracle, w, h)
    levels = ofxcv.farneback_num_levels(w, h, 0.5, 5)
    for k in range(levels + 1):
        lw, lh, sigma, ks = ofxcv.farneback_level_geom(w, h, 0.5, k)
        assert (lw, lh, sigma, ks) == oracle.farneback_level_geom(w, h, 0.5, k)
        ref = oracle.farneback_pyr_image(ga, lw, lh, sigma, ks)
        got = direct_ctx.farneback_pyr_image(_dev(ga), lw, lh, sigma, ks).cpu().numpy()
        assert np.array_equal(ref, got), "level %d max diff %g" % (k, np.abs(ref - got).max())


@pytest.mark.parametrize("w,h", [(64, 64), (256, 64), (72, 200), (328, 200), (1000, 568), (1920, 1080)])
def test_pyr_image_quarter_and_eighth_levels(oracle, ofxcv, w, h):
    """The coarse levels of the default pyramid (exactly 1/4 and 1/8 of the frame, 9 / 19 taps) take pyr_fused_al_kernel (aligned
    dword staging, both filtered columns of an output column from one byte run): bit-identical to the oracle and to the
    byte-wise tile kernel (option farneback.fused_pyramid 2), with partial tiles, reflected edges on all four sides, and a source whose
    rows are not dword-aligned (which must fall back by itself)."""
    import torch
    ga, _ = _gray_pair(oracle, w, h)
    ctx = ofxcv.Context(0)
    try:
        for k in (2, 3):
            lw, lh, sigma, ks = ofxcv.farneback_level_geom(w, h, 0.5, k)
            assert (w, h) == (lw << k, lh << k) and ks == (9, 19)[k - 2]
            ref = oracle.farneback_pyr_image(ga, lw, lh, sigma, ks)
            ctx.set_option("farneback.fused_pyramid", 1)
            got = ctx.farneback_pyr_image(_dev(ga), lw, lh, sigma, ks).cpu().numpy()
            ctx.set_option("farneback.fused_pyramid", 2)
            other = ctx.farneback_pyr_image(_dev(ga), lw, lh, sigma, ks).cpu().numpy()
            ctx.set_option("farneback.fused_pyramid", 1)
            padded = torch.zeros((h, w + 3), dtype=torch.uint8, device="cuda")  # row step w + 3: not a multiple of 4
            padded[:, :w] = _dev(ga)
            odd = ctx.farneback_pyr_image(padded[:, :w], lw, lh, sigma, ks).cpu().numpy()
            assert np.array_equal(ref, got), "level %d max diff %g" % (k, np.abs(ref - got).max())
            assert np.array_equal(ref, other) and np.array_equal(ref, odd)
    finally:
        ctx.close()


@pytest.mark.parametrize("w,h", [(16, 4), (64, 48), (248, 66), (252, 34), (496, 130), (1000, 36), (1920, 1080)])
def test_pyr_image_three_tap_levels_wavefront_rows(oracle, ofxcv, w, h):
    """Levels 0 and 1 of the default pyramid (3 taps; same size / exactly half) on frames whose width is a multiple of four take
    pyr_direct3w_kernel (one dword per lane and source row, the neighbour bytes from the neighbour lanes, several output rows per wavefront):
    bit-identical to the oracle and to the one-group-per-lane kernel (option farneback.fused_pyramid 3) -- tile seams at 248 columns, both image
    edges inside one wavefront, heights that end inside a wavefront's rows -- with the filter contraction and resize generations on as well."""
    ga, _ = _gray_pair(oracle, w, h)
    ctx = ofxcv.Context(0)
    try:
        for fc, rz in ((0, 0), (1, 0), (0, 1), (1, 2)):
            oracle.set_filter_contraction(fc)
            oracle.set_resize_generation(rz)
            ctx.set_option("farneback.filter_contraction", fc)
            ctx.set_option("farneback.resize_generation", rz)
            for k in (0, 1):
                if k == 1 and (w % 2 or h % 2):
                    continue
                lw, lh, sigma, ks = ofxcv.farneback_level_geom(w, h, 0.5, k)
                ref = oracle.farneback_pyr_image(ga, lw, lh, sigma, ks)
                ctx.set_option("farneback.fused_pyramid", 1)
                got = ctx.farneback_pyr_image(_dev(ga), lw, lh, sigma, ks).cpu().numpy()
                ctx.set_option("farneback.fused_pyramid", 3)
                other = ctx.farneback_pyr_image(_dev(ga), lw, lh, sigma, ks).cpu().numpy()
                assert np.array_equal(ref, got), "level %d (fc %d, resize %d) max diff %g" % (k, fc, rz, np.abs(ref - got).max())
                assert np.array_equal(ref, other)
    finally:
        oracle.set_filter_contraction(0)
        oracle.set_resize_generation(0)
        ctx.close()


@pytest.mark.parametrize("w,h", [(64, 48), (160, 120), (98, 74), (640, 480)])
def test_pyr_image_generations_bit_exact(oracle, ofxcv, w, h):
    """The two places where OpenCV generations are known (from their published sources) to differ in the last bit, as matching
    switches of the oracle and of the library: getGaussianKernel (2.4 / 3.x vs 4.x) and the association of cv::resize's exact-2x
    INTER_AREA rewrite.  Every combination: pyramid images bit-identical, the whole call within 1e-4 at every sample."""
    ga, gb = _gray_pair(oracle, w, h)
    levels = ofxcv.farneback_num_levels(w, h, 0.5, 3)
    ctx = ofxcv.Context(0)
    seen = {}
    try:
        for gauss in (3, 4):
            for rz in (0, 1, 2):
                oracle.set_gaussian_kernel_generation(gauss)
                oracle.set_resize_generation(rz)
                ctx.set_option("farneback.gaussian_kernel_generation", gauss)
                ctx.set_option("farneback.resize_generation", rz)
                assert ctx.get_option("farneback.gaussian_kernel_generation") == gauss and ctx.get_option("farneback.resize_generation") == rz
                for fused in (1, 0):  # the dword / LDS-fused kernels and the two-pass fall-back
                    ctx.set_option("farneback.fused_pyramid", fused)
                    for k in range(levels + 1):
                        lw, lh, sigma, ks = ofxcv.farneback_level_geom(w, h, 0.5, k)
                        ref = oracle.farneback_pyr_image(ga, lw, lh, sigma, ks)
                        got = ctx.farneback_pyr_image(_dev(ga), lw, lh, sigma, ks).cpu().numpy()
                        assert np.array_equal(ref, got), "gauss %d resize %d fused %d level %d: max diff %g" % (gauss, rz, fused, k, np.abs(ref - got).max())
                        seen[(gauss, rz, k)] = ref
                ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL)
                got = ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb)).cpu().numpy()
                err = np.abs(ref - got)
                assert (err <= REL_TOL * np.maximum(1, np.abs(ref))).all(), "gauss %d resize %d: max err %g" % (gauss, rz, err.max())
    finally:
        oracle.set_gaussian_kernel_generation(3)
        oracle.set_resize_generation(0)
        ctx.close()
    if levels >= 1 and w % 2 == 0 and h % 2 == 0:
        # the associations differ by at most one ulp and only at level 1 (exactly half the frame)
        a, b, c = (seen[(3, rz, 1)] for rz in (0, 1, 2))
        for x in (b, c):
            assert np.abs(a.view(np.int32) - x.view(np.int32)).max() <= 1
        assert np.array_equal(seen[(3, 0, 0)], seen[(3, 2, 0)])
    with pytest.raises(ofxcv.OfxcvError):
        c2 = ofxcv.Context(0)
        try:
            c2.set_option("farneback.resize_generation", 3)
        finally:
            c2.close()


@pytest.mark.parametrize("w,h,n,sigma", [(64, 48, 5, 1.1), (160, 120, 5, 1.1), (97, 61, 7, 1.5), (33, 40, 5, 1.1)])
def test_polyexp_bit_exact(oracle, ofxcv, direct_ctx, w, h, n, sigma):
    rng = np.random.default_rng(7)
    I = (rng.uniform(0, 255, size=(h, w))).astype(np.float32)
    ref = oracle.polyexp(I, n, sigma)
    got = ofxcv.planes_to_hwc(direct_ctx.farneback_polyexp(_dev(I), n, sigma), w).cpu().numpy()
    assert np.array_equal(ref, got), "max diff %g" % np.abs(ref - got).max()


@pytest.mark.parametrize("w,h", [(64, 48), (161, 119), (12, 12)])
def test_update_matrices_bit_exact(oracle, ofxcv, direct_ctx, w, h):
    rng = np.random.default_rng(11)
    R0 = rng.normal(0, 20, size=(h, w, 5)).astype(np.float32)
    R1 = rng.normal(0, 20, size=(h, w, 5)).astype(np.float32)
    flow = rng.normal(0, 3, size=(h, w, 2)).astype(np.float32)
    flow[0, 0] = (-5.0, -5.0)          # sample falls outside: the else branch
    flow[h - 1, w - 1] = (4.0, 4.0)
    ref = oracle.update_matrices(R0, R1, flow)
    got = direct_ctx.farneback_update_matrices(ofxcv.hwc_to_planes(_dev(R0)), ofxcv.hwc_to_planes(_dev(R1)), _dev(flow))
    got = ofxcv.planes_to_hwc(got, w).cpu().numpy()
    assert np.array_equal(ref, got), "max diff %g" % np.abs(ref - got).max()


@pytest.mark.parametrize("w,h,update", [(64, 48, True), (161, 119, True), (75, 33, False)])
def test_update_flow_blur_matches_direct_oracle(oracle, ofxcv, direct_ctx, w, h, update):
    rng = np.random.default_rng(13)
    R0 = rng.normal(0, 20, size=(h, w, 5)).astype(np.float32)
    R1 = rng.normal(0, 20, size=(h, w, 5)).astype(np.float32)
    M = oracle.update_matrices(R0, R1, rng.normal(0, 1, size=(h, w, 2)).astype(np.float32))
    ref_flow, ref_M = oracle.update_flow_blur(R0, R1, M, 3, update, oracle.BLUR_DIRECT)
    flow, Mo = direct_ctx.farneback_update_flow_blur(ofxcv.hwc_to_planes(_dev(R0)), ofxcv.hwc_to_planes(_dev(R1)),
                                                  ofxcv.hwc_to_planes(_dev(M)), w, 3, update)
    assert np.array_equal(ref_flow, flow.cpu().numpy())
    if update:
        assert np.array_equal(ref_M, ofxcv.planes_to_hwc(Mo, w).cpu().numpy())
    # and the faithful (running-sum) evaluation of the same step agrees within tolerance
    f_flow, _ = oracle.update_flow_blur(R0, R1, M, 3, update, oracle.BLUR_FAITHFUL)
    err = np.abs(f_flow - flow.cpu().numpy())
    assert (err <= REL_TOL * np.maximum(1, np.abs(f_flow))).all()

@pytest.mark.parametrize("w,h", [(333, 257), (640, 480)])
def test_pyr_image_filter_contraction_bit_exact(oracle, ofxcv, w, h):
    """A third generation switch: OpenCV 4.x runs the separable filters and resize's vertical lerp through
    universal-intrinsics code whose taps are fused multiply-adds; the oracle (orc_set_filter_contraction) and the library
    (farneback.filter_contraction) follow together: every pyramid image bit-identical in every kernel family (dword / LDS-fused / two-pass),
    the whole call -- whose flow prolongation is a resize as well -- within 1e-4 at every sample, and the switch is not vacuous."""
    ga, gb = _gray_pair(oracle, w, h)
    levels = ofxcv.farneback_num_levels(w, h, 0.5, 3)
    ctx = ofxcv.Context(0)
    images = {}
    try:
        for fc in (0, 1):
            oracle.set_filter_contraction(fc)
            ctx.set_option("farneback.filter_contraction", fc)
            assert ctx.get_option("farneback.filter_contraction") == fc
            for fused in (1, 0):
                ctx.set_option("farneback.fused_pyramid", fused)
                for k in range(levels + 1):
                    lw, lh, sigma, ks = ofxcv.farneback_level_geom(w, h, 0.5, k)
                    ref = oracle.farneback_pyr_image(ga, lw, lh, sigma, ks)
                    got = ctx.farneback_pyr_image(_dev(ga), lw, lh, sigma, ks).cpu().numpy()
                    assert np.array_equal(ref, got), "contraction %d fused %d level %d: max diff %g" % (fc, fused, k, np.abs(ref - got).max())
                    images[(fc, k)] = ref
            ctx.set_option("farneback.fused_pyramid", 1)
            ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL)
            got = ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb)).cpu().numpy()
            err = np.abs(ref - got)
            assert (err <= REL_TOL * np.maximum(1, np.abs(ref))).all(), "contraction %d: max err %g" % (fc, err.max())
            print("filter contraction %d: whole call bit-identical to the oracle at %.6f of the samples" % (fc, (ref == got).mean()))
    finally:
        oracle.set_filter_contraction(0)
        ctx.close()
    changed = [k for k in range(levels + 1) if not np.array_equal(images[(0, k)], images[(1, k)])]
    assert changed, "the contracted filters gave the same images at every level"
    for k in changed:   # one rounding less per tap: last-bit differences only
        assert np.abs(images[(0, k)] - images[(1, k)]).max() <= 2e-5 * 255



def _check_flow(oracle, got, ga, gb, **kw):
    direct = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_DIRECT, **kw)
    faithful = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL, **kw)
    assert np.isfinite(got).all()
    # same evaluation order -> identical results
    assert np.array_equal(direct, got), "vs direct oracle: max diff %g" % np.abs(direct - got).max()
    err = np.abs(faithful - got)
    bad = err > REL_TOL * np.maximum(1, np.abs(faithful))
    h, w = got.shape[:2]
    assert bad.mean() <= outlier_bound(w, h), "vs faithful oracle: %.3g of samples outside 1e-4 (max err %g)" % (bad.mean(), err.max())
    return bad.mean(), err.max()


@pytest.mark.parametrize("w,h", [(64, 48), (160, 120), (333, 257), (640, 480)])
def test_farneback_end_to_end(oracle, ofxcv, direct_ctx, w, h):
    ga, gb = _gray_pair(oracle, w, h)
    got = direct_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb)).cpu().numpy()
    _check_flow(oracle, got, ga, gb)


def test_farneback_other_parameters(oracle, ofxcv, direct_ctx):
    ga, gb = _gray_pair(oracle, 200, 150, seed=99)
    kw = dict(levels=2, iterations=4, poly_n=7, poly_sigma=1.5)
    got = direct_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb), **kw).cpu().numpy()
    _check_flow(oracle, got, ga, gb, **kw)


def test_farneback_identical_frames(oracle, ofxcv, direct_ctx):
    """prev == next: the flow is exactly zero except where UpdateMatrices' out-of-range branch fires (the last
    row / column sample R1 at x1 == w-1 or y1 == h-1, optflowgf.cpp) and what the box window spreads from there."""
    ga, _ = _gray_pair(oracle, 160, 120)
    got = direct_ctx.calc_optical_flow_farneback(_dev(ga), _dev(ga)).cpu().numpy()
    assert np.array_equal(got, oracle.calc_optical_flow_farneback(ga, ga, blur_mode=oracle.BLUR_DIRECT))
    assert np.array_equal(got[:40, :40], np.zeros((40, 40, 2), np.float32))


def test_farneback_padded_strides(oracle, ofxcv, direct_ctx):
    import torch
    w, h = 150, 100
    ga, gb = _gray_pair(oracle, w, h)
    pa = torch.zeros((h, 256), dtype=torch.uint8, device="cuda")
    pb = torch.zeros((h, 192), dtype=torch.uint8, device="cuda")
    pa[:, :w] = _dev(ga)
    pb[:, :w] = _dev(gb)
    fl = torch.full((h, w + 10, 2), 7.0, dtype=torch.float32, device="cuda")
    direct_ctx.calc_optical_flow_farneback(pa[:, :w], pb[:, :w], fl[:, :w])
    ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_DIRECT)
    assert np.array_equal(ref, fl[:, :w].cpu().numpy())
    assert (fl[:, w:] == 7.0).all()


def test_farneback_1080p_properties(oracle, ofxcv, direct_ctx):
    """BASELINE config 3 size: full comparison against the oracle (about 5 s of CPU) + size-independent checks."""
    import torch
    from openfx_opencv_amd import synth
    w, h = 1920, 1080
    a, b = synth.flow_pair(w, h)
    ga = direct_ctx.to_byte_grayscale(_dev(a))
    gb = direct_ctx.to_byte_grayscale(_dev(b))
    assert np.array_equal(ga.cpu().numpy(), oracle.to_byte_grayscale(a))
    flow = direct_ctx.calc_optical_flow_farneback(ga, gb)
    again = direct_ctx.calc_optical_flow_farneback(ga, gb)
    assert torch.equal(flow, again)                      # deterministic, scratch reuse is clean
    got = flow.cpu().numpy()
    frac, mx = _check_flow(oracle, got, ga.cpu().numpy(), gb.cpu().numpy())
    u, v = synth.known_flow(w, h)
    inner = (slice(40, -40), slice(40, -40))
    assert np.abs(got[..., 0] - u)[inner].mean() < 0.3 and np.abs(got[..., 1] - v)[inner].mean() < 0.3
    zero = direct_ctx.calc_optical_flow_farneback(ga, ga)
    assert float(zero[:500, :900].abs().max()) == 0.0   # identical frames: zero far from the bottom/right border


def test_gray_lut_luma_weights_switch(oracle, ofxcv):
    """The luma weights of supportext's to_byte_grayscale_nodither cannot be verified here (empty submodule): Rec.709 (default) and
    Rec.601 are a switch of the oracle (orc_set_luma) and of the library (lut.luma), bit-exact against each other in every LUT kernel, and they do
    give different gray images."""
    rng = np.random.default_rng(8)
    imgs = [rng.uniform(-0.1, 1.2, size=(37, 53, 4)).astype(np.float32), rng.uniform(-0.1, 1.2, size=(40, 256 + 52, 4)).astype(np.float32),
            rng.uniform(0, 1, size=(33, 64, 3)).astype(np.float32)]
    c = ofxcv.Context(0)
    out = {}
    try:
        for std in (709, 601):
            oracle.set_luma(std)
            c.set_option("lut.luma", std)
            assert c.get_option("lut.luma") == std
            for i, im in enumerate(imgs):
                ref = oracle.to_byte_grayscale(im)
                assert np.array_equal(c.to_byte_grayscale(_dev(im)).cpu().numpy(), ref), (std, i)
                out[(std, i)] = ref
            wide = [imgs[1], imgs[1][::-1].copy()]
            import torch
            dst = [torch.empty((40, 256 + 52), dtype=torch.uint8, device="cuda") for _ in wide]
            c.to_byte_grayscale_batch([_dev(x) for x in wide], dst)
            assert np.array_equal(dst[0].cpu().numpy(), out[(std, 1)])
    finally:
        oracle.set_luma(709)
        c.close()
    assert not np.array_equal(out[(709, 0)], out[(601, 0)])
    with pytest.raises(ofxcv.OfxcvError):
        c2 = ofxcv.Context(0)
        try:
            c2.set_option("lut.luma", 2020)
        finally:
            c2.close()


def test_gray_lut_and_scatter(oracle, ofxcv, direct_ctx):
    import torch
    rng = np.random.default_rng(5)
    img = rng.uniform(-0.2, 1.3, size=(37, 53, 4)).astype(np.float32)
    img[0, 0] = (np.nan, 0, 0, 1)
    img[0, 1] = (np.inf, 1, 1, 1)
    img[0, 2] = (0, 0, 0, 1)
    assert np.array_equal(direct_ctx.to_byte_grayscale(_dev(img)).cpu().numpy(), oracle.to_byte_grayscale(img))
    rgb = np.ascontiguousarray(img[..., :3])
    rgb[0, 0] = 0.5
    rgb[0, 1] = 0.25
    assert np.array_equal(direct_ctx.to_byte_grayscale(_dev(rgb)).cpu().numpy(), oracle.to_byte_grayscale(rgb))
    # widths that are a multiple of four take the four-pixels-per-lane kernel (53 above does not): both against the oracle
    wide = rng.uniform(-0.2, 1.3, size=(37, 256 + 52, 4)).astype(np.float32)
    wide[3, 5] = (np.nan, np.inf, -np.inf, 0)
    assert np.array_equal(direct_ctx.to_byte_grayscale(_dev(wide)).cpu().numpy(), oracle.to_byte_grayscale(wide))
    # ... and the same pixels through the one-pixel-per-lane kernel (a view whose rows are not 16-byte aligned) give the same bytes
    shifted = _dev(np.concatenate([np.zeros((37, 1, 4), np.float32), wide], axis=1))[:, 1:]
    assert shifted.data_ptr() % 16 == 0 and np.array_equal(direct_ctx.to_byte_grayscale(shifted[:, :-1].contiguous()).cpu().numpy(), oracle.to_byte_grayscale(wide[:, :-1]))
    flow = rng.normal(0, 3, size=(37, 53, 2)).astype(np.float32)
    for mu, mv, rs in [(0b0011, 0b1100, (1.0, 1.0)), (0b0001, 0b0010, (0.5, 0.25)), (0b0101, 0b0100, (1.0, 2.0)), (0, 0, (1.0, 1.0))]:
        base = rng.normal(size=(37, 53, 4)).astype(np.float32)
        ref = oracle.flow_to_rgba(flow, base.copy(), [(mu >> c) & 1 for c in range(4)], [(mv >> c) & 1 for c in range(4)], *rs)
        got = direct_ctx.flow_to_rgba(_dev(flow), _dev(base), mu, mv, *rs).cpu().numpy()
        assert np.array_equal(ref, got)


def test_default_mode_is_opencv_order(oracle, ofxcv, gpu_ctx, strict_ctx):
    """a fresh context evaluates the window in the reference's order without any option being set"""
    ga, gb = _gray_pair(oracle, 160, 120)
    a = gpu_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb)).cpu().numpy()
    assert np.array_equal(a, strict_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb)).cpu().numpy())
    ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL)
    assert (np.abs(a - ref) <= REL_TOL * np.maximum(1, np.abs(ref))).all()


@pytest.fixture()
def strict_ctx(ofxcv):
    """a context in the OpenCV-order mode: the box window evaluated with the reference's running sums (every vertical row
    difference rounded to f32 before it enters the f64 column sum), strip-parallel"""
    ctx = ofxcv.Context(0)
    ctx.set_option("farneback.opencv_rounding", 1)
    yield ctx
    ctx.close()


def _strict_vs_faithful(oracle, ctx, w, h, **kw):
    ga, gb = _gray_pair(oracle, w, h)
    got = ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb), **kw).cpu().numpy()
    ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL, **kw)
    err = np.abs(ref - got)
    bad = err > REL_TOL * np.maximum(1, np.abs(ref))
    print("%dx%d OpenCV-order mode vs FAITHFUL oracle: max err %.3g, outside 1e-4: %d samples, bit-identical %.6f" % (w, h, err.max(), bad.sum(), (ref == got).mean()))
    # the contract: EVERY sample within the north_star tolerance
    assert not bad.any(), "max err %g, %d samples outside" % (err.max(), bad.sum())
    # only f64 additions of f32-origin terms are re-associated (strip carries, the 3-column sum): on these frames every
    # partial sum is exact, so the result is in fact identical to the sequential evaluation
    assert (ref == got).mean() > 0.999
    return got


@pytest.mark.parametrize("w,h", [(64, 48), (160, 120), (333, 257), (640, 480), (62, 40), (63, 33), (125, 70)])
def test_opencv_order_mode_matches_faithful_oracle_everywhere(oracle, strict_ctx, w, h):
    """With the reference's f32 rounding of the vertical running sums reproduced, EVERY sample is within the north_star
    tolerance of the faithful oracle -- the outliers of the default path are that rounding noise and nothing else.
    62/63/125 columns: one wavefront owns 62 columns, so these sit on the tile edges."""
    _strict_vs_faithful(oracle, strict_ctx, w, h)


# the three evaluations of OpenCV's running sums the library has: mode 2 (serial column scan, the independent cross-check), the
# overlapped-strip form and the column-owning form (farneback.col_min 1 forces it on every level of any frame)
_FORMS = (dict(opencv_rounding=2), dict(col=0), dict(col_min=1), dict(col_min=1, col_ring=0))
# strip / wavefront geometries of the overlapped-strip form the library otherwise picks by level size
# farneback.halo_geom (test hook): low nibble 1 small / 2 four tall / 3 eight tall wavefronts, bits 4..6 the small form's wavefronts, bits 8.. rows per tall strip
_HG = lambda form=0, small=0, strip=0: dict(halo_geom=form | (small << 4) | (strip << 8))
_HALO_GEOMS = (_HG(1), _HG(2), _HG(3), _HG(2, strip=33), _HG(2, strip=35), _HG(3, strip=65), _HG(3, strip=67), _HG(3, strip=70), _HG(3, strip=72),
               _HG(1, small=2), _HG(1, small=4), _HG(1, small=5), _HG(1, small=6))


def _flow_with(ofxcv, opts, ga, gb, *args, twice=False, **kw):
    ctx = ofxcv.Context(0)
    for k, v in opts.items():
        ctx.set_option("farneback." + k, v)
    for _ in range(2 if twice else 1):   # twice: the second call replays the captured graph
        a = tuple(_dev(x.copy()) for x in args)
        got = ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb), *a, **kw).cpu().numpy()
    assert ctx.get_option("farneback.col_aborts") == 0
    ctx.close()
    return got


@pytest.mark.parametrize("w,h", [(125, 70), (333, 257), (640, 480), (1920, 1080)])
def test_opencv_order_mode_forms_agree(oracle, ofxcv, w, h):
    """OpenCV's running sums as a serial column scan (mode 2), strip-parallel with overlapped strips (one launch per iteration, every
    strip / wavefront geometry) and with column-owning workgroups (two steps per launch, both row geometries): the same flow bit for
    bit, within tolerance of the faithful oracle at every sample; no bounded wait of the column-owning kernel ever ran out"""
    ga, gb = _gray_pair(oracle, w, h)
    ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL)
    outs = [_flow_with(ofxcv, opts, ga, gb, twice=True) for opts in _FORMS]
    assert (np.abs(outs[0] - ref) <= REL_TOL * np.maximum(1, np.abs(ref))).all()
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    for opts in _HALO_GEOMS:
        assert np.array_equal(_flow_with(ofxcv, dict(col=0, **opts), ga, gb), outs[0]), opts


@pytest.mark.parametrize("w,h,levels", [(333, 257, 3), (160, 120, 0), (640, 480, 2), (61, 131, 0), (60, 64, 1), (59, 300, 1)])
def test_opencv_order_mode_step_pairs_and_first_matrix_forms(oracle, ofxcv, w, h, levels):
    """The steps of a level -- first M (zero flow on the coarsest level, the prolongated coarser flow below it, the caller's flow with
    USE_INITIAL_FLOW), iterations - 1 x iterate, last -- in every pairing the column-owning form has: (first, last) for one iteration,
    (first, iterate) + (last, -) for two, (first, iterate) (iterate, last) for three, an (iterate, iterate) between for four and five;
    widths of one tile column minus / plus one, heights around a round of 32 rows.  All forms bit for bit, with and without an initial flow."""
    ga, gb = _gray_pair(oracle, w, h)
    rng = np.random.default_rng(5)
    init = rng.normal(0, 2, size=(h, w, 2)).astype(np.float32)
    for kw in (dict(iterations=1), dict(iterations=2), dict(iterations=3), dict(iterations=4, flags=ofxcv.OPTFLOW_USE_INITIAL_FLOW),
               dict(iterations=5), dict(iterations=2, flags=ofxcv.OPTFLOW_USE_INITIAL_FLOW),
               dict(iterations=1, flags=ofxcv.OPTFLOW_USE_INITIAL_FLOW)):   # at level 0 the given flow and the result share a buffer (fuzz_halo.py, round 4)
        args = (init,) if "flags" in kw else ()
        outs = [_flow_with(ofxcv, opts, ga, gb, *args, levels=levels, **kw) for opts in _FORMS + (dict(col=0, halo_geom=2), dict(col=0, **_HG(1, small=5)))]
        assert all(np.array_equal(outs[0], o) for o in outs[1:]), kw
    ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL, levels=levels, iterations=3)
    got = _flow_with(ofxcv, dict(col_min=1), ga, gb, levels=levels, iterations=3)
    assert (np.abs(got - ref) <= REL_TOL * np.maximum(1, np.abs(ref))).all()


@pytest.mark.parametrize("w,h,n", [(640, 480, 2), (125, 70, 3), (1920, 1080, 8)])
def test_column_owning_form_batches(oracle, ofxcv, w, h, n):
    """batches through the column-owning form (one workgroup per tile column and pair): forced on every level (col_min 1) and, for the
    8 x 1080p batch of the benchmark, as the library picks it (level 0 only) -- equal to single calls through the overlapped strips bit
    for bit, with an initial flow as well; pair 0 within tolerance of the faithful oracle at every sample"""
    prs = _pairs(oracle, w, h, range(7, 7 + n))
    da, db = [_dev(a) for a, _ in prs], [_dev(b) for _, b in prs]
    rng = np.random.default_rng(9)
    inits = [rng.normal(0, 2, size=(h, w, 2)).astype(np.float32) for _ in prs]
    for kw in (dict(), dict(iterations=2, levels=2, flags=ofxcv.OPTFLOW_USE_INITIAL_FLOW)):
        ref_ctx = ofxcv.Context(0)
        ref_ctx.set_option("farneback.col", 0)
        singles = []
        for z in range(n):
            a = (_dev(inits[z].copy()),) if "flags" in kw else ()
            singles.append(ref_ctx.calc_optical_flow_farneback(da[z], db[z], *a, **kw).cpu().numpy())
        ref_ctx.close()
        for opts in (dict(col_min=1), dict()):
            ctx = ofxcv.Context(0)
            for k, v in opts.items():
                ctx.set_option("farneback." + k, v)
            for _ in range(2):
                fl = [_dev(i0.copy()) for i0 in inits] if "flags" in kw else None
                got = [f.cpu().numpy() for f in ctx.calc_optical_flow_farneback_batch(da, db, fl, **kw)]
            assert ctx.get_option("farneback.col_aborts") == 0
            ctx.close()
            for z in range(n):
                assert np.array_equal(singles[z], got[z]), (kw, opts, z)
    ref = oracle.calc_optical_flow_farneback(prs[0][0], prs[0][1], blur_mode=oracle.BLUR_FAITHFUL)
    ctx = ofxcv.Context(0)
    ctx.set_option("farneback.col_min", 1)
    got = ctx.calc_optical_flow_farneback(da[0], db[0]).cpu().numpy()
    ctx.close()
    assert (np.abs(got - ref) <= REL_TOL * np.maximum(1, np.abs(ref))).all()


@pytest.mark.parametrize("w,h", [(1, 1), (2, 3), (3, 2), (7, 5), (17, 4), (64, 1), (1, 64), (130, 9), (9, 130), (63, 22), (33, 24)])
def test_opencv_order_mode_tiny_frames(oracle, ofxcv, w, h):
    """frames smaller than a strip, a tile, the three-row reach of a row difference: every form gives the same flow as
    the faithful oracle's (levels clip to 0 below 32 pixels), scratch reservations hold (column-sum scratch of
    one-row images)"""
    rng = np.random.default_rng(w * 131 + h)
    ga = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    gb = np.roll(ga, 1, axis=1) if w > 1 else ga.copy()
    ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL, iterations=3)
    outs = []
    for opts in _FORMS[:3]:   # frames below 64 rows stay with the overlapped strips even when the column-owning form is forced
        got = _flow_with(ofxcv, opts, ga, gb, twice=True, iterations=3)
        assert (np.abs(got - ref) <= REL_TOL * np.maximum(1, np.abs(ref))).all(), (opts, np.abs(got - ref).max())
        outs.append(got)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_opencv_order_mode_single_step(oracle, ofxcv, strict_ctx):
    rng = np.random.default_rng(13)
    h, w = 119, 161
    R0 = rng.normal(0, 20, size=(h, w, 5)).astype(np.float32)
    R1 = rng.normal(0, 20, size=(h, w, 5)).astype(np.float32)
    M = oracle.update_matrices(R0, R1, rng.normal(0, 1, size=(h, w, 2)).astype(np.float32))
    for update in (True, False):
        ref_flow, ref_M = oracle.update_flow_blur(R0, R1, M, 3, update, oracle.BLUR_FAITHFUL)
        flow, Mo = strict_ctx.farneback_update_flow_blur(ofxcv.hwc_to_planes(_dev(R0)), ofxcv.hwc_to_planes(_dev(R1)), ofxcv.hwc_to_planes(_dev(M)), w, 3, update)
        err = np.abs(ref_flow - flow.cpu().numpy())
        assert (err <= 1e-6 * np.maximum(1, np.abs(ref_flow))).all(), err.max()     # only f64 association differs
        if update:
            errM = np.abs(ref_M - ofxcv.planes_to_hwc(Mo, w).cpu().numpy())
            assert (errM <= 1e-5 * np.maximum(1, np.abs(ref_M))).all(), errM.max()


def test_opencv_order_mode_wide_dynamic_range(oracle, strict_ctx):
    """Flat (exactly constant), barely textured (one or two gray levels) and full-contrast regions side by side: the entries
    of M then span many orders of magnitude inside one column, the f64 column sums are no longer exact, and the strip-parallel
    evaluation really does re-associate them -- the result must still be within tolerance of the sequential oracle at every
    sample (in practice it stays bit-identical almost everywhere)."""
    from openfx_opencv_amd import synth
    w, h = 640, 480
    a, b = synth.flow_pair(w, h, seed=21)
    ga, gb = oracle.to_byte_grayscale(a).astype(np.int32), oracle.to_byte_grayscale(b).astype(np.int32)
    for g in (ga, gb):
        g[:, : w // 4] = 128                                        # flat
        g[:, w // 4 : w // 2] = 128 + (g[:, w // 4 : w // 2] >> 7)  # one gray level of texture
        g[: h // 3, w // 2 :] = 64 + (g[: h // 3, w // 2 :] >> 6)   # a few levels
    ga, gb = ga.astype(np.uint8), gb.astype(np.uint8)
    got = strict_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb)).cpu().numpy()
    ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL)
    err = np.abs(ref - got)
    bad = err > REL_TOL * np.maximum(1, np.abs(ref))
    print("wide dynamic range: max err %.3g, outside %d, bit-identical %.6f" % (err.max(), bad.sum(), (ref == got).mean()))
    assert not bad.any(), "max err %g, %d samples outside" % (err.max(), bad.sum())


def test_column_ring_window_and_fallback(ofxcv):
    """The R1 ring of the column-owning form holds the rows and columns within +- 4 pixels of a pixel; a wavefront-row with a sample beyond that takes
    the global gather.  Small motion (every gather from the ring), a 9 x 7 pixel shift (every gather from memory), a shift at the window's
    edge (rows of both kinds) and a caller's noisy initial flow: ring on / off / overlapped strips agree bit for bit."""
    import torch
    rng = np.random.default_rng(7)
    w, h = 700, 330
    base = rng.integers(0, 256, size=(h + 40, w + 40), dtype=np.uint8)
    blur = ((base[:-2, :-2].astype(np.int32) + base[1:-1, 1:-1] + base[2:, 2:]) // 3).astype(np.uint8)
    crop = lambda dy, dx: _dev(np.ascontiguousarray(blur[20 + dy:20 + dy + h, 20 + dx:20 + dx + w]))
    for name, (dy, dx), kw in (("small", (1, 1), dict()), ("large", (7, -9), dict()), ("edge", (4, -4), dict(iterations=5)),
                               ("initial flow", (2, 1), dict(iterations=3, flags=ofxcv.OPTFLOW_USE_INITIAL_FLOW))):
        pa, pb = [crop(0, 0), crop(1, 0)], [crop(dy, dx), crop(dy + 1, dx)]
        init = [rng.normal(0, 3, size=(h, w, 2)).astype(np.float32) for _ in range(2)]
        outs = []
        for opts in (dict(col=0), dict(col_min=1), dict(col_min=1, col_ring=0)):
            c = ofxcv.Context(0)
            for k, v in opts.items():
                c.set_option("farneback." + k, v)
            fl = [torch.from_numpy(i0.copy()).cuda() for i0 in init] if "flags" in kw else None
            outs.append([f.cpu().numpy() for f in c.calc_optical_flow_farneback_batch(pa, pb, fl, **kw)])
            assert c.get_option("farneback.col_aborts") == 0
            c.close()
        for o in outs[1:]:
            for z in range(2):
                assert np.array_equal(outs[0][z], o[z]), (name, z)


def test_column_forms_on_flat_bands_above_texture(oracle, ofxcv):
    """the column-owning form hands `P + (d0 + d1 + d2 + d3)` to the next wavefront where OpenCV adds row by row; the two are the
    same doubles only while no f64 addition rounds.  Frames with exactly flat and one-gray-level BANDS above full-contrast texture put values
    many orders of magnitude apart into ONE column sum, where the additions do round: every form (serial scan, overlapped strips, column-owning
    with the R1 ring and without, both row geometries) must still be within 1e-4 of the sequential oracle at EVERY sample; how many samples stay
    bit-identical to the serial scan is printed (the forms re-associate f64 additions, nothing else)."""
    from openfx_opencv_amd import synth
    w, h = 333, 257
    a, b = synth.flow_pair(w, h, seed=33)
    ga, gb = oracle.to_byte_grayscale(a).astype(np.int32), oracle.to_byte_grayscale(b).astype(np.int32)
    for g in (ga, gb):
        g[: h // 4, :] = 128                                        # flat band
        g[h // 4 : h // 2, :] = 128 + (g[h // 4 : h // 2, :] >> 7)  # one gray level of texture
        g[h // 2 : h // 2 + 8, :] = 0                               # a black bar: zero matrices in the middle of every column
    ga, gb = ga.astype(np.uint8), gb.astype(np.uint8)
    ref = oracle.calc_optical_flow_farneback(ga, gb, iterations=6, blur_mode=oracle.BLUR_FAITHFUL)
    da, db = _dev(ga), _dev(gb)
    scan = None
    for opts in _FORMS:
        c = ofxcv.Context(0)
        for k, v in opts.items():
            c.set_option("farneback." + k, v)
        got = c.calc_optical_flow_farneback_batch([da, da], [db, db], iterations=6)[1].cpu().numpy()
        assert c.get_option("farneback.col_aborts") == 0
        c.close()
        err = np.abs(ref - got)
        bad = err > REL_TOL * np.maximum(1, np.abs(ref))
        if scan is None:
            scan = got
        print("flat bands, %-40s max err %.3g, outside %d, bit-identical to the oracle %.6f, to the serial scan %.6f"
              % (opts, err.max(), bad.sum(), (ref == got).mean(), (scan == got).mean()))
        assert not bad.any(), (opts, err.max(), bad.sum())


def test_opencv_order_mode_other_parameters(oracle, strict_ctx):
    _strict_vs_faithful(oracle, strict_ctx, 217, 163, levels=2, iterations=4, poly_n=7, poly_sigma=1.5)
    _strict_vs_faithful(oracle, strict_ctx, 217, 163, levels=3, iterations=1)
    _strict_vs_faithful(oracle, strict_ctx, 200, 150, pyr_scale=0.8, levels=4, iterations=3)


def test_opencv_order_mode_1080p(oracle, ofxcv, strict_ctx):
    """BASELINE config 3 size, every sample"""
    _strict_vs_faithful(oracle, strict_ctx, 1920, 1080)


def test_farneback_4k_both_modes(oracle, ofxcv, direct_ctx, strict_ctx):
    """BASELINE config 5 workload (one 3840x2160 pair): default path bit-identical to the DIRECT oracle and within the
    outlier bound of the FAITHFUL one; OpenCV-order mode within tolerance at every sample of the FAITHFUL oracle."""
    w, h = 3840, 2160
    ga, gb = _gray_pair(oracle, w, h)
    got = direct_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb)).cpu().numpy()
    frac, mx = _check_flow(oracle, got, ga, gb)
    print("3840x2160 default path vs FAITHFUL: outside-1e-4 fraction %.3g, max err %.3g" % (frac, mx))
    _strict_vs_faithful(oracle, strict_ctx, w, h)


def test_farneback_4k_batch_of_eight(oracle, ofxcv):
    """BASELINE configs[4] in its batched form: 8 DIFFERENT 3840x2160 pairs (one GPU's share of the 64) through
    ofxcv_calc_optical_flow_farneback_batch_rgba equal 8 single calls bit for bit (flows and RGBA images), and pair 0 is within
    1e-4 of the FAITHFUL oracle at every sample.  Twice: the second call replays the captured graph."""
    w, h, n = 3840, 2160, 8
    prs = _pairs(oracle, w, h, range(1234, 1234 + n))
    da, db = [_dev(a) for a, _ in prs], [_dev(b) for _, b in prs]
    import torch
    single = ofxcv.Context(0)
    singles = []
    for z in range(n):
        f = single.calc_optical_flow_farneback(da[z], db[z])
        singles.append(f.cpu().numpy())
    batch = ofxcv.Context(0)
    for _ in range(2):
        dsts = [torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(n)]
        flows = batch.calc_optical_flow_farneback_batch_rgba(da, db, None, dsts, [0b0101] * n, [0b1010] * n)
        for z in range(n):
            got = flows[z].cpu().numpy()
            assert np.array_equal(got, singles[z]), "pair %d: max diff %g" % (z, np.abs(got - singles[z]).max())
            img = dsts[z].cpu().numpy()
            assert np.array_equal(img[..., 0], got[..., 0]) and np.array_equal(img[..., 1], got[..., 1])
            assert np.array_equal(img[..., 2], got[..., 0]) and np.array_equal(img[..., 3], got[..., 1])
        del dsts
    ref = oracle.calc_optical_flow_farneback(prs[0][0], prs[0][1], blur_mode=oracle.BLUR_FAITHFUL)
    err = np.abs(singles[0] - ref)
    bad = err > REL_TOL * np.maximum(1, np.abs(ref))
    print("3840x2160 batch of 8, pair 0 vs FAITHFUL oracle: max err %.3g, outside 1e-4: %d, bit-identical %.6f" % (err.max(), bad.sum(), (ref == singles[0]).mean()))
    assert not bad.any()
    single.close()
    batch.close()


# ---- OPTFLOW_FARNEBACK_GAUSSIAN / OPTFLOW_USE_INITIAL_FLOW ----

@pytest.mark.parametrize("w,h,winsize", [(64, 48, 3), (160, 120, 5), (333, 257, 7), (320, 240, 9)])
def test_gaussian_window_bit_exact(oracle, ofxcv, direct_ctx, w, h, winsize):
    """Both passes of the Gaussian window accumulate in f32 in the reference's order: identical to the oracle."""
    ga, gb = _gray_pair(oracle, w, h)
    ref = oracle.calc_optical_flow_farneback(ga, gb, winsize=winsize, flags=oracle.OPTFLOW_FARNEBACK_GAUSSIAN)
    got = direct_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb), winsize=winsize, flags=ofxcv.OPTFLOW_FARNEBACK_GAUSSIAN)
    assert np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("w,h,levels", [(320, 240, 3), (333, 257, 3), (160, 120, 0), (640, 480, 2)])
def test_initial_flow_bit_exact(oracle, ofxcv, direct_ctx, w, h, levels):
    """USE_INITIAL_FLOW: INTER_AREA resize of the caller's flow to the top level (integer factor 320->40, fractional
    333->42, same size for levels=0), then the usual walk; compared with the oracle's DIRECT evaluation."""
    ga, gb = _gray_pair(oracle, w, h)
    rng = np.random.default_rng(3)
    init = (rng.standard_normal((h, w, 2)) * 2).astype(np.float32)
    ref = oracle.calc_optical_flow_farneback(ga, gb, levels=levels, iterations=4, flags=oracle.OPTFLOW_USE_INITIAL_FLOW,
                                             initial_flow=init, blur_mode=oracle.BLUR_DIRECT)
    flow = _dev(init.copy())
    got = direct_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb), flow=flow, levels=levels, iterations=4,
                                              flags=ofxcv.OPTFLOW_USE_INITIAL_FLOW)
    assert got.data_ptr() == flow.data_ptr()
    assert np.array_equal(got.cpu().numpy(), ref)


def test_both_flags_and_rejected_flags(oracle, ofxcv, direct_ctx):
    w, h = 200, 150
    ga, gb = _gray_pair(oracle, w, h)
    init = np.full((h, w, 2), 0.5, np.float32)
    fl = oracle.OPTFLOW_USE_INITIAL_FLOW | oracle.OPTFLOW_FARNEBACK_GAUSSIAN
    ref = oracle.calc_optical_flow_farneback(ga, gb, winsize=5, iterations=3, flags=fl, initial_flow=init)
    got = direct_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb), flow=_dev(init.copy()), winsize=5, iterations=3, flags=fl)
    assert np.array_equal(got.cpu().numpy(), ref)
    with pytest.raises(ofxcv.OfxcvError) as e:
        direct_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb), flags=1)
    assert e.value.status == -4  # OFXCV_ERR_UNSUPPORTED


@pytest.mark.parametrize("kw", [
    dict(pyr_scale=0.8, levels=4, iterations=3),          # non-dyadic pyramid: generic blur taps, fractional resize
    dict(pyr_scale=0.6, levels=2, iterations=2, winsize=5),
    dict(levels=1, iterations=1),                          # a single iteration: no updating pass at all
    dict(levels=3, iterations=2),                          # one updating + the final pass (no fused pair)
    dict(levels=3, iterations=6, winsize=7, poly_n=3, poly_sigma=0.9),
    dict(levels=0, iterations=5),                          # no pyramid
])
def test_farneback_parameter_grid_bit_exact(oracle, ofxcv, direct_ctx, kw):
    ga, gb = _gray_pair(oracle, 217, 163, seed=7)
    got = direct_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb), **kw).cpu().numpy()
    ref = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_DIRECT, **kw)
    assert np.array_equal(got, ref)


def test_farneback_unaligned_sources(oracle, ofxcv, direct_ctx):
    """Source pointers / strides that are not multiples of 4: the pyramid kernels must take their byte path."""
    import torch
    w, h = 320, 240   # level 1 is exactly half: the dword kernel would be eligible with aligned sources
    ga, gb = _gray_pair(oracle, w, h)
    buf_a = torch.zeros((h * 323 + 8,), dtype=torch.uint8, device="cuda")
    buf_b = torch.zeros((h * 321 + 8,), dtype=torch.uint8, device="cuda")
    pa = buf_a[1:1 + h * 323].view(h, 323)[:, :w]
    pb = buf_b[3:3 + h * 321].view(h, 321)[:, :w]
    pa.copy_(_dev(ga))
    pb.copy_(_dev(gb))
    got = direct_ctx.calc_optical_flow_farneback(pa, pb).cpu().numpy()
    assert np.array_equal(got, oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_DIRECT))


# ---- batched calls (ofxcv_calc_optical_flow_farneback_batch: BASELINE configs[4] batches pairs per GPU; a default
# VectorGenerator output frame is a batch of two, VectorGenerator.cpp:597-638) ----

def _pairs(oracle, w, h, seeds):
    return [_gray_pair(oracle, w, h, seed) for seed in seeds]


@pytest.mark.parametrize("w,h,n", [(64, 48, 3), (333, 257, 3), (640, 480, 5), (125, 70, 16), (1920, 1080, 3)])
def test_batch_equals_single_calls_bit_for_bit(oracle, ofxcv, w, h, n):
    """n DIFFERENT pairs in one batched call (every launch of the level walk carries all of them in its grid's z dimension)
    give, bit for bit, the flows of n single calls -- and the first pair is within tolerance of the faithful oracle at every
    sample.  Twice: the captured graph is replayed the second time."""
    prs = _pairs(oracle, w, h, range(100, 100 + n))
    single = ofxcv.Context(0)
    batch = ofxcv.Context(0)
    singles = [single.calc_optical_flow_farneback(_dev(a), _dev(b)).cpu().numpy() for a, b in prs]
    da, db = [_dev(a) for a, _ in prs], [_dev(b) for _, b in prs]
    for _ in range(2):
        flows = batch.calc_optical_flow_farneback_batch(da, db)
        got = [f.cpu().numpy() for f in flows]
        for z in range(n):
            assert np.array_equal(got[z], singles[z]), "pair %d of %d: max diff %g" % (z, n, np.abs(got[z] - singles[z]).max())
    ref = oracle.calc_optical_flow_farneback(prs[0][0], prs[0][1], blur_mode=oracle.BLUR_FAITHFUL)
    assert (np.abs(got[0] - ref) <= REL_TOL * np.maximum(1, np.abs(ref))).all()
    single.close()
    batch.close()


def test_batch_shared_first_frame_and_every_window_mode(oracle, ofxcv):
    """forward + backward flow of one reference frame (two pairs that share their first image) as a batch, in every
    evaluation mode of the window (OpenCV order with overlapped strips / column-owning workgroups, serial scan, direct sums, other window size,
    Gaussian window): identical to single calls"""
    w, h = 333, 257
    (a, b), (_, c) = _pairs(oracle, w, h, (5, 6))
    da, db, dc = _dev(a), _dev(b), _dev(c)
    cases = [dict(opts=dict(opencv_rounding=1)), dict(opts=dict(opencv_rounding=1, col_min=1)), dict(opts=dict(opencv_rounding=1, col_min=1, col_ring=0)),
             dict(opts=dict(opencv_rounding=2)), dict(opts=dict(opencv_rounding=0)), dict(opts=dict(opencv_rounding=0), kw=dict(iterations=4)),
             dict(opts=dict(opencv_rounding=1), kw=dict(winsize=5)), dict(opts=dict(opencv_rounding=1), kw=dict(flags=ofxcv.OPTFLOW_FARNEBACK_GAUSSIAN, winsize=5)),
             dict(opts=dict(opencv_rounding=1, halo_geom=2)), dict(opts=dict(opencv_rounding=1, **_HG(1, small=5)))]
    for case in cases:
        ctx = ofxcv.Context(0)
        for k, v in case["opts"].items():
            ctx.set_option("farneback." + k, v)
        kw = case.get("kw", {})
        s0 = ctx.calc_optical_flow_farneback(da, db, **kw).cpu().numpy()
        s1 = ctx.calc_optical_flow_farneback(da, dc, **kw).cpu().numpy()
        f0, f1 = ctx.calc_optical_flow_farneback_batch([da, da], [db, dc], **kw)
        assert np.array_equal(f0.cpu().numpy(), s0) and np.array_equal(f1.cpu().numpy(), s1), case
        ctx.close()


@pytest.mark.parametrize("mb", [1, 20, 100000])
def test_batch_launch_groups(oracle, ofxcv, mb):
    """a level whose working set for the whole batch exceeds the Infinity Cache budget (option farneback.batch_mb) is
    walked in groups of pairs: groups of one (1 MiB), of two with a remainder (20 MiB), everything in one launch"""
    w, h, n = 333, 257, 5
    prs = _pairs(oracle, w, h, range(40, 40 + n))
    ctx = ofxcv.Context(0)
    singles = [ctx.calc_optical_flow_farneback(_dev(a), _dev(b)).cpu().numpy() for a, b in prs]
    ctx.set_option("farneback.batch_mb", mb)
    for geom in (0, 2, 3):
        ctx.set_option("farneback.halo_geom", geom)
        flows = ctx.calc_optical_flow_farneback_batch([_dev(a) for a, _ in prs], [_dev(b) for _, b in prs])
        for f, s in zip(flows, singles):
            assert np.array_equal(f.cpu().numpy(), s)
    ctx.close()


@pytest.mark.parametrize("opts", [dict(), dict(col_min=1), dict(col_min=1, col_ring=0), dict(opencv_rounding=0), dict(opencv_rounding=2)])
def test_flow_to_rgba_fused_into_the_call(oracle, ofxcv, opts):
    """ofxcv_calc_optical_flow_farneback_batch_rgba: F7 rides on the launch that produces the final flow (overlapped strips: the default for these
    small batches; column-owning form: col_min 1) or is appended by the library (other window modes) -- the RGBA images equal ofxcv_flow_to_rgba applied to the returned flows bit for bit: all
    four channels mapped, two, one, none; render scales; a pair without an image; forward + backward flow of one output frame
    written into ONE image with disjoint channels; unmapped channels keep their content.  Twice (graph replay)."""
    w, h = 333, 257
    prs = _pairs(oracle, w, h, (21, 22, 23, 24))
    da, db = [_dev(a) for a, _ in prs], [_dev(b) for _, b in prs]
    ctx = ofxcv.Context(0)
    for k, v in opts.items():
        ctx.set_option("farneback." + k, v)
    rng = np.random.default_rng(2)
    base = [rng.normal(size=(h, w, 4)).astype(np.float32) for _ in range(4)]
    mus, mvs = [0b0001, 0b0101, 0b0100, 0b0000], [0b0010, 0b1010, 0b0000, 0b0000]
    for rs in ((1.0, 1.0), (0.5, 0.25)):
        for _ in range(2):
            dsts = [_dev(b.copy()) for b in base]
            dsts[3] = None
            flows = ctx.calc_optical_flow_farneback_batch_rgba(da, db, None, dsts, mus, mvs, *rs)
            for z in range(3):
                ref = ctx.flow_to_rgba(flows[z], _dev(base[z].copy()), mus[z], mvs[z], *rs).cpu().numpy()
                assert np.array_equal(dsts[z].cpu().numpy(), ref), (opts, rs, z)
    # one output frame: forward flow -> R,G, backward flow -> B,A of the same image
    shared = _dev(base[0].copy())
    flows = ctx.calc_optical_flow_farneback_batch_rgba([da[0], da[0]], [db[0], db[1]], None, [shared, shared], [0b0001, 0b0100], [0b0010, 0b1000])
    ref = ctx.flow_to_rgba(flows[0], _dev(base[0].copy()), 0b0001, 0b0010)
    ref = ctx.flow_to_rgba(flows[1], ref, 0b0100, 0b1000).cpu().numpy()
    assert np.array_equal(shared.cpu().numpy(), ref)
    plain = ctx.calc_optical_flow_farneback_batch([da[0], da[0]], [db[0], db[1]])
    assert all(np.array_equal(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(flows, plain))
    ctx.close()


def test_batch_initial_flow_and_argument_checks(oracle, ofxcv):
    w, h = 320, 240
    prs = _pairs(oracle, w, h, (1, 2, 3))
    rng = np.random.default_rng(3)
    inits = [rng.normal(0, 2, size=(h, w, 2)).astype(np.float32) for _ in prs]
    ctx = ofxcv.Context(0)
    singles = [ctx.calc_optical_flow_farneback(_dev(a), _dev(b), _dev(i0), flags=ofxcv.OPTFLOW_USE_INITIAL_FLOW).cpu().numpy() for (a, b), i0 in zip(prs, inits)]
    flows = ctx.calc_optical_flow_farneback_batch([_dev(a) for a, _ in prs], [_dev(b) for _, b in prs], [_dev(i0) for i0 in inits],
                                                  flags=ofxcv.OPTFLOW_USE_INITIAL_FLOW)
    for f, s in zip(flows, singles):
        assert np.array_equal(f.cpu().numpy(), s)
    with pytest.raises(ofxcv.OfxcvError):   # more pairs than the pointer tables hold
        ctx.calc_optical_flow_farneback_batch([_dev(prs[0][0])] * 17, [_dev(prs[0][1])] * 17)
    ctx.close()


def test_gray_lut_batch_equals_single_calls(gpu_ctx):
    """ofxcv_to_byte_grayscale_batch: the frames of a batched call in one launch -- the same bytes as one call per frame; RGB
    frames, odd widths and unaligned views take the per-frame path behind the same entry point."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(5)
    for (h, w, nc, n) in ((120, 256, 4, 16), (67, 130, 4, 3), (40, 96, 3, 4), (33, 125, 4, 2), (64, 64, 4, 32)):
        srcs = [(torch.rand((h, w, nc), generator=g) * 1.4 - 0.2).cuda() for _ in range(n)]
        srcs[0][0, 0, 0] = float("nan")
        srcs[-1][1, 1, :3] = float("inf")
        want = [gpu_ctx.to_byte_grayscale(s) for s in srcs]
        outs = [torch.full((h, w), 7, dtype=torch.uint8, device="cuda") for _ in range(n)]
        gpu_ctx.to_byte_grayscale_batch(srcs, outs)
        torch.cuda.synchronize()
        for a, b in zip(want, outs):
            assert torch.equal(a, b), (h, w, nc, n)


def test_deep_pyramid_and_two_pass_images_in_batches(oracle, ofxcv):
    """The pyramid images of levels beyond 1/8 (levels = 5 at 1024x576: 39 and 79 blur taps) come from the two-pass kernels
    (row filter into a buffer, column filter + resize), which take as many frames of a call per launch as the buffer holds; option
    farneback.fused_pyramid 0 sends every level that way.  A batch equals its single calls bit for bit either way, the two-pass
    images equal the fused ones, and pair 0 is within tolerance of the faithful oracle."""
    w, h, n = 1024, 576, 5
    prs = _pairs(oracle, w, h, range(300, 300 + n))
    da, db = [_dev(a) for a, _ in prs], [_dev(b) for _, b in prs]
    kw = dict(levels=5, iterations=2)
    single = ofxcv.Context(0)
    singles = [single.calc_optical_flow_farneback(x, y, **kw).cpu().numpy() for x, y in zip(da, db)]
    for fused in (1, 0):
        c = ofxcv.Context(0)
        c.set_option("farneback.fused_pyramid", fused)
        got = [f.cpu().numpy() for f in c.calc_optical_flow_farneback_batch(da, db, **kw)]
        for z in range(n):
            assert np.array_equal(got[z], singles[z]), (fused, z)
        c.close()
    ref = oracle.calc_optical_flow_farneback(prs[0][0], prs[0][1], blur_mode=oracle.BLUR_FAITHFUL, **kw)
    assert (np.abs(singles[0] - ref) <= REL_TOL * np.maximum(1, np.abs(ref))).all()
    single.close()


def test_column_owning_form_for_part_of_a_batch(oracle, ofxcv):
    """One workgroup per tile column and pair, one workgroup per CU: the pairs of a call that would start another round of the
    chip keep the overlapped strips (a 1921-pixel-wide frame has 33 tile columns: 7 pairs fill the 256 CUs,
    the eighth does not).  With the column-owning form forced from 3 workgroups on a small frame (6 tile columns) the split is
    driven through the same plan: whatever mix of forms a batch is walked in, every pair equals its single call bit for bit."""
    w, h, n = 333, 257, 7
    prs = _pairs(oracle, w, h, range(400, 400 + n))
    da, db = [_dev(a) for a, _ in prs], [_dev(b) for _, b in prs]
    single = ofxcv.Context(0)
    singles = [single.calc_optical_flow_farneback(x, y, iterations=4).cpu().numpy() for x, y in zip(da, db)]
    single.close()
    for opts in (dict(col_min=1), dict(col_min=12), dict(col_min=30), dict(col_min=36), dict(col=0)):
        c = ofxcv.Context(0)
        for k, v in opts.items():
            c.set_option("farneback." + k, v)
        got = [f.cpu().numpy() for f in c.calc_optical_flow_farneback_batch(da, db, iterations=4)]
        for z in range(n):
            assert np.array_equal(got[z], singles[z]), (opts, z)
        assert c.get_option("farneback.col_aborts") == 0
        c.close()
    # the real thing: 33 tile columns x 8 pairs (7 in the column-owning form + 1 in strips), pairs 0 and 7 against single calls
    w, h, n = 1921, 360, 8
    prs = _pairs(oracle, w, h, range(500, 500 + n))
    da, db = [_dev(a) for a, _ in prs], [_dev(b) for _, b in prs]
    c = ofxcv.Context(0)
    got = [f.cpu().numpy() for f in c.calc_optical_flow_farneback_batch(da, db, iterations=3)]
    s1 = ofxcv.Context(0)
    for z in (0, 6, 7):
        assert np.array_equal(got[z], s1.calc_optical_flow_farneback(da[z], db[z], iterations=3).cpu().numpy()), z
    ref = oracle.calc_optical_flow_farneback(prs[7][0], prs[7][1], iterations=3, blur_mode=oracle.BLUR_FAITHFUL)
    assert (np.abs(got[7] - ref) <= REL_TOL * np.maximum(1, np.abs(ref))).all()
    c.close()
    s1.close()


def test_column_owning_abort_word_is_reported_once(oracle, ofxcv):
    """A bounded LDS wait of iterate_col_kernel that runs out leaves wrong flows behind: the library's own synchronisation points
    (ofxcv_ctx_synchronize, the host-image entry points) must fail for THAT call and not for the next one.  Forced
    here with farneback.col_spin 1 (a wait gives up after one poll); the same context then repeats the call with the default bound
    and gets the single-call result bit for bit."""
    w, h, n = 333, 257, 2
    prs = _pairs(oracle, w, h, range(610, 610 + n))
    da, db = [_dev(a) for a, _ in prs], [_dev(b) for _, b in prs]
    single = ofxcv.Context(0)
    singles = [single.calc_optical_flow_farneback(x, y, iterations=4).cpu().numpy() for x, y in zip(da, db)]
    single.close()
    c = ofxcv.Context(0)
    c.set_option("farneback.col_min", 1)
    c.set_option("farneback.col_spin", 1)
    c.calc_optical_flow_farneback_batch(da, db, iterations=4)
    if c.get_option("farneback.col_aborts") == 0:
        c.close()
        pytest.skip("no wait ran out with a bound of one poll on this box (timing)")
    with pytest.raises(ofxcv.OfxcvError):
        c.synchronize()
    c.synchronize()  # reported once
    c.set_option("farneback.col_spin", 1 << 22)
    got = [f.cpu().numpy() for f in c.calc_optical_flow_farneback_batch(da, db, iterations=4)]
    c.synchronize()
    for z in range(n):
        assert np.array_equal(got[z], singles[z]), z
    c.close()


def test_column_owning_plan(gpu_ctx):
    """ofxcv_farneback_col_pairs: how many pairs of a call walk level 0 in the column-owning form (256 CUs, one workgroup per
    tile column and pair, launches charged in rounds of the chip against 0.237 x w / 1920 of a round per pair in strips: from 5 pairs at 1920x1080 since round 6)."""
    if gpu_ctx.get_option("farneback.col") != 1:
        pytest.skip("column-owning form switched off on the shared context")
    plan = lambda w, h, n: gpu_ctx.farneback_col_pairs(w, h, n)
    assert [plan(1920, 1080, n) for n in (1, 4, 5, 6, 7, 8, 9, 12, 16)] == [0, 0, 5, 6, 7, 8, 8, 8, 16]
    assert [plan(1921, 1081, n) for n in (8, 16)] == [7, 15]          # 33 tile columns: 264 workgroups would be two rounds
    assert [plan(3840, 2160, n) for n in (1, 2, 3, 4, 5, 8)] == [0, 0, 3, 4, 4, 8]
    assert plan(1920, 40, 8) == 0                                     # too few rows for a column walk
