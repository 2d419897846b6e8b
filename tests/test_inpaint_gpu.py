"""Parity of the inpaint path (C ABI) against the CPU oracle: masks, distance map and fill-order index map
bit-exact (north_star), colours bit-exact too (same IEEE operation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _frame(w, h, seed=1234, holes=12):
    from openfx_opencv_amd import synth
    return synth.inpaint_frame(w, h, seed=seed, n_holes=holes)


@pytest.mark.parametrize("w,h,iters", [(64, 48, 0), (160, 120, 1), (333, 257, 2), (640, 480, 1)])
def test_mask_bit_exact(oracle, gpu_ctx, w, h, iters):
    fr = _frame(w, h)
    fr[5, 7, :3] = (1, 0, 0)     # integer luma 0 -> hole
    fr[5, 9, :3] = (0, 1, 0)     # (9617 + 8192) >> 14 = 1 -> not a hole
    ref = oracle.inpaint_mask(fr, iters)
    got = gpu_ctx.inpaint_mask(_dev(fr), iters).cpu().numpy()
    assert np.array_equal(ref, got)
    if iters == 0:
        assert got[5, 7] == 255 and got[5, 9] == 0


@pytest.mark.parametrize("w,h,radius,cn", [(64, 48, 3, 3), (160, 120, 3, 4), (200, 150, 5, 3), (97, 83, 1, 4)])
def test_telea_maps_and_colours(oracle, gpu_ctx, w, h, radius, cn):
    fr = _frame(w, h, holes=6)
    mask = oracle.inpaint_mask(fr, 1)
    rgb = np.ascontiguousarray(fr[..., :3])
    ref, t_ref, f_ref, ord_ref = oracle.inpaint_telea(rgb, mask, radius, maps=True)
    src = rgb if cn == 3 else fr
    dst, t, order = gpu_ctx.inpaint_telea(_dev(src), _dev(mask), radius, maps=True)
    assert np.array_equal(order.cpu().numpy(), ord_ref)                       # fill-order index map
    assert np.array_equal(t.cpu().numpy(), t_ref)                            # distance map incl. negated ring
    got = dst.cpu().numpy()
    assert np.array_equal(got[..., :3], ref), "colours differ at %d pixels" % (got[..., :3] != ref).any(axis=2).sum()
    if cn == 4:
        assert np.array_equal(got[..., 3], fr[..., 3])
    # only hole pixels change
    assert np.array_equal(got[..., :3][mask == 0], rgb[mask == 0])


def test_telea_edge_cases(oracle, gpu_ctx):
    h, w = 40, 56
    rng = np.random.default_rng(2)
    rgb = rng.integers(1, 255, size=(h, w, 3), dtype=np.uint8)
    # empty mask: identity
    m0 = np.zeros((h, w), np.uint8)
    assert np.array_equal(gpu_ctx.inpaint_telea(_dev(rgb), _dev(m0)).cpu().numpy(), rgb)
    # holes touching every image border, a one-pixel hole, and a symmetric square (FIFO tie order)
    m = np.zeros((h, w), np.uint8)
    m[0:6, 0:9] = 255
    m[h - 5:h, w - 7:w] = 255
    m[20, 30] = 255
    m[10:20, 40:50] = 255
    ref, t_ref, _, ord_ref = oracle.inpaint_telea(rgb, m, 3, maps=True)
    dst, t, order = gpu_ctx.inpaint_telea(_dev(rgb), _dev(m), 3, maps=True)
    assert np.array_equal(order.cpu().numpy(), ord_ref) and np.array_equal(t.cpu().numpy(), t_ref)
    assert np.array_equal(dst.cpu().numpy(), ref)
    # first image row / column are never marched (padded index <= 1): they keep their input colour
    assert ord_ref[0, 0:9].max() == 0 and ord_ref[0:6, 0].max() == 0
    # everything masked: nothing to march from, output == input
    mall = np.full((h, w), 255, np.uint8)
    assert np.array_equal(gpu_ctx.inpaint_telea(_dev(rgb), _dev(mall)).cpu().numpy(), oracle.inpaint_telea(rgb, mall))


def test_render_host_640x480(oracle, gpu_ctx):
    """BASELINE config 1/2 generator at 640x480: whole render() body, host image in, host image out."""
    fr = _frame(640, 480)
    ref = oracle.inpaint_render(fr, 3.0, 1.0)
    got, mask = gpu_ctx.inpaint_render_host(fr, 3.0, 1.0, want_mask=True)
    assert np.array_equal(mask, oracle.inpaint_mask(fr, 1))
    assert np.array_equal(got, ref)
    assert (got[..., 3] == 255).all()


def test_render_host_1080p_properties(oracle, gpu_ctx):
    """BASELINE config 2 size: compared with the oracle in full (0.3 s of CPU) + idempotence on a hole-free frame."""
    fr = _frame(1920, 1080)
    got = gpu_ctx.inpaint_render_host(fr, 3.0, 1.0)
    assert np.array_equal(got, oracle.inpaint_render(fr, 3.0, 1.0))
    clean = fr.copy()
    clean[..., :3] = np.maximum(clean[..., :3], 1)
    clean[(clean[..., :3] == 0).all(axis=2)] = (9, 9, 9, 255)
    again = gpu_ctx.inpaint_render_host(got, 3.0, 1.0)
    hole_left = oracle.inpaint_mask(got, 1) > 0
    assert np.array_equal(again[~hole_left], got[~hole_left])


def test_telea_large_radii(oracle, ofxcv, gpu_ctx):
    """Radius 6 .. 12 takes the large-window instantiation of the dataflow fill (the (2r+3)^2 neighbourhood of a pixel still fits
    LDS with eight wavefronts per workgroup: up to 729 entries, ten tap chunks at radius 12), larger radii the kernel variant without
    the LDS-staged neighbourhood (barrier-scheduled levels).  Same maps and colours as the oracle either way, for both methods,
    and when the large-window fill is made to give up (spin limit 0: the barrier-scheduled repeat)."""
    fr = _frame(120, 90, holes=4)
    mask = oracle.inpaint_mask(fr, 1)
    rgb = np.ascontiguousarray(fr[..., :3])
    for radius in (6, 9, 12, 14):
        ref, t_ref, _, ord_ref = oracle.inpaint_telea(rgb, mask, radius, maps=True)
        dst, t, order = gpu_ctx.inpaint_telea(_dev(rgb), _dev(mask), radius, maps=True)
        assert np.array_equal(order.cpu().numpy(), ord_ref) and np.array_equal(t.cpu().numpy(), t_ref), radius
        assert np.array_equal(dst.cpu().numpy(), ref), radius
    for radius in (7, 13):
        ref = oracle.inpaint(rgb, mask, radius, oracle.INPAINT_NS)
        assert np.array_equal(gpu_ctx.inpaint(_dev(rgb), _dev(mask), radius, ofxcv.INPAINT_NS).cpu().numpy(), ref), radius
    c = ofxcv.Context(0)
    c.set_option("inpaint.spin_limit", 0)
    ref = oracle.inpaint_telea(rgb, mask, 8)
    assert np.array_equal(c.inpaint_telea(_dev(rgb), _dev(mask), 8).cpu().numpy(), ref) and c.inpaint_fallback_count() >= 1
    c.close()
    big = _frame(640, 360, holes=8)                      # several portions and many tiles at a large radius
    ref = oracle.inpaint_render(big, 10.0, 1.0)
    assert np.array_equal(gpu_ctx.inpaint_render_host(big, 10.0, 1.0), ref)


# ---- CV_INPAINT_NS: the other value of cvInpaint's method argument ----

@pytest.mark.parametrize("w,h,radius,cn", [(64, 48, 3, 3), (160, 120, 3, 4), (200, 150, 5, 3), (97, 83, 1, 4), (120, 90, 12, 3)])
def test_ns_maps_and_colours(oracle, ofxcv, gpu_ctx, w, h, radius, cn):
    fr = _frame(w, h, holes=6)
    mask = oracle.inpaint_mask(fr, 1)
    rgb = np.ascontiguousarray(fr[..., :3])
    ref, t_ref, f_ref, ord_ref = oracle.inpaint(rgb, mask, radius, oracle.INPAINT_NS, maps=True)
    src = rgb if cn == 3 else fr
    dst, t, order = gpu_ctx.inpaint(_dev(src), _dev(mask), radius, ofxcv.INPAINT_NS, maps=True)
    assert np.array_equal(order.cpu().numpy(), ord_ref)
    assert np.array_equal(t.cpu().numpy(), t_ref)                            # no negated ring: 1e6 off the band
    got = dst.cpu().numpy()
    assert np.array_equal(got[..., :3], ref), "colours differ at %d pixels" % (got[..., :3] != ref).any(axis=2).sum()
    known = mask == 0
    assert np.array_equal(got[..., :3][known], rgb[known])


def test_ns_constant_colour_and_bad_method(oracle, ofxcv, gpu_ctx):
    c = np.full((40, 50, 3), 77, np.uint8)
    m = np.zeros((40, 50), np.uint8)
    m[10:20, 15:30] = 255
    got = gpu_ctx.inpaint(_dev(c), _dev(m), 3.0, ofxcv.INPAINT_NS).cpu().numpy()
    assert np.all(got == 77)
    with pytest.raises(ofxcv.OfxcvError) as e:
        gpu_ctx.inpaint(_dev(c), _dev(m), 3.0, 7)
    assert e.value.status == -4


@pytest.mark.timeout(180)
def test_concurrent_inpaint_calls(oracle, ofxcv):
    """Eight host threads, one context each, inpaint the same frame at once: the dataflow fill kernels of different
    calls share the device (at most four are in flight), every result is the oracle's."""
    import threading
    import torch  # noqa: F401
    fr = _frame(320, 240, holes=8)
    ref = oracle.inpaint_render(fr, 3.0, 1.0)
    results, errors = {}, []

    def work(k):
        try:
            c = ofxcv.Context(0)
            for _ in range(3):
                results[k] = c.inpaint_render_host(fr, 3.0, 1.0)
            c.close()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(8):
        assert np.array_equal(results[k], ref), k


@pytest.mark.timeout(300)
def test_concurrent_large_fills_share_the_tile_budget(oracle, ofxcv):
    """Four host threads inpaint a 1280x720 frame at once.  A fill launch takes its share of the chip's resident workgroups
    (192 over the fills in flight, re-read per portion of the fill order), so calls that start while another call's larger
    launch is running overlap with it: every result must still be the oracle's, and so must a call with the cap forced to
    its extremes."""
    import threading
    fr = _frame(1280, 720, holes=12)
    ref = oracle.inpaint_render(fr, 3.0, 1.0)
    results, errors, fallbacks = {}, [], []

    def work(k):
        try:
            c = ofxcv.Context(0)
            for _ in range(3):
                results[k] = c.inpaint_render_host(fr, 3.0, 1.0)
            fallbacks.append(c.inpaint_fallback_count())
            c.close()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(4):
        assert np.array_equal(results[k], ref), k
    print("fallbacks of the four concurrent contexts:", fallbacks)
    for cap in (1, 48, 240):
        c = ofxcv.Context(0)
        c.set_option("inpaint.max_tiles", cap)
        assert np.array_equal(c.inpaint_render_host(fr, 3.0, 1.0), ref), cap
        c.close()


def test_index_map_known_answers_on_the_hip_path(gpu_ctx):
    """The analytic known-answer tests that pin the oracle (tests/test_oracle_inpaint_segment.py) run against the maps
    ofxcv_inpaint_telea itself returns, so the distance / fill-order parity is not only oracle-vs-product: straight-edge
    hole -> T = 1, 2, 3, ... and a negated ring outside; row-major seeding + FIFO ties -> first hole column filled top to
    bottom; symmetric hole -> every pixel filled exactly once, top-left corner first; constant colour comes back."""
    h, w = 24, 40
    rgb = np.full((h, w, 3), 100, np.uint8)
    mask = np.zeros((h, w), np.uint8)
    mask[:, 20:] = 255
    dst, t, order = gpu_ctx.inpaint_telea(_dev(rgb), _dev(mask), 3.0, maps=True)
    t, order = t.cpu().numpy(), order.cpu().numpy()
    row = t[12, 1:-1]
    assert row[19] == 0.0
    assert np.allclose(row[20:30], np.arange(1, 11), atol=1e-5)
    assert np.allclose(row[17:19], [-2, -1], atol=1e-5)
    assert row[16] == 1.0e6 and row[10] == 1.0e6
    col = order[:, 20]
    assert col[0] == 0 and np.array_equal(col[1:], np.arange(1, h))
    assert order[:, :20].max() == 0
    c = np.full((40, 50, 3), (10, 120, 200), np.uint8)                         # constant image + bounded hole: comes back within
    mc = np.zeros((40, 50), np.uint8)                                          # the reference's two roundings
    mc[10:25, 15:30] = 255
    oc = gpu_ctx.inpaint_telea(_dev(c), _dev(mc), 3.0).cpu().numpy()
    assert np.abs(oc.astype(int) - c).max() <= 3 and np.array_equal(oc[mc == 0], c[mc == 0])
    mask2 = np.zeros((21, 21), np.uint8)
    mask2[6:15, 6:15] = 255
    _, t2, o2 = gpu_ctx.inpaint_telea(_dev(np.full((21, 21, 3), 50, np.uint8)), _dev(mask2), 3.0, maps=True)
    t2, o2 = t2.cpu().numpy(), o2.cpu().numpy()
    assert o2[6, 6] == 1 and o2[6, 6] < o2[6, 14] < o2[14, 6] < o2[14, 14]
    assert o2.max() == 81 and sorted(o2[o2 > 0].tolist()) == list(range(1, 82))
    assert t2[1:-1, 1:-1][10, 10] > t2[1:-1, 1:-1][6, 6] > 0
    # mask rule on the HIP path: (1,0,0) and (0,0,4) are holes, (0,1,0) and (0,0,5) are not
    px = np.zeros((1, 6, 4), np.uint8)
    px[0, :, 3] = 255
    px[0, 1, :3] = (1, 0, 0)
    px[0, 2, :3] = (0, 1, 0)
    px[0, 3, :3] = (0, 0, 4)
    px[0, 4, :3] = (0, 0, 5)
    px[0, 5, :3] = (255, 255, 255)
    assert gpu_ctx.inpaint_mask(_dev(px), 0).cpu().numpy()[0].tolist() == [255, 255, 0, 255, 0, 0]


def test_dataflow_fill_bounded_polls_fall_back_to_the_barrier_kernel(oracle, ofxcv):
    """the polls of the dataflow fill are bounded; with the bound forced to 0 every awaited colour 'times out', the launch
    raises its error flag and the fill is repeated with the barrier-scheduled kernel: same colours, and the context
    counts the fall-back.  With the default bound there is no fall-back."""
    fr = _frame(333, 257, holes=8)
    mask = oracle.inpaint_mask(fr, 1)
    rgb = np.ascontiguousarray(fr[..., :3])
    ref = oracle.inpaint_telea(rgb, mask, 3.0)
    c = ofxcv.Context(0)
    assert np.array_equal(c.inpaint_telea(_dev(rgb), _dev(mask), 3.0).cpu().numpy(), ref) and c.inpaint_fallback_count() == 0
    c.set_option("inpaint.spin_limit", 0)
    assert np.array_equal(c.inpaint_telea(_dev(rgb), _dev(mask), 3.0).cpu().numpy(), ref)
    assert c.inpaint_fallback_count() == 1
    assert np.array_equal(c.inpaint(_dev(rgb), _dev(mask), 3.0, ofxcv.INPAINT_NS).cpu().numpy(), oracle.inpaint(rgb, mask, 3.0, oracle.INPAINT_NS))
    assert c.inpaint_fallback_count() == 2
    c.close()


def test_one_context_many_masks_and_sizes(oracle, ofxcv):
    """the front-march state (host and device distance / order maps) persists with the context and is reset sparsely: masks of
    every kind and frame sizes in any order on ONE context, both methods, maps and colours bit-exact each time.  Includes the
    almost-everything-is-a-hole mask (no ring pixel exists, the reference still marches) and the pipelined fill with tiny
    portions (several launches per call)."""
    rng = np.random.default_rng(3)
    ctx = ofxcv.Context(0)
    ctx.set_option("inpaint.portion", 700)
    for case in range(15):
        w, h = int(rng.integers(8, 200)), int(rng.integers(8, 140))
        if case % 4 == 3:
            w, h = 120, 90                                            # same size again: sparse reset instead of re-initialisation
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        mask = np.zeros((h, w), np.uint8)
        kind = case % 5
        if kind == 0:
            mask[rng.random((h, w)) < 0.05] = 255                    # salt
        elif kind == 1:
            mask[h // 4:h // 2, w // 5:w - 3] = 255                  # rectangle
            mask[0:4, 0:6] = 255                                     # touching the border
        elif kind == 2:
            mask[:, ::7] = 255                                       # vertical lines
        elif kind == 3:
            mask[:] = 255
            mask[h // 2, w // 2] = 0                                 # almost everything is a hole
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            mask[(xx - w / 2) ** 2 + (yy - h / 2) ** 2 < (min(w, h) / 3) ** 2] = 255
        radius = float(rng.choice([1, 3, 5, 7]))
        method = case % 2
        ref, t_ref, f_ref, o_ref = oracle.inpaint(rgb, mask, radius, method, maps=True)
        got, t, order = ctx.inpaint(_dev(rgb), _dev(mask), radius, method, maps=True)
        assert np.array_equal(order.cpu().numpy(), o_ref), (case, kind)
        assert np.array_equal(t.cpu().numpy(), t_ref), (case, kind)
        assert np.array_equal(got.cpu().numpy(), ref), (case, kind)
    assert ctx.inpaint_fallback_count() == 0
    ctx.close()


def test_parallel_front_march_option_gives_the_same_maps_and_colours(oracle, ofxcv):
    """option inpaint.parallel_march: the hole's 4-connected components are marched on host threads and merged into the exact
    sequential fill order (telea_march.h) -- distance map, order map and colours equal the serial form's and the oracle's"""
    import torch
    from openfx_opencv_amd import synth
    fr = synth.inpaint_frame(640, 480)
    d = torch.from_numpy(fr).cuda()
    outs = []
    for par in (0, 1):
        c = ofxcv.Context(0)
        c.set_option("inpaint.parallel_march", 2 if par else 0)   # (2: from two hole pixels on)
        m = c.inpaint_mask(d, 1)
        for _ in range(2):                                         # twice: the host state is reused
            dst, t, order = c.inpaint_telea(d, m, 3.0, maps=True)
        torch.cuda.synchronize()
        outs.append((dst.cpu().numpy(), t.cpu().numpy(), order.cpu().numpy()))
        c.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    ref = oracle.inpaint_render(fr, 3.0, 1.0)
    assert np.array_equal(outs[1][0][..., :3], ref[..., :3])
