"""Known-answer tests pinning the CPU oracle's inpaint and mean-shift restatements (SURVEY.md 8(c) KATs 6-8). CPU only."""
import numpy as np
import pytest


def test_mask_rule_integer_luma(oracle):
    px = np.zeros((1, 6, 4), np.uint8)
    px[0, :, 3] = 255
    px[0, 0, :3] = (0, 0, 0)      # hole
    px[0, 1, :3] = (1, 0, 0)      # (4899 + 8192) >> 14 = 0 -> hole
    px[0, 2, :3] = (0, 1, 0)      # (9617 + 8192) >> 14 = 1 -> not a hole
    px[0, 3, :3] = (0, 0, 4)      # (7472 + 8192) >> 14 = 0 -> hole
    px[0, 4, :3] = (0, 0, 5)      # (9340 + 8192) >> 14 = 1
    px[0, 5, :3] = (255, 255, 255)
    assert oracle.inpaint_mask(px, 0)[0].tolist() == [255, 255, 0, 255, 0, 0]
    # dilation: n iterations of a 3x3 rect == one (2n+1)^2 rect, clipped at the border
    img = np.full((9, 9, 4), 200, np.uint8)
    img[4, 4, :3] = 0
    for n in (1, 2, 3):
        m = oracle.inpaint_mask(img, n)
        ref = np.zeros((9, 9), np.uint8)
        ref[4 - n:5 + n, 4 - n:5 + n] = 255
        assert np.array_equal(m, ref)


def test_telea_distance_map_straight_edge_and_fifo_order(oracle):
    h, w = 24, 40
    rgb = np.full((h, w, 3), 100, np.uint8)
    mask = np.zeros((h, w), np.uint8)
    mask[:, 20:] = 255                                   # half-plane hole with a straight vertical edge at x = 20
    out, t, f, order = oracle.inpaint_telea(rgb, mask, 3, maps=True)
    row = t[12, 1:-1]                                    # padded map, interior row
    assert row[19] == 0.0                                # band pixel left of the hole
    assert np.allclose(row[20:30], np.arange(1, 11), atol=1e-5)     # T = 1, 2, 3, ... into the hole
    assert np.allclose(row[17:19], [-2, -1], atol=1e-5)             # negated outward distances: ring = 7x7 dilation - hole - band
    assert row[16] == 1.0e6 and row[10] == 1.0e6         # beyond the ring: untouched
    # the band is seeded in row-major order and ties pop FIFO: the first hole column is filled top to bottom
    # (image row 0 is never marched: padded row index 1 is skipped)
    col = order[:, 20]
    assert col[0] == 0 and np.array_equal(col[1:], np.arange(1, h))
    assert order[:, :20].max() == 0
    # a symmetric hole: T is assigned once, at first contact, so neither T nor the fill order is symmetric --
    # the row-major seeding and the FIFO ties make the top-left corner go first
    mask2 = np.zeros((21, 21), np.uint8)
    mask2[6:15, 6:15] = 255
    _, t2, _, o2 = oracle.inpaint_telea(np.full((21, 21, 3), 50, np.uint8), mask2, 3, maps=True)
    assert o2[6, 6] == 1 and o2[6, 6] < o2[6, 14] < o2[14, 6] < o2[14, 14]
    assert o2.max() == 81 and sorted(o2[o2 > 0].tolist()) == list(range(1, 82))          # every hole pixel filled exactly once
    assert t2[1:-1, 1:-1][10, 10] > t2[1:-1, 1:-1][6, 6] > 0                              # the centre is the farthest


def test_telea_colours_near_constant_and_only_hole_changes(oracle):
    rgb = np.full((40, 50, 3), (10, 120, 200), np.uint8)
    mask = np.zeros((40, 50), np.uint8)
    mask[10:25, 15:30] = 255
    out = oracle.inpaint_telea(rgb, mask, 3)
    # the reference adds 0.5 and then rounds to nearest (two roundings): a constant image comes back within 3 levels
    assert np.abs(out.astype(int) - rgb).max() <= 3
    assert np.array_equal(out[mask == 0], rgb[mask == 0])
    empty = oracle.inpaint_telea(rgb, np.zeros_like(mask), 3)
    assert np.array_equal(empty, rgb)


def test_inpaint_render_fills_holes_plausibly(oracle):
    from openfx_opencv_amd import synth
    fr = synth.inpaint_frame(160, 120)
    res = oracle.inpaint_render(fr, 3.0, 1.0)
    hole = oracle.inpaint_mask(fr, 1) > 0
    assert (res[..., 3] == 255).all() and np.array_equal(res[~hole][:, :3], fr[~hole][:, :3])
    inner = hole.copy()
    inner[0, :] = inner[:, 0] = False                    # row / column 0 are never marched
    assert (res[inner][:, :3].max(axis=1) > 0).mean() > 0.99      # holes are no longer black


def test_mean_shift_kats(oracle):
    const = np.full((37, 53, 3), (10, 120, 201), np.uint8)
    assert np.array_equal(oracle.pyr_mean_shift(const, 10, 20, 2), const)            # constant image -> identity
    two = const.copy()
    two[:, 26:] = (200, 30, 90)
    assert np.array_equal(oracle.pyr_mean_shift(two, 10, 20, 0), two)                # tones further apart than sr -> identity
    # a noisy flat patch is smoothed towards its mean, and the result stays within the input range
    rng = np.random.default_rng(1)
    noisy = np.clip(rng.normal(128, 4, size=(40, 40, 3)), 0, 255).astype(np.uint8)
    out = oracle.pyr_mean_shift(noisy, 5, 30, 0)
    assert out.astype(float).std() < 0.5 * noisy.astype(float).std()
    assert out.min() >= noisy.min() and out.max() <= noisy.max()


def test_ns_shares_the_front_and_keeps_constant_colour(oracle):
    """CV_INPAINT_NS: same fill order and inside distances as Telea (the march depends on the mask only), no negated
    ring outside, and a constant image is reproduced exactly."""
    from openfx_opencv_amd import synth
    fr = synth.inpaint_frame(96, 72, n_holes=4)
    mask = oracle.inpaint_mask(fr, 1)
    rgb = np.ascontiguousarray(fr[..., :3])
    a, ta, fa, oa = oracle.inpaint(rgb, mask, 3.0, oracle.INPAINT_TELEA, maps=True)
    b, tb, fb, ob = oracle.inpaint(rgb, mask, 3.0, oracle.INPAINT_NS, maps=True)
    assert np.array_equal(oa, ob)
    inside = mask > 0
    assert np.array_equal(ta[1:-1, 1:-1][inside], tb[1:-1, 1:-1][inside])
    assert ta.min() < 0 and tb.min() == 0                                   # Telea marches outwards too, NS does not
    assert np.array_equal(a[~inside], rgb[~inside]) and np.array_equal(b[~inside], rgb[~inside])
    assert np.abs(a.astype(int) - b.astype(int))[inside].mean() < 12       # two colour rules, similar fills
    c = np.full((30, 40, 3), 123, np.uint8)
    m = np.zeros((30, 40), np.uint8)
    m[8:14, 10:22] = 255
    assert np.all(oracle.inpaint(c, m, 3.0, oracle.INPAINT_NS) == 123)
    with pytest.raises(ValueError):
        oracle.inpaint(c, m, 3.0, 5)
