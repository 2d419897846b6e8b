"""Synthetic inputs for the hot path: seeded, numpy only.

The reference ships no test data, so every workload is generated:
  * flow_pair(): an f32 linear RGBA frame pair related by a smooth known warp (configs 3 and 5)
  * inpaint_frame(): the same texture as 8-bit RGBA with seeded black ellipses (configs 1 and 2)
"""
import numpy as np


def _box3(a):
    p = np.pad(a, 1, mode="edge")
    h, w = a.shape
    s = np.zeros_like(a)
    for dy in range(3):
        for dx in range(3):
            s += p[dy:dy + h, dx:dx + w]
    return s / 9.0


def texture(w, h, seed=1234):
    """Luminance in [0,1]: 0.5 + 0.25 sin(2 pi x/64) cos(2 pi y/48) + 0.15 n, n = 3x3-boxed uniform(-1,1)."""
    rng = np.random.default_rng(seed)
    n = _box3(rng.uniform(-1.0, 1.0, size=(h, w)))
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    lum = 0.5 + 0.25 * np.sin(2 * np.pi * x / 64.0) * np.cos(2 * np.pi * y / 48.0) + 0.15 * n
    return np.clip(lum, 0.0, 1.0)


def known_flow(w, h):
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    u = 2.5 + 1.0 * np.sin(2 * np.pi * y / h) + 0 * x
    v = -1.25 + 0.75 * np.cos(2 * np.pi * x / w) + 0 * y
    return u, v


def _bilinear(img, sx, sy):
    h, w = img.shape
    sx = np.clip(sx, 0, w - 1)
    sy = np.clip(sy, 0, h - 1)
    x0 = np.minimum(np.floor(sx).astype(np.int64), w - 2)
    y0 = np.minimum(np.floor(sy).astype(np.int64), h - 2)
    fx = sx - x0
    fy = sy - y0
    return ((1 - fx) * (1 - fy) * img[y0, x0] + fx * (1 - fy) * img[y0, x0 + 1]
            + (1 - fx) * fy * img[y0 + 1, x0] + fx * fy * img[y0 + 1, x0 + 1])


def _rgba(lum):
    h, w = lum.shape
    out = np.empty((h, w, 4), np.float32)
    out[..., 0] = out[..., 1] = out[..., 2] = lum.astype(np.float32)
    out[..., 3] = 1.0
    return out


def flow_pair(w, h, seed=1234):
    """(frame_a, frame_b) f32 RGBA HxWx4; b(q) = a(q - d(q)) with d = known_flow, so flow a->b ~ d."""
    a = texture(w, h, seed)
    u, v = known_flow(w, h)
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    b = _bilinear(a, x - u, y - v)
    return _rgba(a), _rgba(b)


def sequence(w, h, n, seed=1234):
    """n consecutive f32 RGBA frames of one shot: frame k(q) = a(q - k d(q)) with d = known_flow, so neighbouring frames differ by ~d
    (2.5 +- 1 px, -1.25 +- 0.75 px) in either direction -- what the three source frames of a VectorGenerator output frame (t - 1, t, t + 1,
    VectorGenerator.cpp:597-638) look like during playback."""
    a = texture(w, h, seed)
    u, v = known_flow(w, h)
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    return [_rgba(a if k == 0 else _bilinear(a, x - k * u, y - k * v)) for k in range(n)]


def pingpong(t, n):
    """index into n buffers for frame time t of an endless shot that runs 0, 1, ..., n-1, n-2, ..., 1, 0, 1, ...: neighbouring times are
    always neighbouring frames"""
    p = 2 * (n - 1)
    r = t % p
    return r if r < n else p - r


def inpaint_frame(w, h, seed=1234, hole_seed=42, n_holes=12):
    """8-bit RGBA texture (every channel >= 1) with n_holes seeded filled ellipses set to (0,0,0,255)."""
    lum = texture(w, h, seed)
    rgb = np.empty((h, w, 4), np.uint8)
    base = np.clip(np.rint(lum * 255.0), 1, 255).astype(np.uint8)
    # de-correlate the channels a little so the three colour planes are not identical
    rgb[..., 0] = base
    rgb[..., 1] = np.clip(np.rint(lum * 200.0 + 20), 1, 255).astype(np.uint8)
    rgb[..., 2] = np.clip(255 - np.rint(lum * 180.0), 1, 255).astype(np.uint8)
    rgb[..., 3] = 255
    rng = np.random.default_rng(hole_seed)
    yy, xx = np.mgrid[0:h, 0:w]
    s = min(w, h) / 480.0
    for _ in range(n_holes):
        cx = rng.uniform(0, w)
        cy = rng.uniform(0, h)
        ra = rng.uniform(8, 40) * s
        rb = rng.uniform(8, 40) * s
        th = rng.uniform(0, np.pi)
        dx = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        dy = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        m = (dx / ra) ** 2 + (dy / rb) ** 2 <= 1.0
        rgb[m, 0:3] = 0
    return rgb
