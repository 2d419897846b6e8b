"""ctypes binding of oracle/liboracle.so (test infrastructure only; PARITY UNPINNED, see ofxcv_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BLUR_FAITHFUL = 0
BLUR_DIRECT = 1


def build():
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def cv_round(v):
    f = lib().orc_cv_round
    f.restype = C.c_int
    return f(C.c_double(v))


def gaussian_kernel(n, sigma):
    k = np.empty(n, np.float32)
    lib().orc_gaussian_kernel_f32(C.c_int(n), C.c_double(sigma), _p(k))
    return k


def gaussian_blur(src, ksize, sigma):
    src = np.ascontiguousarray(src, np.float32)
    h, w = src.shape
    dst = np.empty_like(src)
    lib().orc_gaussian_blur_f32(_p(src), C.c_int(w), C.c_int(h), _p(dst), C.c_int(ksize), C.c_double(sigma))
    return dst


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.float32)
    if src.ndim == 2:
        src = src[:, :, None]
    sh, sw, cn = src.shape
    dst = np.empty((dh, dw, cn), np.float32)
    lib().orc_resize_linear_f32(_p(src), C.c_int(sw), C.c_int(sh), C.c_int(cn), _p(dst), C.c_int(dw), C.c_int(dh))
    return dst[:, :, 0] if cn == 1 else dst


def polyexp_prepare(n, sigma):
    g = np.empty(2 * n + 1, np.float32)
    xg = np.empty(2 * n + 1, np.float32)
    xxg = np.empty(2 * n + 1, np.float32)
    ig = np.empty(4, np.float64)
    lib().orc_polyexp_prepare(C.c_int(n), C.c_double(sigma), _p(g), _p(xg), _p(xxg), _p(ig))
    return g, xg, xxg, ig


def polyexp(I, n, sigma):
    I = np.ascontiguousarray(I, np.float32)
    h, w = I.shape
    R = np.empty((h, w, 5), np.float32)
    lib().orc_polyexp(_p(I), C.c_int(w), C.c_int(h), _p(R), C.c_int(n), C.c_double(sigma))
    return R


def update_matrices(R0, R1, flow):
    h, w, _ = R0.shape
    M = np.empty((h, w, 5), np.float32)
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    flow = np.ascontiguousarray(flow, np.float32)
    lib().orc_update_matrices(_p(R0), _p(R1), _p(flow), _p(M), C.c_int(w), C.c_int(h), C.c_int(0), C.c_int(h))
    return M


def update_flow_blur(R0, R1, M, block_size=3, update=True, mode=BLUR_FAITHFUL):
    """returns (flow, M_new); M is not modified."""
    h, w, _ = R0.shape
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    M = np.array(M, np.float32, copy=True, order="C")
    flow = np.zeros((h, w, 2), np.float32)
    lib().orc_update_flow_blur(_p(R0), _p(R1), _p(flow), _p(M), C.c_int(w), C.c_int(h), C.c_int(block_size),
                               C.c_int(1 if update else 0), C.c_int(mode))
    return flow, M


def farneback_num_levels(w, h, pyr_scale=0.5, levels=3):
    f = lib().orc_farneback_num_levels
    f.restype = C.c_int
    return f(C.c_int(w), C.c_int(h), C.c_double(pyr_scale), C.c_int(levels))


def farneback_level_geom(w, h, pyr_scale, k):
    lw, lh, ks = C.c_int(), C.c_int(), C.c_int()
    sg = C.c_double()
    lib().orc_farneback_level_geom(C.c_int(w), C.c_int(h), C.c_double(pyr_scale), C.c_int(k),
                                   C.byref(lw), C.byref(lh), C.byref(sg), C.byref(ks))
    return lw.value, lh.value, sg.value, ks.value


def farneback_pyr_image(img, lw, lh, sigma, ksize):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    I = np.empty((lh, lw), np.float32)
    lib().orc_farneback_pyr_image(_p(img), C.c_size_t(w), C.c_int(w), C.c_int(h), C.c_int(lw), C.c_int(lh),
                                  C.c_double(sigma), C.c_int(ksize), _p(I))
    return I


OPTFLOW_USE_INITIAL_FLOW = 4
OPTFLOW_FARNEBACK_GAUSSIAN = 256


def set_gaussian_kernel_generation(generation):
    """3 (default) = getGaussianKernel of OpenCV 2.4 / 3.x, 4 = 4.x; only tests/test_cv2_crosscheck.py flips it"""
    lib().orc_set_gaussian_kernel_generation(C.c_int(generation))


def set_luma(standard):
    """709 (default) or 601: the luma weights of to_byte_grayscale"""
    lib().orc_set_luma(C.c_int(int(standard)))


def set_filter_contraction(on):
    """1: separable filters / resize's vertical lerp with fused multiply-adds (OpenCV 4.x AVX2 / NEON paths); 0 (default): scalar order"""
    lib().orc_set_filter_contraction(C.c_int(int(on)))


def set_resize_generation(generation):
    """association of cv::resize's exact-2x INTER_AREA rewrite: 0 (default) bilinear = 4.x SIMD, 1 scalar loop, 2 3.x SSE2"""
    lib().orc_set_resize_generation(C.c_int(generation))


def calc_optical_flow_farneback(prev, nxt, pyr_scale=0.5, levels=3, winsize=3, iterations=15, poly_n=5,
                                poly_sigma=1.1, flags=0, blur_mode=BLUR_FAITHFUL, initial_flow=None):
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    assert prev.shape == nxt.shape and prev.ndim == 2
    h, w = prev.shape
    if flags & OPTFLOW_USE_INITIAL_FLOW:
        flow = np.array(initial_flow, np.float32, order="C", copy=True)
        assert flow.shape == (h, w, 2)
    else:
        flow = np.empty((h, w, 2), np.float32)
    f = lib().orc_calc_optical_flow_farneback
    f.restype = C.c_int
    rc = f(_p(prev), _p(nxt), C.c_size_t(w), C.c_int(w), C.c_int(h), _p(flow), C.c_double(pyr_scale),
           C.c_int(levels), C.c_int(winsize), C.c_int(iterations), C.c_int(poly_n), C.c_double(poly_sigma),
           C.c_int(flags), C.c_int(blur_mode))
    if rc != 0:
        raise ValueError("orc_calc_optical_flow_farneback rc=%d" % rc)
    return flow


def update_flow_gaussian(R0, R1, flow, M, winsize, update):
    """One FarnebackUpdateFlow_GaussianBlur pass; returns (flow, M) copies."""
    R0 = np.ascontiguousarray(R0, np.float32); R1 = np.ascontiguousarray(R1, np.float32)
    flow = np.array(flow, np.float32, order="C", copy=True); M = np.array(M, np.float32, order="C", copy=True)
    h, w, _ = flow.shape
    lib().orc_update_flow_gaussian(_p(R0), _p(R1), _p(flow), _p(M), C.c_int(w), C.c_int(h), C.c_int(winsize), C.c_int(int(update)))
    return flow, M


def resize_area(src, dw, dh):
    src = np.ascontiguousarray(src, np.float32)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.float32)
    lib().orc_resize_area_f32(_p(src), C.c_int(sw), C.c_int(sh), C.c_int(cn), _p(dst), C.c_int(dw), C.c_int(dh))
    return dst


def srgb_lut():
    lut = np.empty(65536, np.uint16)
    lib().orc_srgb_lut_build(_p(lut))
    return lut


def to_byte_grayscale(img):
    """img: HxWx{3,4} float32 linear RGB(A) -> HxW uint8 sRGB luma."""
    img = np.ascontiguousarray(img, np.float32)
    h, w, nc = img.shape
    dst = np.empty((h, w), np.uint8)
    lib().orc_to_byte_grayscale(_p(img), C.c_ssize_t(w * nc * 4), C.c_int(nc), C.c_int(w), C.c_int(h), _p(dst),
                                C.c_ssize_t(w))
    return dst


def flow_to_rgba(flow, dst, chan_u, chan_v, rs_x=1.0, rs_y=1.0):
    flow = np.ascontiguousarray(flow, np.float32)
    h, w, _ = flow.shape
    assert dst.dtype == np.float32 and dst.shape == (h, w, 4) and dst.flags.c_contiguous
    cu = (C.c_int * 4)(*[int(v) for v in chan_u])
    cv = (C.c_int * 4)(*[int(v) for v in chan_v])
    lib().orc_flow_to_rgba(_p(flow), C.c_int(w), C.c_int(h), _p(dst), C.c_ssize_t(w * 16), cu, cv,
                           C.c_double(rs_x), C.c_double(rs_y))
    return dst


# ---- inpaint ----------------------------------------------------------------------------------------------
def inpaint_mask(rgba, dilate_iters=1):
    rgba = np.ascontiguousarray(rgba, np.uint8)
    h, w, _ = rgba.shape
    mask = np.empty((h, w), np.uint8)
    lib().orc_inpaint_mask(_p(rgba), C.c_ssize_t(w * 4), C.c_int(w), C.c_int(h), C.c_int(dilate_iters), _p(mask))
    return mask


INPAINT_NS = 0
INPAINT_TELEA = 1


def inpaint(rgb, mask, radius=3.0, method=INPAINT_TELEA, maps=False):
    """cvInpaint(rgb, mask, out, radius, method); maps=True also returns (t, f, order)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w, _ = rgb.shape
    out = np.empty_like(rgb)
    t = np.empty((h + 2, w + 2), np.float32)
    f = np.empty((h + 2, w + 2), np.uint8)
    order = np.empty((h, w), np.int32)
    fn = lib().orc_inpaint
    fn.restype = C.c_int
    rc = fn(_p(rgb), _p(mask), C.c_int(w), C.c_int(h), C.c_double(radius), C.c_int(method), _p(out), _p(t), _p(f), _p(order))
    if rc != 0:
        raise ValueError("orc_inpaint rc=%d" % rc)
    return (out, t, f, order) if maps else out


def inpaint_telea(rgb, mask, radius=3.0, maps=False):
    return inpaint(rgb, mask, radius, INPAINT_TELEA, maps)


def inpaint_render(rgba, radius=3.0, dilation=1.0):
    rgba = np.ascontiguousarray(rgba, np.uint8)
    h, w, _ = rgba.shape
    dst = np.empty_like(rgba)
    fn = lib().orc_inpaint_render
    fn.restype = C.c_int
    fn(_p(rgba), C.c_ssize_t(w * 4), C.c_int(w), C.c_int(h), C.c_double(radius), C.c_double(dilation), _p(dst), C.c_ssize_t(w * 4))
    return dst


# ---- segment (mean-shift) ---------------------------------------------------------------------------------
def pyr_mean_shift(rgb, sp=10.0, sr=20.0, max_level=2, max_iter=5, eps=1.0):
    """cv::pyrMeanShiftFiltering(rgb, dst, sp, sr, max_level, TermCriteria(ITER+EPS, max_iter, eps))."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.empty_like(rgb)
    fn = lib().orc_pyr_mean_shift
    fn.restype = C.c_int
    rc = fn(_p(rgb), C.c_int(w), C.c_int(h), C.c_double(sp), C.c_double(sr), C.c_int(max_level), C.c_int(max_iter), C.c_double(eps), _p(out))
    if rc != 0:
        raise ValueError("orc_pyr_mean_shift rc=%d" % rc)
    return out
