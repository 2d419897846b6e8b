"""openfx-opencv on MI355X: Python host-side mirror of the C ABI in include/ofxcv_hip.h.

The product is lib/libofxcv_hip.so (hand-written gfx950 HIP kernels behind a C ABI) plus the OFX
plugin bundles built from plugin/.  This module is test/bench plumbing around it: it loads the
library with ctypes and passes torch CUDA(ROCm) tensors' device pointers and the current torch
stream straight through.  There is NO CPU fallback: if the library is missing or fails to load,
importing a compute entry point raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFXCV_LIB_PATH") or os.path.join(_HERE, "lib", "libofxcv_hip.so")  # (override: A/B of two builds)

OK = 0
_lib = None


class OfxcvError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("ofxcv status %d: %s" % (status, text))
        self.status = status


def build(verbose=False):
    """Compile libofxcv_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", _HERE, "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """The loaded C-ABI library.  import torch first so both share one libamdhip64.so.7."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OfxcvError(-5, "libofxcv_hip.so not built (%s); run __graft_entry__.build()" % LIB_PATH)
        try:
            import torch  # noqa: F401  (loads torch's HIP runtime so SONAME libamdhip64.so.7 is shared)
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        l.ofxcv_last_error.restype = C.c_char_p
        l.ofxcv_status_string.restype = C.c_char_p
        l.ofxcv_ctx_stream.restype = C.c_void_p
        _lib = l
    return _lib


# every symbol include/ofxcv_hip.h declares (checked by tests/test_abi.py)
INPAINT_NS = 0     # CV_INPAINT_NS
INPAINT_TELEA = 1  # CV_INPAINT_TELEA
OPTFLOW_USE_INITIAL_FLOW = 4      # cv::OPTFLOW_USE_INITIAL_FLOW: `flow` is read as the initial flow
OPTFLOW_FARNEBACK_GAUSSIAN = 256  # cv::OPTFLOW_FARNEBACK_GAUSSIAN: Gaussian window instead of the box

EXPORTS = [
    "ofxcv_device_count", "ofxcv_ctx_create", "ofxcv_ctx_destroy", "ofxcv_last_error", "ofxcv_status_string",
    "ofxcv_ctx_device", "ofxcv_lock_hold", "ofxcv_ctx_stream", "ofxcv_ctx_synchronize", "ofxcv_ctx_set_option", "ofxcv_ctx_get_option", "ofxcv_profile_enable", "ofxcv_profile_read", "ofxcv_to_byte_grayscale", "ofxcv_to_byte_grayscale_batch", "ofxcv_calc_optical_flow_farneback", "ofxcv_calc_optical_flow_farneback_batch", "ofxcv_calc_optical_flow_farneback_batch_rgba",
    "ofxcv_flow_to_rgba", "ofxcv_vectorgen_flow_host", "ofxcv_vectorgen_flows_host", "ofxcv_vectorgen_flows_host_keyed", "ofxcv_host_cache_hits", "ofxcv_host_cache_misses", "ofxcv_host_cache_stats", "ofxcv_host_cache_clear", "ofxcv_host_coalesce_stats", "ofxcv_host_zero_copy_calls", "ofxcv_host_direct_calls", "ofxcv_farneback_plane_pitch", "ofxcv_farneback_col_pairs", "ofxcv_farneback_num_levels",
    "ofxcv_farneback_level_geom", "ofxcv_farneback_pyr_image", "ofxcv_farneback_polyexp",
    "ofxcv_farneback_update_matrices", "ofxcv_farneback_update_flow_blur",
    "ofxcv_inpaint_mask", "ofxcv_inpaint_telea", "ofxcv_inpaint", "ofxcv_inpaint_fallback_count", "ofxcv_inpaint_render_host",
    "ofxcv_pyr_mean_shift_filtering", "ofxcv_segment_render_host",
]


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class _Ordered:
    """Runs a C-ABI call on the context's own compute stream, ordered after the work already queued on
    torch's current stream and before whatever torch queues next (two event waits, skipped when the
    caller already works inside `with torch.cuda.stream(ctx.stream)`)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        import torch
        self.cur = torch.cuda.current_stream()
        self.same = self.cur.cuda_stream == self.ctx.stream.cuda_stream
        if not self.same:
            self.ctx.stream.wait_stream(self.cur)
        return None  # NULL stream argument = the context's compute stream

    def __exit__(self, *exc):
        if not self.same:
            self.cur.wait_stream(self.ctx.stream)
        return False


class Context:
    """One device context (ofxcv_ctx): scratch planes, LUT, staging.  Not thread-safe: one per thread."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().ofxcv_ctx_create(C.c_int(device), C.byref(self._h))
        if rc != OK:
            raise OfxcvError(rc, lib().ofxcv_status_string(rc).decode())
        self.device = device
        import torch
        self.stream = torch.cuda.ExternalStream(lib().ofxcv_ctx_stream(self._h), device=device % max(1, torch.cuda.device_count()))  # (logical -> physical: OFXCV_VIRTUAL_DEVICES)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().ofxcv_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise OfxcvError(rc, lib().ofxcv_last_error(self._h).decode())

    def synchronize(self):
        self._check(lib().ofxcv_ctx_synchronize(self._h, None))

    def lock_hold(self):
        """(nanoseconds, holds): time this context's Farneback calls have held the runtime lock exclusively"""
        ns, n = C.c_long(), C.c_long()
        self._check(lib().ofxcv_lock_hold(self._h, C.byref(ns), C.byref(n)))
        return ns.value, n.value

    def set_option(self, name, value):
        self._check(lib().ofxcv_ctx_set_option(self._h, name.encode(), C.c_int(int(value))))

    def get_option(self, name):
        v = C.c_int()
        self._check(lib().ofxcv_ctx_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def profile_enable(self, on=True):
        """True/1: event pairs around the dominant kernel at level 0; 2: around the carry pre-pass (OpenCV-order mode); 0: off"""
        self._check(lib().ofxcv_profile_enable(self._h, C.c_int(int(on))))

    def profile_read(self, reset=True):
        """(total_ms, launches) of the dominant kernel (level-0 fused iteration) since the last reset."""
        ms, n = C.c_double(), C.c_long()
        self._check(lib().ofxcv_profile_read(self._h, C.byref(ms), C.byref(n), C.c_int(1 if reset else 0)))
        return ms.value, n.value

    def _call(self, fn, *args):
        """fn(ctx, *args, stream=NULL) on the context's compute stream, ordered against torch's current stream."""
        with _Ordered(self):
            self._check(fn(self._h, *args, None))

    # ---- F0 ----
    def to_byte_grayscale(self, src, out=None):
        """src: HxWx{3,4} float32 CUDA tensor (linear RGB[A]) -> HxW uint8 sRGB luma."""
        import torch
        assert src.is_cuda and src.dtype == torch.float32 and src.dim() == 3 and src.stride(2) == 1 and src.stride(1) == src.shape[2]
        h, w, nc = src.shape
        if out is None:
            out = torch.empty((h, w), dtype=torch.uint8, device=src.device)
        self._call(lib().ofxcv_to_byte_grayscale, _ptr(src), C.c_ssize_t(src.stride(0) * 4), C.c_int(nc), C.c_int(w), C.c_int(h),
                   _ptr(out), C.c_ssize_t(out.stride(0)))
        return out

    def to_byte_grayscale_batch(self, srcs, outs):
        """n frames of one size in one launch: srcs HxWx{3,4} float32 CUDA tensors, outs HxW uint8 (written)."""
        import torch
        n = len(srcs)
        assert n >= 1 and len(outs) == n
        h, w, nc = srcs[0].shape
        for s_, o in zip(srcs, outs):
            assert s_.is_cuda and s_.dtype == torch.float32 and tuple(s_.shape) == (h, w, nc) and s_.stride(2) == 1 and s_.stride(1) == nc
            assert o.is_cuda and o.dtype == torch.uint8 and tuple(o.shape) == (h, w) and o.stride(1) == 1
        ps = (C.c_void_p * n)(*[s_.data_ptr() for s_ in srcs])
        po = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
        rs = (C.c_ssize_t * n)(*[s_.stride(0) * 4 for s_ in srcs])
        ro = (C.c_ssize_t * n)(*[o.stride(0) for o in outs])
        self._call(lib().ofxcv_to_byte_grayscale_batch, C.c_int(n), ps, rs, C.c_int(nc), C.c_int(w), C.c_int(h), po, ro)
        return outs

    # ---- F1-F6 ----
    def calc_optical_flow_farneback(self, prev, nxt, flow=None, pyr_scale=0.5, levels=3, winsize=3, iterations=15,
                                    poly_n=5, poly_sigma=1.1, flags=0):
        """Mirror of cv::calcOpticalFlowFarneback: prev/nxt HxW uint8 CUDA tensors -> HxWx2 float32 flow."""
        import torch
        assert prev.is_cuda and nxt.is_cuda and prev.dtype == torch.uint8 and nxt.dtype == torch.uint8
        assert prev.shape == nxt.shape and prev.dim() == 2 and prev.stride(1) == 1 and nxt.stride(1) == 1
        h, w = prev.shape
        if flow is None:
            flow = torch.empty((h, w, 2), dtype=torch.float32, device=prev.device)
        assert flow.shape == (h, w, 2) and flow.dtype == torch.float32 and flow.stride(2) == 1 and flow.stride(1) == 2
        self._call(lib().ofxcv_calc_optical_flow_farneback, _ptr(prev), C.c_size_t(prev.stride(0)), _ptr(nxt),
                   C.c_size_t(nxt.stride(0)), _ptr(flow), C.c_size_t(flow.stride(0) * 4), C.c_int(w), C.c_int(h),
                   C.c_double(pyr_scale), C.c_int(levels), C.c_int(winsize), C.c_int(iterations), C.c_int(poly_n),
                   C.c_double(poly_sigma), C.c_int(flags))
        return flow

    @staticmethod
    def _check_batch(prevs, nxts, flows):
        """the argument checks both batched entry points share: equal list lengths (ctypes would zero-fill a short array), dtype,
        device, shape and contiguity of every image and flow field"""
        import torch
        n = len(prevs)
        assert n == len(nxts) and n >= 1
        h, w = prevs[0].shape
        if flows is None:
            flows = [torch.empty((h, w, 2), dtype=torch.float32, device=prevs[0].device) for _ in range(n)]
        assert len(flows) == n
        for p, q, f in zip(prevs, nxts, flows):
            assert p.is_cuda and q.is_cuda and p.dtype == torch.uint8 and q.dtype == torch.uint8 and p.shape == (h, w) and q.shape == (h, w)
            assert p.stride(1) == 1 and q.stride(1) == 1
            assert f.is_cuda and f.shape == (h, w, 2) and f.dtype == torch.float32 and f.stride(2) == 1 and f.stride(1) == 2
        return flows

    def calc_optical_flow_farneback_batch(self, prevs, nxts, flows=None, pyr_scale=0.5, levels=3, winsize=3, iterations=15,
                                          poly_n=5, poly_sigma=1.1, flags=0):
        """n independent frame pairs of one size in one call (ofxcv_calc_optical_flow_farneback_batch): lists of HxW uint8
        CUDA tensors -> list of HxWx2 float32 flows.  Pairs may share images."""
        flows = self._check_batch(prevs, nxts, flows)
        n = len(prevs)
        h, w = prevs[0].shape
        vp = (C.c_void_p * n)
        sz = (C.c_size_t * n)
        self._call(lib().ofxcv_calc_optical_flow_farneback_batch, C.c_int(n),
                   vp(*[p.data_ptr() for p in prevs]), sz(*[p.stride(0) for p in prevs]),
                   vp(*[q.data_ptr() for q in nxts]), sz(*[q.stride(0) for q in nxts]),
                   vp(*[f.data_ptr() for f in flows]), sz(*[f.stride(0) * 4 for f in flows]),
                   C.c_int(w), C.c_int(h), C.c_double(pyr_scale), C.c_int(levels), C.c_int(winsize), C.c_int(iterations), C.c_int(poly_n),
                   C.c_double(poly_sigma), C.c_int(flags))
        return flows

    def calc_optical_flow_farneback_batch_rgba(self, prevs, nxts, flows, dsts, chan_u_masks, chan_v_masks, rs_x=1.0, rs_y=1.0, pyr_scale=0.5, levels=3,
                                               winsize=3, iterations=15, poly_n=5, poly_sigma=1.1, flags=0):
        """the batched call with F7 fused in (ofxcv_calc_optical_flow_farneback_batch_rgba): pair i also writes flow / render scale
        into the mapped channels of the HxWx4 float32 image dsts[i] (None: no image for that pair)"""
        import torch
        flows = self._check_batch(prevs, nxts, flows)
        n = len(prevs)
        h, w = prevs[0].shape
        assert len(dsts) == n and len(chan_u_masks) == n and len(chan_v_masks) == n
        for d in dsts:
            assert d is None or (d.is_cuda and d.dtype == torch.float32 and d.shape[0] == h and d.shape[1] == w and d.shape[2] == 4 and d.stride(2) == 1 and d.stride(1) == 4)
        vp, sz, pd, un = (C.c_void_p * n), (C.c_size_t * n), (C.c_ssize_t * n), (C.c_uint * n)
        self._call(lib().ofxcv_calc_optical_flow_farneback_batch_rgba, C.c_int(n),
                   vp(*[p.data_ptr() for p in prevs]), sz(*[p.stride(0) for p in prevs]),
                   vp(*[q.data_ptr() for q in nxts]), sz(*[q.stride(0) for q in nxts]),
                   vp(*[f.data_ptr() for f in flows]), sz(*[f.stride(0) * 4 for f in flows]),
                   C.c_int(w), C.c_int(h), C.c_double(pyr_scale), C.c_int(levels), C.c_int(winsize), C.c_int(iterations), C.c_int(poly_n),
                   C.c_double(poly_sigma), C.c_int(flags),
                   vp(*[(d.data_ptr() if d is not None else None) for d in dsts]), pd(*[(d.stride(0) * 4 if d is not None else 0) for d in dsts]),
                   un(*chan_u_masks), un(*chan_v_masks), C.c_double(rs_x), C.c_double(rs_y))
        return flows

    # ---- F7 ----
    def flow_to_rgba(self, flow, dst, chan_u_mask, chan_v_mask, rs_x=1.0, rs_y=1.0):
        import torch
        h, w, _ = flow.shape
        assert dst.shape == (h, w, 4) and dst.dtype == torch.float32 and dst.is_cuda
        self._call(lib().ofxcv_flow_to_rgba, _ptr(flow), C.c_size_t(flow.stride(0) * 4), C.c_int(w), C.c_int(h), _ptr(dst),
                   C.c_ssize_t(dst.stride(0) * 4), C.c_uint(chan_u_mask), C.c_uint(chan_v_mask), C.c_double(rs_x), C.c_double(rs_y))
        return dst

    # ---- whole calcOpticalFlow on host images (numpy arrays) ----
    def vectorgen_flow_host(self, ref, other, dst, chan_u_mask, chan_v_mask, rs_x=1.0, rs_y=1.0, levels=3, iterations=15,
                            poly_n=5, poly_sigma=1.1):
        import numpy as np
        h, w, nc = ref.shape
        assert ref.dtype == np.float32 and other.dtype == np.float32 and dst.dtype == np.float32 and dst.shape == (h, w, 4)
        assert ref.strides[1] == nc * 4 and other.strides[1] == nc * 4 and dst.strides[1] == 16
        self._check(lib().ofxcv_vectorgen_flow_host(
            self._h, C.c_void_p(ref.ctypes.data), C.c_ssize_t(ref.strides[0]), C.c_void_p(other.ctypes.data),
            C.c_ssize_t(other.strides[0]), C.c_int(nc), C.c_int(w), C.c_int(h), C.c_void_p(dst.ctypes.data),
            C.c_ssize_t(dst.strides[0]), C.c_uint(chan_u_mask), C.c_uint(chan_v_mask), C.c_double(rs_x), C.c_double(rs_y),
            C.c_int(levels), C.c_int(iterations), C.c_int(poly_n), C.c_double(poly_sigma)))
        return dst

    def vectorgen_flows_host(self, ref, fwd, bwd, dst, fwd_u, fwd_v, bwd_u, bwd_v, rs_x=1.0, rs_y=1.0, levels=3, iterations=15,
                             poly_n=5, poly_sigma=1.1, keys=None):
        """Both directions of a VectorGenerator output frame in one call; fwd or bwd may be None.  keys = (ref, fwd, bwd) names
        of the frames' pixels (str or None): ofxcv_vectorgen_flows_host_keyed."""
        import numpy as np
        h, w, nc = ref.shape
        for a in (ref, fwd, bwd):
            assert a is None or (a.dtype == np.float32 and a.shape == ref.shape and a.strides[1] == nc * 4)
        assert dst.dtype == np.float32 and dst.shape == (h, w, 4) and dst.strides[1] == 16
        ptr = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None
        rb = lambda a: C.c_ssize_t(a.strides[0] if a is not None else 0)
        args = (self._h, ptr(ref), rb(ref), ptr(fwd), rb(fwd), ptr(bwd), rb(bwd), C.c_int(nc), C.c_int(w), C.c_int(h), ptr(dst), rb(dst),
                C.c_uint(fwd_u), C.c_uint(fwd_v), C.c_uint(bwd_u), C.c_uint(bwd_v), C.c_double(rs_x), C.c_double(rs_y),
                C.c_int(levels), C.c_int(iterations), C.c_int(poly_n), C.c_double(poly_sigma))
        if keys is None:
            self._check(lib().ofxcv_vectorgen_flows_host(*args))
        else:
            kk = [C.c_char_p(k.encode()) if k else C.c_char_p(None) for k in keys]
            self._check(lib().ofxcv_vectorgen_flows_host_keyed(*args, *kk))
        return dst

    def farneback_col_pairs(self, width, height, n):
        """pairs of a batched call of n that walk level 0 in the column-owning form (ofxcv_farneback_col_pairs)"""
        return int(lib().ofxcv_farneback_col_pairs(self._h, C.c_int(width), C.c_int(height), C.c_int(n)))

    def host_cache_hits(self):
        lib().ofxcv_host_cache_hits.restype = C.c_long
        return int(lib().ofxcv_host_cache_hits(self._h))

    def host_cache_misses(self):
        lib().ofxcv_host_cache_misses.restype = C.c_long
        return int(lib().ofxcv_host_cache_misses(self._h))

    def host_cache_stats(self):
        """(bytes, frames) held by the cache of this context's device"""
        b, n = C.c_size_t(), C.c_int()
        self._check(lib().ofxcv_host_cache_stats(self._h, C.byref(b), C.byref(n)))
        return b.value, n.value

    def host_coalesce_stats(self):
        """(calls, pairs, batch_pairs): host-image calls of this context served by the device's submission queue, their pairs, and the
        summed size (pairs) of the batched calls they rode in"""
        a, b, c = C.c_long(), C.c_long(), C.c_long()
        self._check(lib().ofxcv_host_coalesce_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def host_cache_clear(self):
        self._check(lib().ofxcv_host_cache_clear(self._h))

    # ---- inpaint ----
    def inpaint_mask(self, rgba, dilate_iters=1):
        """rgba: HxWx4 uint8 CUDA tensor -> HxW uint8 hole mask (255 = hole)."""
        import torch
        h, w, _ = rgba.shape
        mask = torch.empty((h, w), dtype=torch.uint8, device=rgba.device)
        self._call(lib().ofxcv_inpaint_mask, _ptr(rgba), C.c_ssize_t(rgba.stride(0)), C.c_int(w), C.c_int(h), C.c_int(dilate_iters),
                   _ptr(mask), C.c_ssize_t(w))
        return mask

    def inpaint(self, src, mask, radius=3.0, method=INPAINT_TELEA, maps=False):
        """Mirror of cvInpaint(src, mask, dst, radius, method); src HxWx{3,4} uint8, mask HxW uint8."""
        import torch
        h, w, cn = src.shape
        dst = torch.empty_like(src)
        t = torch.empty((h + 2, w + 2), dtype=torch.float32, device=src.device) if maps else None
        order = torch.empty((h, w), dtype=torch.int32, device=src.device) if maps else None
        self._call(lib().ofxcv_inpaint, _ptr(src), C.c_ssize_t(src.stride(0)), C.c_int(cn), _ptr(mask), C.c_ssize_t(mask.stride(0)),
                   C.c_int(w), C.c_int(h), C.c_double(radius), C.c_int(method), _ptr(dst), C.c_ssize_t(dst.stride(0)),
                   _ptr(t) if maps else None, _ptr(order) if maps else None)
        return (dst, t, order) if maps else dst

    def host_zero_copy_calls(self):
        lib().ofxcv_host_zero_copy_calls.restype = C.c_long
        return int(lib().ofxcv_host_zero_copy_calls(self._h))

    def host_direct_calls(self):
        lib().ofxcv_host_direct_calls.restype = C.c_long
        return int(lib().ofxcv_host_direct_calls(self._h))

    def inpaint_fallback_count(self):
        lib().ofxcv_inpaint_fallback_count.restype = C.c_long
        return int(lib().ofxcv_inpaint_fallback_count(self._h))

    def inpaint_telea(self, src, mask, radius=3.0, maps=False):
        return self.inpaint(src, mask, radius, INPAINT_TELEA, maps)

    def inpaint_render_host(self, rgba, radius=3.0, dilation=1.0, want_mask=False):
        """Whole inpaint render() body on a host image (numpy HxWx4 uint8)."""
        import numpy as np
        h, w, _ = rgba.shape
        assert rgba.dtype == np.uint8 and rgba.strides[1] == 4 and rgba.strides[2] == 1
        dst = np.empty((h, w, 4), np.uint8)
        mask = np.empty((h, w), np.uint8) if want_mask else None
        self._check(lib().ofxcv_inpaint_render_host(self._h, C.c_void_p(rgba.ctypes.data), C.c_ssize_t(rgba.strides[0]), C.c_int(w),
                                                    C.c_int(h), C.c_double(radius), C.c_double(dilation), C.c_void_p(dst.ctypes.data),
                                                    C.c_ssize_t(w * 4), C.c_void_p(mask.ctypes.data) if want_mask else None))
        return (dst, mask) if want_mask else dst

    # ---- segment ----
    def pyr_mean_shift_filtering(self, src, sp=10.0, sr=20.0, max_level=2, max_iter=5, eps=1.0):
        """Mirror of cv::pyrMeanShiftFiltering; src HxWx{3,4} uint8 CUDA tensor."""
        import torch
        h, w, cn = src.shape
        dst = torch.empty_like(src)
        self._call(lib().ofxcv_pyr_mean_shift_filtering, _ptr(src), C.c_ssize_t(src.stride(0)), C.c_int(cn), C.c_int(w), C.c_int(h),
                   C.c_double(sp), C.c_double(sr), C.c_int(max_level), C.c_int(max_iter), C.c_double(eps), _ptr(dst), C.c_ssize_t(dst.stride(0)))
        return dst

    def segment_render_host(self, rgba, sp=10.0, sr=20.0, max_level=2):
        import numpy as np
        h, w, _ = rgba.shape
        assert rgba.dtype == np.uint8 and rgba.strides[1] == 4
        dst = np.empty((h, w, 4), np.uint8)
        self._check(lib().ofxcv_segment_render_host(self._h, C.c_void_p(rgba.ctypes.data), C.c_ssize_t(rgba.strides[0]), C.c_int(w), C.c_int(h),
                                                    C.c_double(sp), C.c_double(sr), C.c_int(max_level), C.c_void_p(dst.ctypes.data),
                                                    C.c_ssize_t(w * 4)))
        return dst

    # ---- stage-level (parity tests) ----
    def farneback_pyr_image(self, img, lw, lh, sigma, ksize):
        import torch
        h, w = img.shape
        I = torch.empty((lh, lw), dtype=torch.float32, device=img.device)
        self._call(lib().ofxcv_farneback_pyr_image, _ptr(img), C.c_size_t(img.stride(0)), C.c_int(w), C.c_int(h), C.c_int(lw),
                   C.c_int(lh), C.c_double(sigma), C.c_int(ksize), _ptr(I))
        return I

    def farneback_polyexp(self, I, poly_n=5, poly_sigma=1.1):
        """I: HxW float32 -> 5 x H x pitch float32 planes (use planes_to_hwc to compare with the oracle)."""
        import torch
        h, w = I.shape
        I = I.contiguous()
        R = torch.zeros((5, h, plane_pitch(w)), dtype=torch.float32, device=I.device)
        self._call(lib().ofxcv_farneback_polyexp, _ptr(I), C.c_int(w), C.c_int(h), _ptr(R), C.c_int(poly_n), C.c_double(poly_sigma))
        return R

    def farneback_update_matrices(self, R0, R1, flow):
        import torch
        h, w, _ = flow.shape
        M = torch.zeros_like(R0)
        self._call(lib().ofxcv_farneback_update_matrices, _ptr(R0), _ptr(R1), _ptr(flow), C.c_size_t(flow.stride(0) * 4), C.c_int(w),
                   C.c_int(h), _ptr(M))
        return M

    def farneback_update_flow_blur(self, R0, R1, M, w, winsize=3, update=True, flow=None, Mo=None):
        """returns (flow HxWx2, M_out planes or None)"""
        import torch
        h = M.shape[1]
        if flow is None:
            flow = torch.empty((h, w, 2), dtype=torch.float32, device=M.device)
        if update and Mo is None:
            Mo = torch.zeros_like(M)
        self._call(lib().ofxcv_farneback_update_flow_blur, _ptr(R0), _ptr(R1), _ptr(M), _ptr(Mo) if update else None, _ptr(flow),
                   C.c_size_t(w * 8), C.c_int(w), C.c_int(h), C.c_int(winsize), C.c_int(1 if update else 0))
        return flow, Mo


def plane_pitch(w):
    return int(lib().ofxcv_farneback_plane_pitch(C.c_int(w)))


def farneback_num_levels(w, h, pyr_scale=0.5, levels=3):
    return int(lib().ofxcv_farneback_num_levels(C.c_int(w), C.c_int(h), C.c_double(pyr_scale), C.c_int(levels)))


def farneback_level_geom(w, h, pyr_scale, k):
    lw, lh, ks = C.c_int(), C.c_int(), C.c_int()
    sg = C.c_double()
    rc = lib().ofxcv_farneback_level_geom(C.c_int(w), C.c_int(h), C.c_double(pyr_scale), C.c_int(k), C.byref(lw), C.byref(lh),
                                          C.byref(sg), C.byref(ks))
    if rc != OK:
        raise OfxcvError(rc, "level_geom")
    return lw.value, lh.value, sg.value, ks.value


def hwc_to_planes(a, pitch=None):
    """HxWx5 (torch) -> 5xHxpitch planes, zero padded."""
    import torch
    h, w, c = a.shape
    pitch = pitch or plane_pitch(w)
    p = torch.zeros((c, h, pitch), dtype=a.dtype, device=a.device)
    p[:, :, :w] = a.permute(2, 0, 1)
    return p


def planes_to_hwc(p, w):
    return p[:, :, :w].permute(1, 2, 0).contiguous()
