"""Error behaviour of the C ABI and concurrency of the drop-in boundary (GPU)."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_c_abi_rejects_bad_arguments(ofxcv, gpu_ctx):
    import torch
    lib = ofxcv.lib()
    h = gpu_ctx._h
    g = torch.zeros((48, 64), dtype=torch.uint8, device="cuda")
    fl = torch.zeros((48, 64, 2), dtype=torch.float32, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    call = lambda **kw: lib.ofxcv_calc_optical_flow_farneback(
        h, kw.get("prev", P(g)), C.c_size_t(kw.get("pstep", 64)), P(g), C.c_size_t(64), kw.get("flow", P(fl)), C.c_size_t(kw.get("fstep", 512)),
        C.c_int(kw.get("w", 64)), C.c_int(kw.get("h", 48)), C.c_double(kw.get("ps", 0.5)), C.c_int(3), C.c_int(kw.get("win", 3)),
        C.c_int(kw.get("it", 15)), C.c_int(kw.get("n", 5)), C.c_double(1.1), C.c_int(kw.get("flags", 0)), None)
    assert call() == 0
    assert call(prev=None) == -1 and call(flow=None) == -1                 # NULL pointers
    assert call(w=0) == -1 and call(h=-3) == -1                            # sizes
    assert call(pstep=32) == -1 and call(fstep=100) == -1                  # strides smaller than a row / unaligned
    assert call(ps=1.0) == -1 and call(ps=0.0) == -1 and call(it=0) == -1 and call(win=4) == -1
    assert call(flags=1) == -4 and call(flags=4 | 256 | 8) == -4             # only USE_INITIAL_FLOW (4) and FARNEBACK_GAUSSIAN (256) exist
    assert b"flags" in lib.ofxcv_last_error(h)
    assert call(n=99) == -4                                                # poly_n beyond the coefficient tables
    assert lib.ofxcv_ctx_set_option(h, b"no.such.option", 1) == -1
    assert lib.ofxcv_to_byte_grayscale(h, P(fl), C.c_ssize_t(512), C.c_int(2), 64, 48, P(g), C.c_ssize_t(64), None) == -4   # 2 components
    assert lib.ofxcv_inpaint_telea(h, P(g), C.c_ssize_t(64), C.c_int(1), P(g), C.c_ssize_t(64), 16, 16, C.c_double(3.0), P(g), C.c_ssize_t(64), None, None, None) == -1
    assert lib.ofxcv_pyr_mean_shift_filtering(h, P(g), C.c_ssize_t(64), 3, 16, 16, C.c_double(10), C.c_double(20), 9, 5, C.c_double(1), P(fl), C.c_ssize_t(64), None) == -1
    assert call() == 0                                                      # the context is still usable after errors


def test_other_window_sizes_use_the_generic_kernel(oracle, direct_ctx):
    gpu_ctx = direct_ctx
    from openfx_opencv_amd import synth
    a, b = synth.flow_pair(160, 120)
    ga, gb = oracle.to_byte_grayscale(a), oracle.to_byte_grayscale(b)
    for win in (1, 5):
        got = gpu_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb), winsize=win, iterations=4).cpu().numpy()
        ref = oracle.calc_optical_flow_farneback(ga, gb, winsize=win, iterations=4, blur_mode=oracle.BLUR_DIRECT)
        assert np.array_equal(ref, got), win
    # odd / even iteration counts exercise the pair + single launch mix of the fused kernel
    for it in (1, 2, 3, 6):
        got = gpu_ctx.calc_optical_flow_farneback(_dev(ga), _dev(gb), iterations=it).cpu().numpy()
        assert np.array_equal(oracle.calc_optical_flow_farneback(ga, gb, iterations=it, blur_mode=oracle.BLUR_DIRECT), got), it


def test_graph_replay_and_eager_agree(ofxcv, oracle):
    import torch
    from openfx_opencv_amd import synth
    a, b = synth.flow_pair(320, 240)
    ga, gb = _dev(oracle.to_byte_grayscale(a)), _dev(oracle.to_byte_grayscale(b))
    c1, c2 = ofxcv.Context(0), ofxcv.Context(0)
    c1.set_option("farneback.graph", 1)      # captured once, replayed (the default of rounds 1-5)
    c2.set_option("farneback.graph", 0)      # eager launches (the default)
    assert ofxcv.Context(0).get_option("farneback.graph") == 0
    f_out = torch.empty((240, 320, 2), device="cuda")
    first = c1.calc_optical_flow_farneback(ga, gb, f_out).clone()          # capture + launch
    again = c1.calc_optical_flow_farneback(ga, gb, f_out).clone()          # replay of the cached graph
    eager = c2.calc_optical_flow_farneback(ga, gb)
    assert torch.equal(first, again) and torch.equal(first, eager)
    other = c1.calc_optical_flow_farneback(gb, ga, f_out).clone()          # different pointers: a new graph, not a stale replay
    assert torch.equal(other, c2.calc_optical_flow_farneback(gb, ga)) and not torch.equal(other, first)
    c1.close()
    c2.close()


def test_every_kernel_variant_gives_the_same_flow(ofxcv, oracle):
    """fused / unfused iterations, prep stream on / off, fused / two-pass pyramid: identical results."""
    import torch
    from openfx_opencv_amd import synth
    a, b = synth.flow_pair(333, 257)
    ga, gb = _dev(oracle.to_byte_grayscale(a)), _dev(oracle.to_byte_grayscale(b))
    base = None
    for opts in [{}, {"farneback.graph": 1}, {"farneback.fused_pyramid": 0}, {"farneback.fused_pyramid": 2}, {"farneback.fused_pyramid": 3},
                 {"farneback.graph": 1, "farneback.fused_pyramid": 0}]:
        c = ofxcv.Context(0)
        c.set_option("farneback.opencv_rounding", 0)   # the direct-window kernels (fused pairs exist only there)
        for k, v in opts.items():
            c.set_option(k, v)
        f = c.calc_optical_flow_farneback(ga, gb).clone()
        c.close()
        if base is None:
            base = f
            assert np.array_equal(f.cpu().numpy(), oracle.calc_optical_flow_farneback(ga.cpu().numpy(), gb.cpu().numpy(), blur_mode=oracle.BLUR_DIRECT))
        else:
            assert torch.equal(base, f), opts


def test_concurrent_renders_on_one_instance(oracle):
    """VectorGenerator is eRenderFullySafe (VectorGenerator.cpp:108): the host may call render() on one instance from
    several threads at different times.  Four threads render four frames concurrently through the OFX boundary."""
    import os
    import subprocess
    import test_ofx_boundary as tb
    Plugin = tb.Plugin
    subprocess.check_call(["make", "-s", "-C", os.path.join(tb.ROOT, "tests", "mock_host")])
    import torch  # noqa: F401
    h = C.CDLL(os.path.join(tb.ROOT, "tests", "mock_host", "libmockhost.so"))
    h.mh_open.restype = C.c_void_p
    h.mh_create_instance.restype = C.c_void_p
    h.mh_last_message.restype = C.c_char_p
    h.mh_plugin_identifier.restype = C.c_char_p
    from openfx_opencv_amd import synth
    w, hh = 256, 192
    frames = [synth.flow_pair(w, hh, seed=100 + i)[0] for i in range(6)]    # frames t = 0..5
    pl = Plugin(h, "VectorGenerator")
    inst = pl.instance()
    outs = {}
    for t, fr in enumerate(frames):
        pl.set_image(inst, "Source", float(t), fr, "OfxBitDepthFloat")
    for t in range(1, 5):
        outs[t] = np.zeros((hh, w, 4), np.float32)
        pl.set_image(inst, "Output", float(t), outs[t], "OfxBitDepthFloat")
    status = {}

    def work(t):
        status[t] = pl.render(inst, float(t), w, hh)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(1, 5)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert all(status[t] == 0 for t in range(1, 5)), status
    gray = [oracle.to_byte_grayscale(f) for f in frames]
    for t in range(1, 5):
        # the plugin runs the library default: the OpenCV-order window, every sample within 1e-4 of the faithful oracle
        fwd = oracle.calc_optical_flow_farneback(gray[t], gray[t + 1], blur_mode=oracle.BLUR_FAITHFUL)
        bwd = oracle.calc_optical_flow_farneback(gray[t], gray[t - 1], blur_mode=oracle.BLUR_FAITHFUL)
        for got, ref in ((outs[t][..., :2], fwd), (outs[t][..., 2:], bwd)):
            assert (np.abs(got - ref) <= 1e-4 * np.maximum(1, np.abs(ref))).all(), t
    assert h.mh_clip_balance(inst, b"Source") == 0 and h.mh_clip_balance(inst, b"Output") == 0
    pl.destroy(inst)


def test_two_direction_host_call_equals_two_single_calls(ofxcv, oracle):
    """ofxcv_vectorgen_flows_host (forward + backward in one call, shared reference upload) == the two single calls."""
    from openfx_opencv_amd import synth
    w, h = 352, 264
    prev, ref = synth.flow_pair(w, h, seed=11)
    _, nxt = synth.flow_pair(w, h, seed=12)
    c = ofxcv.Context(0)
    a = np.full((h, w, 4), 7.0, np.float32)
    c.vectorgen_flow_host(ref, nxt, a, 0b0001, 0b0010)
    c.vectorgen_flow_host(ref, prev, a, 0b0100, 0b1000)
    b = np.full((h, w, 4), 7.0, np.float32)
    c.vectorgen_flows_host(ref, nxt, prev, b, 0b0001, 0b0010, 0b0100, 0b1000)
    assert np.array_equal(a, b)
    # one direction only, partial channel map, render scale
    a2 = np.full((h, w, 4), 7.0, np.float32)
    b2 = np.full((h, w, 4), 7.0, np.float32)
    c.vectorgen_flow_host(ref, prev, a2, 0b0100, 0, 0.5, 0.5)
    c.vectorgen_flows_host(ref, None, prev, b2, 0, 0, 0b0100, 0, 0.5, 0.5)
    assert np.array_equal(a2, b2) and np.all(b2[..., [0, 1, 3]] == 7.0)
    c.close()


def test_bottom_up_host_images(ofxcv, oracle):
    """OFX images may have negative row strides (bottom-up storage): all three host entry points accept them."""
    from openfx_opencv_amd import synth
    c = ofxcv.Context(0)
    # inpaint / segment: uint8 RGBA, source and destination bottom-up
    fr = synth.inpaint_frame(96, 72, n_holes=4)
    phys = np.ascontiguousarray(fr[::-1])          # rows stored bottom-up ...
    view = phys[::-1]                              # ... and addressed through a negative stride
    assert view.strides[0] < 0 and np.array_equal(view, fr)
    out_phys = np.zeros_like(phys)
    out = out_phys[::-1]
    lib = ofxcv.lib()
    rc = lib.ofxcv_inpaint_render_host(c._h, C.c_void_p(view.ctypes.data), C.c_ssize_t(view.strides[0]), C.c_int(96), C.c_int(72),
                                       C.c_double(3.0), C.c_double(1.0), C.c_void_p(out.ctypes.data), C.c_ssize_t(out.strides[0]), None)
    assert rc == 0 and np.array_equal(out, oracle.inpaint_render(fr, 3.0, 1.0))
    out_phys[:] = 0
    rc = lib.ofxcv_segment_render_host(c._h, C.c_void_p(view.ctypes.data), C.c_ssize_t(view.strides[0]), C.c_int(96), C.c_int(72),
                                       C.c_double(10.0), C.c_double(20.0), C.c_int(2), C.c_void_p(out.ctypes.data), C.c_ssize_t(out.strides[0]))
    ref = oracle.pyr_mean_shift(np.ascontiguousarray(fr[..., :3]), 10.0, 20.0, 2)
    assert rc == 0 and np.array_equal(out[..., :3], ref) and np.all(out[..., 3] == 255)
    # VectorGenerator: float RGBA frames bottom-up
    a, b = synth.flow_pair(128, 96)
    ap, bp = np.ascontiguousarray(a[::-1])[::-1], np.ascontiguousarray(b[::-1])[::-1]
    d1 = np.zeros((96, 128, 4), np.float32)
    d2p = np.zeros((96, 128, 4), np.float32)
    d2 = d2p[::-1]
    c.vectorgen_flow_host(a, b, d1, 1, 2)
    c.vectorgen_flow_host(ap, bp, d2, 1, 2)
    assert np.array_equal(d1, d2)
    c.close()


def test_host_path_forms_agree(ofxcv):
    """ofxcv_vectorgen_flows_host moves the host's frames in one of three ways: copies straight from / into the host's pageable
    images (default; with all four destination channels mapped a kernel composes the RGBA image in HBM and one copy brings it
    back), the host's buffers registered for the call (option host.register = 2: the copy engine reads the frames in place and a
    kernel stores whole pixels into the host image), or staged through the pinned ring (host.register = 0; also what bottom-up
    images get).  Same pixels every way; buffers that the host frees and re-allocates between calls are fine."""
    from openfx_opencv_amd import synth
    w, h = 200, 120
    direct, zc, ring = ofxcv.Context(0), ofxcv.Context(0), ofxcv.Context(0)
    zc.set_option("host.register", 2)
    ring.set_option("host.register", 0)
    n_reg = n_dir = 0
    for rep in range(3):                       # fresh numpy buffers every time: the allocator hands the same addresses again
        ref, nxt = synth.flow_pair(w, h, seed=5 + rep)
        prev, _ = synth.flow_pair(w, h, seed=16 + rep)
        for fu, fv, bu, bv, rx, ry, registered in [(1, 2, 4, 8, 1.0, 1.0, True), (4, 8, 1, 2, 0.5, 0.25, True), (3, 12, 0, 0, 1.0, 2.0, True),
                                                   (1, 0, 0, 8, 1.0, 1.0, False), (1, 2, 2, 4, 1.0, 1.0, False)]:
            outs = []
            for c in (direct, zc, ring):
                o = np.full((h, w, 4), -3.0, np.float32)
                c.vectorgen_flows_host(ref, nxt, prev, o, fu, fv, bu, bv, rx, ry)
                outs.append(o)
            assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[2]), (rep, fu, fv, bu, bv)
            n_reg += registered
            n_dir += 1
            assert zc.host_zero_copy_calls() == n_reg, (rep, fu, fv, bu, bv)
            assert direct.host_direct_calls() == n_dir and zc.host_direct_calls() == n_dir - n_reg
    assert ring.host_zero_copy_calls() == 0 and ring.host_direct_calls() == 0 and direct.host_zero_copy_calls() == 0
    # padded destination / source rows and RGB sources, one direction mapped to all four channels
    pad = np.full((h, w + 12, 4), 5.0, np.float32)
    a, b, d0 = pad.copy(), pad.copy(), pad.copy()
    ref3, nxt3 = np.ascontiguousarray(ref[..., :3]), np.ascontiguousarray(nxt[..., :3])
    zc.vectorgen_flow_host(ref3, nxt3, a[:, :w], 0b0101, 0b1010)
    ring.vectorgen_flow_host(ref3, nxt3, b[:, :w], 0b0101, 0b1010)
    direct.vectorgen_flow_host(ref3, nxt3, d0[:, :w], 0b0101, 0b1010)
    assert np.array_equal(a, b) and np.array_equal(d0, b) and (a[:, w:] == 5.0).all() and zc.host_zero_copy_calls() == n_reg + 1
    # padded source rows (a view into a wider image)
    wide = np.zeros((h, w + 7, 4), np.float32)
    wide[:, :w] = ref
    e0, e1 = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    direct.vectorgen_flow_host(wide[:, :w], nxt, e0, 0b0101, 0b1010)
    ring.vectorgen_flow_host(ref, nxt, e1, 0b0101, 0b1010)
    assert np.array_equal(e0, e1)
    # bottom-up images cannot be addressed as one ascending range: the ring serves them
    refp, nxtp = np.ascontiguousarray(ref[::-1])[::-1], np.ascontiguousarray(nxt[::-1])[::-1]    # same images, stored bottom-up
    for c in (zc, direct):
        n0 = c.host_direct_calls()
        cp = np.zeros((h, w, 4), np.float32)
        c.vectorgen_flow_host(refp, nxtp, cp[::-1], 0b0101, 0b1010)
        assert np.array_equal(cp[::-1], e1) and c.host_direct_calls() == n0
    assert zc.host_zero_copy_calls() == n_reg + 1
    for c in (direct, zc, ring):
        c.close()


def test_host_path_split_form_gives_the_same_frame(ofxcv):
    """With two directions the forward pair can run as a single-pair call while the third frame is still being uploaded
    (option host.split = 1; 2 = only while no other host-image call is in flight, the default) instead of one batched call after
    the last upload (0): the same output frame, through the direct copies and through the pinned ring, and the counter tells
    which form ran."""
    from openfx_opencv_amd import synth
    w, h = 333, 200
    ref, nxt = synth.flow_pair(w, h, seed=7)
    prev, _ = synth.flow_pair(w, h, seed=19)
    outs = {}
    for reg in (1, 0):
        for split in (0, 1, 2):
            c = ofxcv.Context(0)
            c.set_option("host.register", reg)
            c.set_option("host.split", split)
            for fu, fv, bu, bv in ((1, 2, 4, 8), (1, 0, 0, 8)):
                o = np.full((h, w, 4), -3.0, np.float32)
                c.vectorgen_flows_host(ref, nxt, prev, o, fu, fv, bu, bv)
                outs[(reg, split, fu)] = o
            assert c.get_option("host.split_calls") == (0 if split == 0 else 2), (reg, split)
            one = np.zeros((h, w, 4), np.float32)       # one direction: nothing to split
            c.vectorgen_flow_host(ref, nxt, one, 0b0101, 0b1010)
            assert c.get_option("host.split_calls") == (0 if split == 0 else 2)
            c.close()
    for key, o in outs.items():
        assert np.array_equal(o, outs[(1, 0, key[2])]), key


def test_virtual_devices_have_their_own_caches_and_locks(ofxcv, monkeypatch):
    """OFXCV_VIRTUAL_DEVICES=N: N logical devices over the one physical GPU -- what is per device in a multi-GPU
    host process (contexts, the cache of named frames, with OFXCV_LOCK_PER_DEVICE the runtime lock) runs here as on an N-GPU node: a frame
    named on logical device 0 is NOT found on device 1, results are the same on every device, and the lock-hold counter of a context counts
    its graph launches."""
    from openfx_opencv_amd import synth
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("written for the one-GPU box")
    monkeypatch.setenv("OFXCV_VIRTUAL_DEVICES", "3")
    assert ofxcv.lib().ofxcv_device_count() == 3
    w, h = 256, 144
    seq = [synth.flow_pair(w, h, seed=70 + i)[0] for i in range(3)]
    ctxs = [ofxcv.Context(d) for d in range(3)]
    for c in ctxs:
        c.set_option("farneback.graph", 1)   # (the lock-hold counter counts graph launches; the default, eager launches, holds no lock)
    with pytest.raises(ofxcv.OfxcvError):
        ofxcv.Context(3)
    for c in ctxs:
        c.host_cache_clear()
    outs = []
    for c in ctxs:
        o = np.zeros((h, w, 4), np.float32)
        c.vectorgen_flows_host(seq[1], seq[2], seq[0], o, 1, 2, 4, 8, keys=("vd:1", "vd:2", "vd:0"))
        outs.append(o)
        assert (c.host_cache_misses(), c.host_cache_hits()) == (3, 0)   # every logical device uploads for itself
        assert c.host_cache_stats()[1] == 3
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    o = np.zeros((h, w, 4), np.float32)
    ctxs[1].vectorgen_flows_host(seq[1], seq[2], seq[0], o, 1, 2, 4, 8, keys=("vd:1", "vd:2", "vd:0"))
    assert ctxs[1].host_cache_hits() == 3 and np.array_equal(o, outs[0])
    ns, holds = ctxs[1].lock_hold()
    assert holds >= 2 and ns > 0
    for c in ctxs:
        c.host_cache_clear()
        c.close()
    monkeypatch.delenv("OFXCV_VIRTUAL_DEVICES")
    assert ofxcv.lib().ofxcv_device_count() == 1


def test_named_frames_stay_on_the_device(ofxcv):
    """ofxcv_vectorgen_flows_host_keyed: a frame whose name was seen before (same geometry, same device) is neither uploaded nor
    converted again -- rendering a sequence in order uploads ONE frame per output frame instead of three -- and the output is
    the unnamed call's, pixel for pixel.  The cache is per device and shared by contexts; a new name for new pixels gives the
    new frame; the budget evicts least-recently-used frames; unnamed / partly named calls work as before."""
    from openfx_opencv_amd import synth
    w, h = 256, 144
    seq = [synth.flow_pair(w, h, seed=40 + i)[0] for i in range(6)]
    plain, a, b = ofxcv.Context(0), ofxcv.Context(0), ofxcv.Context(0)
    a.host_cache_clear()
    want = {}
    for t in range(1, 5):
        o = np.zeros((h, w, 4), np.float32)
        plain.vectorgen_flows_host(seq[t], seq[t + 1], seq[t - 1], o, 1, 2, 4, 8)
        want[t] = o
    name = lambda t, gen=0: "clipA:%d:%d" % (t, gen)
    for c, frames in ((a, (1, 2)), (b, (3, 4))):          # two contexts (two render threads) share the device's cache
        for t in frames:
            o = np.full((h, w, 4), 7.0, np.float32)
            c.vectorgen_flows_host(seq[t], seq[t + 1], seq[t - 1], o, 1, 2, 4, 8, keys=(name(t), name(t + 1), name(t - 1)))
            assert np.array_equal(o, want[t]), t
    assert (a.host_cache_misses(), a.host_cache_hits()) == (4, 2)     # t=1: three uploads; t=2: only frame 3
    assert (b.host_cache_misses(), b.host_cache_hits()) == (2, 4)     # t=3: frame 4, t=4: frame 5 -- the rest came from `a`'s calls
    nbytes, nframes = a.host_cache_stats()
    assert nframes == 6 and nbytes == 6 * 256 * h                     # gray rows padded to 256 bytes
    # all three on the device: one batched call, no upload at all
    o = np.zeros((h, w, 4), np.float32)
    a.vectorgen_flows_host(seq[2], seq[3], seq[1], o, 1, 2, 4, 8, keys=(name(2), name(3), name(1)))
    assert np.array_equal(o, want[2]) and a.host_cache_hits() == 5
    # the pixels of frame 3 change and so does its name: the new frame is used (the old entry just ages)
    new3 = synth.flow_pair(w, h, seed=99)[0]
    ref = np.zeros((h, w, 4), np.float32)
    plain.vectorgen_flows_host(seq[2], new3, seq[1], ref, 1, 2, 4, 8)
    a.vectorgen_flows_host(seq[2], new3, seq[1], o, 1, 2, 4, 8, keys=(name(2), name(3, 1), name(1)))
    assert np.array_equal(o, ref) and not np.array_equal(o, want[2])
    # partly named, one direction, the same frame twice in a call
    a.vectorgen_flows_host(seq[2], seq[3], seq[1], o, 1, 2, 4, 8, keys=(None, name(3), ""))
    assert np.array_equal(o, want[2])
    one, one_ref = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    plain.vectorgen_flows_host(seq[2], seq[3], None, one_ref, 1, 2, 0, 0)
    a.vectorgen_flows_host(seq[2], seq[3], None, one, 1, 2, 0, 0, keys=(name(2), name(3), None))
    assert np.array_equal(one, one_ref)
    plain.vectorgen_flows_host(seq[2], seq[2], seq[2], ref, 1, 2, 4, 8)
    a.vectorgen_flows_host(seq[2], seq[2], seq[2], o, 1, 2, 4, 8, keys=(name(2), name(2), name(2)))
    assert np.array_equal(o, ref)
    # a budget of three frames (rounded down to whole MB: use frames of >= 1/3 MB): least-recently-used entries go
    a.host_cache_clear()
    assert a.host_cache_stats() == (0, 0)
    W2, H2 = 1024, 400                                                # 400 KB of gray per frame
    big = [synth.flow_pair(W2, H2, seed=60 + i)[0] for i in range(5)]
    small = ofxcv.Context(0)
    small.set_option("host.cache_mb", 1)                              # two frames fit, the third of a call does not
    ref = np.zeros((H2, W2, 4), np.float32)
    o = np.zeros((H2, W2, 4), np.float32)
    for t in (1, 2, 3):
        plain.vectorgen_flows_host(big[t], big[t + 1], big[t - 1], ref, 1, 2, 4, 8)
        small.vectorgen_flows_host(big[t], big[t + 1], big[t - 1], o, 1, 2, 4, 8, keys=("b%d" % t, "b%d" % (t + 1), "b%d" % (t - 1)))
        assert np.array_equal(o, ref), t
        assert small.host_cache_stats()[0] <= 1 << 20
    off = ofxcv.Context(0)
    off.set_option("host.cache_mb", 0)
    off.vectorgen_flows_host(big[1], big[2], big[0], o, 1, 2, 4, 8, keys=("b1", "b2", "b0"))
    assert off.host_cache_hits() == 0 and off.host_cache_misses() == 0
    for c in (plain, a, b, small, off):
        c.close()


@pytest.mark.timeout(300)
def test_named_frames_from_several_threads(ofxcv):
    """Four render threads, one context each, render neighbouring output frames of one sequence at the same time, again and
    again: they find, fill and wait for each other's frames in the device's cache; every output frame is the unnamed call's."""
    import threading
    from openfx_opencv_amd import synth
    w, h = 320, 200
    n = 8
    seq = [synth.flow_pair(w, h, seed=70 + i)[0] for i in range(n + 2)]
    plain = ofxcv.Context(0)
    plain.host_cache_clear()
    want = []
    for t in range(1, n + 1):
        o = np.zeros((h, w, 4), np.float32)
        plain.vectorgen_flows_host(seq[t], seq[t + 1], seq[t - 1], o, 1, 2, 4, 8)
        want.append(o)
    errors, counts = [], []

    def work(k):
        try:
            c = ofxcv.Context(0)
            c.set_option("host.cache_mb", 1)       # small enough that entries are evicted while other threads run
            for rep in range(3):
                for t in range(1 + k, n + 1, 4):
                    o = np.full((h, w, 4), -1.0, np.float32)
                    c.vectorgen_flows_host(seq[t], seq[t + 1], seq[t - 1], o, 1, 2, 4, 8, keys=("s%d" % t, "s%d" % (t + 1), "s%d" % (t - 1)))
                    if not np.array_equal(o, want[t - 1]):
                        errors.append((k, rep, t))
            counts.append((c.host_cache_hits(), c.host_cache_misses()))
            c.close()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    print("hits / misses per thread:", counts)
    plain.close()


def test_coalesced_host_call_equals_its_own_call(ofxcv):
    """Option host.coalesce = 2 sends a lone host-image call through the device's submission queue (gray frames gathered into the batch
    context's slots, ONE batched Farneback call there, the RGBA image composed -- or the flows copied back -- on the batch stream): the same
    output as the call's own Farneback call (host.coalesce = 0) for the full channel map, a partial one, one direction, a render scale,
    named frames and the pinned-ring form; the counters tell which path ran."""
    from openfx_opencv_amd import synth
    w, h = 333, 200
    ref, nxt = synth.flow_pair(w, h, seed=7)
    prev, _ = synth.flow_pair(w, h, seed=19)
    own, queued, ring = ofxcv.Context(0), ofxcv.Context(0), ofxcv.Context(0)
    own.set_option("host.coalesce", 0)
    queued.set_option("host.coalesce", 2)
    queued.set_option("host.split", 0)    # (a lone call otherwise takes the split form: two single-pair calls of its own)
    ring.set_option("host.coalesce", 2)
    ring.set_option("host.split", 0)
    ring.set_option("host.register", 0)
    assert queued.get_option("host.coalesce") == 2
    n = 0
    for fu, fv, bu, bv, rx, ry in [(1, 2, 4, 8, 1.0, 1.0), (4, 8, 1, 2, 0.5, 0.25), (3, 12, 0, 0, 1.0, 2.0), (1, 0, 0, 8, 1.0, 1.0), (1, 2, 2, 4, 2.0, 1.0)]:
        outs = []
        for c in (own, queued, ring):
            o = np.full((h, w, 4), -3.0, np.float32)
            c.vectorgen_flows_host(ref, nxt, prev, o, fu, fv, bu, bv, rx, ry)
            outs.append(o)
        n += 1
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), (fu, fv, bu, bv)
        assert queued.host_coalesce_stats() == (n, 2 * n, 2 * n) and own.host_coalesce_stats() == (0, 0, 0)
    # one direction
    a, b = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    own.vectorgen_flow_host(ref, nxt, a, 0b0101, 0b1010)
    queued.vectorgen_flow_host(ref, nxt, b, 0b0101, 0b1010)
    assert np.array_equal(a, b) and queued.host_coalesce_stats() == (n + 1, 2 * n + 1, 2 * n + 1)
    # named frames, found on the device the second time
    queued.host_cache_clear()
    for rep in range(2):
        b[:] = 0
        queued.vectorgen_flows_host(ref, nxt, prev, b, 1, 2, 4, 8, keys=("cq:1", "cq:2", "cq:0"))
        own.vectorgen_flows_host(ref, nxt, prev, a, 1, 2, 4, 8)
        assert np.array_equal(a, b)
    assert queued.host_cache_hits() == 3
    # other numerics switches travel with the request (the batch context takes them over)
    for c in (own, queued):
        c.set_option("farneback.gaussian_kernel_generation", 4)
        c.set_option("farneback.resize_generation", 1)
    own.vectorgen_flows_host(ref, nxt, prev, a, 1, 2, 4, 8)
    queued.vectorgen_flows_host(ref, nxt, prev, b, 1, 2, 4, 8)
    assert np.array_equal(a, b)
    for c in (own, queued, ring):
        c.host_cache_clear()
        c.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("nthreads", [4, 8])
def test_coalesced_host_calls_from_render_threads(ofxcv, nthreads):
    """VectorGenerator is eRenderFullySafe (VectorGenerator.cpp:108): several render threads, one context each, render output frames at
    once.  Their calls are coalesced per device: whoever finds the queue idle runs everything queued as one batched Farneback call.  Every
    output frame is the frame the thread's own call gives (two frame sizes in flight at once: requests of another geometry never share a
    call), and calls did ride together."""
    import threading
    from openfx_opencv_amd import synth
    sizes = [(320, 200), (256, 144)]
    seqs = [[synth.flow_pair(w, h, seed=70 + i)[0] for i in range(10)] for (w, h) in sizes]
    plain = ofxcv.Context(0)
    plain.set_option("host.coalesce", 0)
    plain.host_cache_clear()
    want = {}
    for si, (w, h) in enumerate(sizes):
        for t in range(1, 9):
            o = np.zeros((h, w, 4), np.float32)
            plain.vectorgen_flows_host(seqs[si][t], seqs[si][t + 1], seqs[si][t - 1], o, 1, 2, 4, 8)
            want[(si, t)] = o
    errors, stats = [], []
    start = threading.Barrier(nthreads)

    def work(k):
        try:
            c = ofxcv.Context(0)
            c.set_option("host.coalesce_min", 2)             # (default 4: with four threads the queue would only engage when all four are in flight at once)
            si = k % 2 if k >= nthreads // 2 else 0           # most threads share a size, some render the other one
            w, h = sizes[si]
            start.wait()
            for rep in range(4):
                for t in range(1 + k % 8, 9, 3):
                    o = np.full((h, w, 4), -1.0, np.float32)
                    keys = ("q%d:%d" % (si, t), "q%d:%d" % (si, t + 1), "q%d:%d" % (si, t - 1)) if rep % 2 else None
                    c.vectorgen_flows_host(seqs[si][t], seqs[si][t + 1], seqs[si][t - 1], o, 1, 2, 4, 8, keys=keys)
                    if not np.array_equal(o, want[(si, t)]):
                        errors.append((k, rep, t))
            stats.append(c.host_coalesce_stats())
            c.close()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    calls, pairs, batch_pairs = (sum(s[i] for s in stats) for i in range(3))
    print("coalesced calls %d, their pairs %d, mean pairs of the call they rode in %.2f" % (calls, pairs, batch_pairs / max(1, calls)))
    assert calls > 0 and batch_pairs > pairs      # some calls shared a batched call
    plain.host_cache_clear()
    plain.close()
