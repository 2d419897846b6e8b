"""The drop-in boundary: the built .ofx bundles are loaded by an in-repo mock OFX host (tests/mock_host) and must
declare exactly what the reference plugins declare (identifiers, versions, describe() properties, clips, the full
parameter set -- SURVEY.md section 8(b), citing the reference file:line) and follow its action / error convention.
CPU tests stop before render(); the render tests need the GPU."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUNDLES = os.path.join(ROOT, "openfx-opencv_amd", "plugin", "bundles")
STAT_OK, STAT_FAILED, STAT_MISSING, STAT_REPLY_DEFAULT, STAT_IMAGE_FORMAT = 0, 1, 4, 14, 1000


def _ofx(name):
    return os.path.join(BUNDLES, "%s.ofx.bundle" % name, "Contents", "Linux-x86-64", "%s.ofx" % name)


@pytest.fixture(scope="session")
def host():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "mock_host")])
    if not os.path.exists(_ofx("inpaint")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "openfx-opencv_amd"), "-j4"])
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "openfx-opencv_amd", "plugin"), "-j4"])
    try:
        import torch  # noqa: F401  one HIP runtime per process: load torch's first (see openfx_opencv_amd.lib)
    except ImportError:
        pass
    h = C.CDLL(os.path.join(ROOT, "tests", "mock_host", "libmockhost.so"))
    h.mh_open.restype = C.c_void_p
    h.mh_plugin_identifier.restype = C.c_char_p
    h.mh_plugin_api.restype = C.c_char_p
    h.mh_create_instance.restype = C.c_void_p
    h.mh_last_message.restype = C.c_char_p
    return h


class Plugin:
    def __init__(self, host, name, context="OfxImageEffectContextFilter", load=True):
        self.h = host
        self.p = C.c_void_p(host.mh_open(_ofx(name).encode()))
        assert self.p.value, "dlopen of %s failed" % name
        if load:
            assert host.mh_load(self.p, 1) == STAT_OK
            assert host.mh_describe(self.p, context.encode()) == STAT_OK

    def dump(self):
        buf = C.create_string_buffer(1 << 18)
        n = self.h.mh_dump(self.p, buf, len(buf))
        assert n > 0
        return json.loads(buf.value.decode())

    def instance(self):
        st = C.c_int()
        inst = C.c_void_p(self.h.mh_create_instance(self.p, C.byref(st)))
        assert st.value == STAT_OK and inst.value
        return inst

    def set_image(self, inst, clip, t, arr, depth, comps="OfxImageComponentRGBA", rs=(1.0, 1.0), origin=(0, 0)):
        hh, ww = arr.shape[:2]
        assert self.h.mh_set_image(inst, clip.encode(), C.c_double(t), C.c_void_p(arr.ctypes.data), origin[0], origin[1], origin[0] + ww,
                                   origin[1] + hh, arr.strides[0], depth.encode(), comps.encode(), C.c_double(rs[0]), C.c_double(rs[1])) == 0

    def render(self, inst, t, w, h, rs=(1.0, 1.0)):
        return self.h.mh_render(self.p, inst, C.c_double(t), 0, 0, w, h, C.c_double(rs[0]), C.c_double(rs[1]))

    def destroy(self, inst):
        assert self.h.mh_destroy_instance(self.p, inst) == STAT_OK


def _props(d):
    return {k: (v[0] if len(v) == 1 else v) for k, v in d.items()}


# ------------------------------------------------------------------------------------------------ CPU: load / describe

@pytest.mark.parametrize("name,ident,major,minor", [
    ("inpaint", b"uk.org.bratwurstandhaggis:cvInpaint", 0, 5),                 # inpaint.cpp:584-593
    ("segment", b"uk.org.bratwurstandhaggis:cvPyrSegmentation", 0, 5),         # segment.cpp:533-542
    ("VectorGenerator", b"net.sf.openfx.VectorGenerator", 1, 0)])              # VectorGenerator.cpp:101-103
def test_plugin_struct(host, name, ident, major, minor):
    pl = Plugin(host, name, load=False)
    assert host.mh_plugin_count(pl.p) == 1
    assert host.mh_get_plugin_is_null(pl.p, 0) == 0 and host.mh_get_plugin_is_null(pl.p, 1) == 1 and host.mh_get_plugin_is_null(pl.p, -1) == 1
    assert host.mh_plugin_api(pl.p) == b"OfxImageEffectPluginAPI" and host.mh_plugin_api_version(pl.p) == 1
    assert host.mh_plugin_identifier(pl.p) == ident
    assert host.mh_plugin_version(pl.p, 0) == major and host.mh_plugin_version(pl.p, 1) == minor


@pytest.mark.parametrize("name", ["inpaint", "segment", "VectorGenerator"])
def test_error_convention(host, name):
    pl = Plugin(host, name, load=False)
    # Load before setHost: no host -> kOfxStatErrMissingHostFeature (inpaint.cpp:522)
    assert host.mh_load(pl.p, 0) == STAT_MISSING
    assert host.mh_load(pl.p, 1) == STAT_OK
    assert host.mh_action_raw(pl.p, b"OfxActionSomethingElse") == STAT_REPLY_DEFAULT      # unknown actions (:572)
    assert host.mh_action_raw(pl.p, b"OfxImageEffectActionGetRegionOfDefinition") == STAT_REPLY_DEFAULT
    # a host without the parameter suite: describeInContext reports the missing feature (:414)
    host.mh_hide_param_suite(1)
    try:
        assert host.mh_describe(pl.p, b"OfxImageEffectContextFilter") == STAT_MISSING
    finally:
        host.mh_hide_param_suite(0)
    assert host.mh_describe(pl.p, b"OfxImageEffectContextFilter") == STAT_OK
    # a suite error inside an action is returned as a status, never thrown across the C ABI: a second
    # DescribeInContext on the same descriptor would re-define the clips (kOfxStatErrExists = 6) -- the mock host
    # hands out a fresh descriptor per mh_describe, so provoke it through a render on an instance without images
    inst = pl.instance()
    assert pl.render(inst, 1.0, 8, 8) == STAT_FAILED           # clipGetImage fails -> status, no crash
    assert host.mh_clip_balance(inst, b"Output") == 0 and host.mh_clip_balance(inst, b"Source") == 0
    pl.destroy(inst)


def _same_text(text, length, sha256):
    import hashlib
    assert len(text) == length and hashlib.sha256(text.encode()).hexdigest() == sha256


def test_describe_inpaint(host):
    d = Plugin(host, "inpaint").dump()
    p = _props(d["props"])
    assert p["OfxImageEffectPropMultipleClipDepths"] == 0 and p["OfxImageEffectPropSupportedPixelDepths"] == "OfxBitDepthByte"
    assert p["OfxPropLabel"] == "openCV Inpaint" and p["OfxImageEffectPluginPropGrouping"] == "Draw"
    assert p["OfxImageEffectPropSupportedContexts"] == "OfxImageEffectContextFilter"
    assert (p["OfxImageEffectPluginPropSingleInstance"], p["OfxImageEffectPluginPropHostFrameThreading"],
            p["OfxImageEffectPropSupportsMultiResolution"], p["OfxImageEffectPropSupportsTiles"], p["OfxImageEffectPropTemporalClipAccess"],
            p["OfxImageEffectPluginPropFieldRenderTwiceAlways"], p["OfxImageEffectPropSupportsMultipleClipPARs"]) == (0, 0, 0, 0, 0, 1, 0)
    assert "OfxImageEffectPluginRenderThreadSafety" not in p          # default (instance safe), as in the reference
    assert [c["name"] for c in d["clips"]] == ["Output", "Source"]
    assert all(_props(c["props"])["OfxImageEffectPropSupportedComponents"] == "OfxImageComponentRGBA" for c in d["clips"])
    params = {q["name"]: (q["type"], _props(q["props"])) for q in d["params"]}
    assert [q["name"] for q in d["params"]] == ["threshold1", "threshold2", "inpaintnoise", "Main"]
    for name, label, dmin, dmax, default in [("threshold1", "Radius", 1, 10, 3), ("threshold2", "Dilation", 1, 5, 1), ("inpaintnoise", "Inpaint noise", 0, 1, 0)]:
        t, q = params[name]
        assert t == "OfxParamTypeDouble" and q["OfxParamPropDoubleType"] == "OfxParamDoubleTypeScale"
        assert (q["OfxPropLabel"], q["OfxParamPropDisplayMin"], q["OfxParamPropDisplayMax"], q["OfxParamPropDefault"], q["OfxParamPropMin"]) == (label, dmin, dmax, default, 0)
        assert q["OfxParamPropScriptName"] == name
    assert params["Main"][0] == "OfxParamTypePage" and params["Main"][1]["OfxParamPropPageChild"] == ["threshold1", "threshold2"]
    # kOfxPropPluginDescription is the reference's value (inpaint.cpp:40-67: credit line + the OFX Association licence notice,
    # incl. its missing line breaks): length and SHA-256 of the reference's string literal
    _same_text(p["OfxPropPluginDescription"], 1618, "5d613ff3611447ff99c450a0c2cad42cc9cb3b7b316112509c478c783310a53c")
    assert p["OfxPropPluginDescription"].startswith("OpenCV inpaint. Wrapper provided by Bernd Porr")


def test_describe_segment(host):
    d = Plugin(host, "segment").dump()
    p = _props(d["props"])
    assert p["OfxPropLabel"] == "openCV Segment" and p["OfxImageEffectPluginPropGrouping"] == "Draw"
    assert p["OfxImageEffectPropSupportedPixelDepths"] == "OfxBitDepthByte" and p["OfxImageEffectPluginPropFieldRenderTwiceAlways"] == 1
    assert [q["name"] for q in d["params"]] == ["threshold1", "threshold2", "Main"]
    params = {q["name"]: _props(q["props"]) for q in d["params"]}
    assert (params["threshold1"]["OfxPropLabel"], params["threshold1"]["OfxParamPropDefault"], params["threshold1"]["OfxParamPropDisplayMax"]) == ("threshold 1", 250, 255)
    assert (params["threshold2"]["OfxPropLabel"], params["threshold2"]["OfxParamPropDefault"], params["threshold2"]["OfxParamPropDisplayMin"]) == ("threshold 2", 30, 1)
    assert params["Main"]["OfxParamPropPageChild"] == ["threshold1", "threshold2"]
    _same_text(p["OfxPropPluginDescription"], 1540, "f9fe0293138e7f67a2bd39cf812ed1e507848271978c8efd5f3fe2a8e8d1458d")   # segment.cpp:41-66
    assert '"AS IS" ANDANY EXPRESS' in p["OfxPropPluginDescription"]                    # the reference's missing line break


def test_describe_vectorgenerator(host):
    for ctx in ("OfxImageEffectContextFilter", "OfxImageEffectContextGeneral"):
        d = Plugin(host, "VectorGenerator", ctx).dump()
        p = _props(d["props"])
        assert p["OfxPropLabel"] == p["OfxPropShortLabel"] == p["OfxPropLongLabel"] == "VectorGeneratorOFX"
        assert p["OfxImageEffectPluginPropGrouping"] == "Time"
        assert p["OfxPropPluginDescription"] == "Compute optical flow for the input sequence, using OpenCV."
        assert p["OfxImageEffectPropSupportedContexts"] == ["OfxImageEffectContextFilter", "OfxImageEffectContextGeneral"]
        assert p["OfxImageEffectPropSupportedPixelDepths"] == "OfxBitDepthFloat"
        assert (p["OfxImageEffectPluginPropSingleInstance"], p["OfxImageEffectPluginPropHostFrameThreading"], p["OfxImageEffectPropSupportsMultiResolution"],
                p["OfxImageEffectPropSupportsTiles"], p["OfxImageEffectPropTemporalClipAccess"], p["OfxImageEffectPluginPropFieldRenderTwiceAlways"],
                p["OfxImageEffectPropSupportsMultipleClipPARs"]) == (0, 0, 1, 0, 1, 0, 0)
        assert p["OfxImageEffectPluginRenderThreadSafety"] == "OfxImageEffectRenderFullySafe"
        clips = {c["name"]: _props(c["props"]) for c in d["clips"]}
        assert clips["Source"]["OfxImageEffectPropSupportedComponents"] == ["OfxImageComponentRGBA", "OfxImageComponentRGB", "OfxImageComponentAlpha"]
        assert clips["Source"]["OfxImageEffectPropTemporalClipAccess"] == 1 and clips["Source"]["OfxImageClipPropIsMask"] == 0
        assert clips["Output"]["OfxImageEffectPropSupportedComponents"] == "OfxImageComponentRGBA"
        names = [q["name"] for q in d["params"]]
        assert names == ["Controls", "rChannel", "gChannel", "bChannel", "aChannel", "method", "levels", "iterations", "neighborhood", "sigma",
                         "tau", "lambda", "theta", "nScales", "warps", "epsilon"]
        params = {q["name"]: (q["type"], _props(q["props"])) for q in d["params"]}
        assert params["Controls"][1]["OfxParamPropPageChild"] == names[1:]
        for i, n in enumerate(["rChannel", "gChannel", "bChannel", "aChannel"]):
            t, q = params[n]
            assert t == "OfxParamTypeChoice" and q["OfxParamPropDefault"] == i + 1 and q["OfxParamPropAnimates"] == 1
            assert q["OfxParamPropChoiceOption"] == ["0", "forward.u", "forward.v", "backward.u", "backward.v"]
            # appendOption(name, hint), VectorGenerator.cpp:126-137,734-738
            assert q["OfxParamPropChoiceLabelOption"] == ["0 constant channel", "x flow (in pixels) to the next frame.", "y flow (in pixels) to the next frame.",
                                                          "x flow (in pixels) to the previous frame.", "x flow (in pixels) to the previous frame."]
        assert params["method"][1]["OfxParamPropChoiceOption"] == ["Farneback", "Dual TV L1"] and params["method"][1]["OfxParamPropAnimates"] == 0
        for n, t, dflt in [("levels", "OfxParamTypeInteger", 3), ("iterations", "OfxParamTypeInteger", 15), ("neighborhood", "OfxParamTypeInteger", 5),
                           ("sigma", "OfxParamTypeDouble", 1.1), ("tau", "OfxParamTypeDouble", 0.25), ("lambda", "OfxParamTypeDouble", 0.15),
                           ("theta", "OfxParamTypeDouble", 0.3), ("nScales", "OfxParamTypeInteger", 5), ("warps", "OfxParamTypeInteger", 5),
                           ("epsilon", "OfxParamTypeDouble", 0.01)]:
            assert params[n][0] == t and params[n][1]["OfxParamPropDefault"] == dflt


def test_vectorgenerator_frames_needed_and_visibility(host):
    pl = Plugin(host, "VectorGenerator")
    inst = pl.instance()
    rng = (C.c_double * 2)()
    have = C.c_int()
    assert host.mh_get_frames_needed(pl.p, inst, C.c_double(10.0), rng, C.byref(have)) == STAT_OK
    assert have.value == 1 and list(rng) == [9.0, 11.0]                     # defaults need t-1 .. t+1 (VectorGenerator.cpp:688-693)
    for n in ("bChannel", "aChannel"):
        host.mh_set_param_double(inst, n.encode(), C.c_double(0))
    host.mh_get_frames_needed(pl.p, inst, C.c_double(10.0), rng, C.byref(have))
    assert list(rng) == [10.0, 11.0]                                          # forward only
    for n in ("rChannel", "gChannel"):
        host.mh_set_param_double(inst, n.encode(), C.c_double(0))
    host.mh_get_frames_needed(pl.p, inst, C.c_double(10.0), rng, C.byref(have))
    assert have.value == 0                                                    # nothing mapped: no request
    # method -> parameter visibility (updateVisibility, :642-662)
    assert host.mh_get_param_secret(inst, b"levels") == 0 and host.mh_get_param_secret(inst, b"tau") == 1
    host.mh_set_param_double(inst, b"method", C.c_double(2))
    assert host.mh_instance_changed(pl.p, inst, b"method") == STAT_OK
    assert host.mh_get_param_secret(inst, b"levels") == 1 and host.mh_get_param_secret(inst, b"tau") == 0 and host.mh_get_param_secret(inst, b"iterations") == 0
    assert host.mh_instance_changed(pl.p, inst, b"sigma") == STAT_REPLY_DEFAULT
    pl.destroy(inst)


# ------------------------------------------------------------------------------------------------ GPU: render through the boundary

@pytest.mark.gpu
def test_inpaint_render_through_ofx(host, oracle):
    from openfx_opencv_amd import synth
    fr = synth.inpaint_frame(640, 480)                                       # BASELINE config 1 frame
    pl = Plugin(host, "inpaint")
    inst = pl.instance()
    out = np.zeros_like(fr)
    pl.set_image(inst, "Source", 1.0, fr, "OfxBitDepthByte")
    pl.set_image(inst, "Output", 1.0, out, "OfxBitDepthByte")
    assert pl.render(inst, 1.0, 640, 480) == STAT_OK
    assert np.array_equal(out, oracle.inpaint_render(fr, 3.0, 1.0))
    assert host.mh_clip_balance(inst, b"Output") == 0 and host.mh_clip_balance(inst, b"Source") == 0    # every image released
    host.mh_set_param_double(inst, b"threshold1", C.c_double(5.0))
    host.mh_set_param_double(inst, b"threshold2", C.c_double(2.0))
    assert pl.render(inst, 1.0, 640, 480) == STAT_OK
    assert np.array_equal(out, oracle.inpaint_render(fr, 5.0, 2.0))
    # wrong bit depth -> kOfxStatErrImageFormat, images still released
    f32 = np.zeros((480, 640, 4), np.float32)
    pl.set_image(inst, "Source", 2.0, f32, "OfxBitDepthFloat")
    pl.set_image(inst, "Output", 2.0, out, "OfxBitDepthByte")
    assert pl.render(inst, 2.0, 640, 480) == STAT_IMAGE_FORMAT
    assert host.mh_clip_balance(inst, b"Source") == 0
    pl.destroy(inst)


@pytest.mark.gpu
def test_inpaint_noise_through_ofx(host, oracle):
    """inpaintnoise > 0: the plugin's write-back adds libc rand() noise on masked pixels (inpaint.cpp:320-358).  Replayed
    here with the same libc: srand(rs) for every x of every 4th row, a draw on every 4th x of a masked pixel, rs chained
    on the red value; noise_div = (int)(1/noise)."""
    from openfx_opencv_amd import synth
    libc = C.CDLL(None)
    w, h = 160, 120
    fr = synth.inpaint_frame(w, h)
    for noise in (0.5, 0.2):
        pl = Plugin(host, "inpaint")
        inst = pl.instance()
        out = np.zeros_like(fr)
        pl.set_image(inst, "Source", 1.0, fr, "OfxBitDepthByte")
        pl.set_image(inst, "Output", 1.0, out, "OfxBitDepthByte")
        host.mh_set_param_double(inst, b"inpaintnoise", C.c_double(noise))
        assert pl.render(inst, 1.0, w, h) == STAT_OK
        clean = oracle.inpaint_render(fr, 3.0, 1.0)
        mask = oracle.inpaint_mask(fr, 1)
        noise_div = int(1 / noise) or 1
        ref = clean.copy()
        rs = 0
        for y in range(h):
            for x in range(w):
                if y % 4 == 0:
                    libc.srand(rs)
                a = 0
                if mask[y, x] > 0 and x % 4 == 0:
                    a = int((libc.rand() % 10 - 5) / noise_div)          # C division truncates towards zero
                    rs = (rs + int(clean[y, x, 0])) % 256
                ref[y, x, :3] = np.clip(clean[y, x, :3].astype(int) + a, 0, 255)
        assert np.array_equal(out, ref)
        assert (out != clean).any()                                          # the noise really was applied
        pl.destroy(inst)


@pytest.mark.gpu
def test_segment_render_through_ofx(host, oracle):
    from openfx_opencv_amd import synth
    fr = synth.inpaint_frame(322, 243, n_holes=0)                            # not a multiple of 4: the plugin rounds down
    pl = Plugin(host, "segment")
    inst = pl.instance()
    out = np.full_like(fr, 7)
    pl.set_image(inst, "Source", 1.0, fr, "OfxBitDepthByte")
    pl.set_image(inst, "Output", 1.0, out, "OfxBitDepthByte")
    assert pl.render(inst, 1.0, 322, 243) == STAT_OK
    w, h = 320, 240
    src = np.ascontiguousarray(fr[:h, :w, :3])
    ref = oracle.pyr_mean_shift(src, 10.0, 30.0, 2)                           # threshold1 250 / 25 -> spatial radius 10, threshold2 30 -> colour radius
    assert np.array_equal(out[:h, :w, :3], ref) and (out[:h, :w, 3] == 255).all()
    assert (out[h:] == 7).all() and (out[:, w:] == 7).all()                   # outside the reduced rectangle: untouched
    # the substitution of cvPyrSegmentation is announced through the message suite
    assert b"pyramid mean-shift filtering" in host.mh_last_message(inst)
    # threshold 1 is a live control: 125 -> spatial radius 5
    host.mh_set_param_double(inst, b"threshold1", C.c_double(125.0))
    host.mh_set_param_double(inst, b"threshold2", C.c_double(20.0))
    assert pl.render(inst, 1.0, 322, 243) == STAT_OK
    assert np.array_equal(out[:h, :w, :3], oracle.pyr_mean_shift(src, 5.0, 20.0, 2))
    # OfxActionUnload releases the device contexts; a later render re-creates them
    assert host.mh_action_raw(pl.p, b"OfxActionUnload") == STAT_REPLY_DEFAULT       # as in the reference (:483-522 has no Unload case)
    assert pl.render(inst, 1.0, 322, 243) == STAT_OK
    assert np.array_equal(out[:h, :w, :3], oracle.pyr_mean_shift(src, 5.0, 20.0, 2))
    pl.destroy(inst)


@pytest.mark.gpu
def test_vectorgenerator_render_through_ofx(host, oracle):
    from openfx_opencv_amd import synth
    w, h = 320, 240
    a, b = synth.flow_pair(w, h)
    c, _ = synth.flow_pair(w, h, seed=77)
    pl = Plugin(host, "VectorGenerator")
    inst = pl.instance()
    out = np.full((h, w, 4), -9.0, np.float32)
    for t, img in ((4.0, c), (5.0, a), (6.0, b)):
        pl.set_image(inst, "Source", t, img, "OfxBitDepthFloat")
    pl.set_image(inst, "Output", 5.0, out, "OfxBitDepthFloat")
    assert pl.render(inst, 5.0, w, h) == STAT_OK
    ga, gb, gc = (oracle.to_byte_grayscale(x) for x in (a, b, c))
    # the plugin runs the library default (OpenCV-order box window): every sample within the north_star tolerance of the
    # faithful oracle (on these frames the result is in fact identical to it)
    fwd = oracle.calc_optical_flow_farneback(ga, gb, blur_mode=oracle.BLUR_FAITHFUL)
    bwd = oracle.calc_optical_flow_farneback(ga, gc, blur_mode=oracle.BLUR_FAITHFUL)
    close = lambda got, ref: bool((np.abs(got - ref) <= 1e-4 * np.maximum(1, np.abs(ref))).all())
    assert close(out[..., 0], fwd[..., 0]) and close(out[..., 1], fwd[..., 1])      # R,G = forward.u,v
    assert close(out[..., 2], bwd[..., 0]) and close(out[..., 3], bwd[..., 1])      # B,A = backward.u,v
    assert (np.stack([out[..., 0], out[..., 1]], -1) == fwd).mean() > 0.999
    assert host.mh_clip_balance(inst, b"Source") == 0 and host.mh_clip_balance(inst, b"Output") == 0
    # forward.v only into alpha, render scale 0.5: flow is divided by the scale; other channels untouched
    for n, v in (("rChannel", 0), ("gChannel", 0), ("bChannel", 0), ("aChannel", 2)):
        host.mh_set_param_double(inst, n.encode(), C.c_double(v))
    out2 = np.full((h, w, 4), -9.0, np.float32)
    for t, img in ((5.0, a), (6.0, b)):
        pl.set_image(inst, "Source", t, img, "OfxBitDepthFloat", rs=(0.5, 0.5))
    pl.set_image(inst, "Output", 5.0, out2, "OfxBitDepthFloat", rs=(0.5, 0.5))
    assert pl.render(inst, 5.0, w, h, rs=(0.5, 0.5)) == STAT_OK
    assert close(out2[..., 3], (fwd[..., 1].astype(np.float64) / 0.5).astype(np.float32)) and (out2[..., :3] == -9.0).all()
    # scale mismatch between the render call and the image -> kOfxStatFailed with the reference's message
    assert pl.render(inst, 5.0, w, h, rs=(1.0, 1.0)) == STAT_FAILED
    assert b"wrong scale or field" in host.mh_last_message(inst)
    # Dual TV L1 is outside the accelerated path
    host.mh_set_param_double(inst, b"method", C.c_double(2))
    assert pl.render(inst, 5.0, w, h, rs=(0.5, 0.5)) == STAT_FAILED
    pl.destroy(inst)


@pytest.mark.gpu
def test_vectorgenerator_named_frames_follow_their_block_across_devices(host, monkeypatch):
    """with several devices the plugin sends BLOCKS of 16 consecutive frame times to one device when the host names its
    images (a named frame stays on the device it was uploaded to), instead of whatever device the calling thread happens to own.  Two logical
    devices over the one GPU (OFXCV_VIRTUAL_DEVICES): output frame 16 (block 1 -> device 1) leaves frames 15, 16, 17 on device 1; frame 16's pixels
    then change under the SAME name; output frame 15 (block 0 -> device 0) has never seen the name there and uploads the new pixels, output frame 17
    (block 1) finds the old ones on its device -- the contract of the identifier, per device."""
    from openfx_opencv_amd import synth
    w, h = 288, 160
    seq = {u: synth.flow_pair(w, h, seed=300 + u)[0].copy() for u in range(13, 20)}
    pl = Plugin(host, "VectorGenerator")

    def render(inst, t, named):
        out = np.full((h, w, 4), -9.0, np.float32)
        for u in (t - 1, t, t + 1):
            pl.set_image(inst, "Source", float(u), seq[u], "OfxBitDepthFloat")
            if named:
                assert host.mh_set_image_id(inst, b"Source", C.c_double(float(u)), ("ofxcv-blocks-%d" % u).encode()) == 0
        pl.set_image(inst, "Output", float(t), out, "OfxBitDepthFloat")
        assert pl.render(inst, float(t), w, h) == STAT_OK
        return out

    monkeypatch.setenv("OFXCV_VIRTUAL_DEVICES", "2")
    inst = pl.instance()
    old17 = render(inst, 17, False)                      # (unnamed renders: the pixels as they are)
    first16 = render(inst, 16, True)                     # names 15, 16, 17 now live on device 1
    assert np.array_equal(first16, render(inst, 16, False))
    seq[16][...] = synth.flow_pair(w, h, seed=999)[0]    # new pixels under the old name
    new15 = render(inst, 15, False)
    assert np.array_equal(render(inst, 15, True), new15)       # block 0 = device 0: the name is new there, the new pixels are uploaded
    assert np.array_equal(render(inst, 17, True), old17)       # block 1 = device 1: frame 16 is found there, old pixels
    assert not np.array_equal(render(inst, 17, False), old17)
    pl.destroy(inst)


@pytest.mark.gpu
def test_vectorgenerator_sequence_with_image_identifiers(host):
    """A host that names its images' pixels (kOfxImagePropUniqueIdentifier): rendering the frames of a sequence one after the
    other through the OFX boundary gives the frames a host without identifiers gets.  That the names reach the library (whose
    cache lives inside the bundle) shows in the contract itself: pixels that change under an UNCHANGED identifier are not
    looked at again -- the host must rename them, as the OFX property says -- and under a new identifier they are."""
    from openfx_opencv_amd import synth
    w, h = 288, 160
    seq = [synth.flow_pair(w, h, seed=20 + i)[0].copy() for i in range(5)]
    pl = Plugin(host, "VectorGenerator")
    outs = {}

    def render(inst, t, ids, frames=None):
        frames = seq if frames is None else frames
        out = np.full((h, w, 4), -9.0, np.float32)
        for u in (t - 1, t, t + 1):
            pl.set_image(inst, "Source", float(u), frames[u], "OfxBitDepthFloat")
            if ids:
                assert host.mh_set_image_id(inst, b"Source", C.c_double(float(u)), ids[u].encode()) == 0
        pl.set_image(inst, "Output", float(t), out, "OfxBitDepthFloat")
        assert pl.render(inst, float(t), w, h) == STAT_OK
        return out

    ids = ["ofxcv-test-seq-%d-v1" % u for u in range(5)]
    for named in (False, True):
        inst = pl.instance()
        for t in (1, 2, 3):
            outs[(named, t)] = render(inst, t, ids if named else None)
        pl.destroy(inst)
    for t in (1, 2, 3):
        assert np.array_equal(outs[(True, t)], outs[(False, t)]), t
    inst = pl.instance()
    assert np.array_equal(render(inst, 2, ids), outs[(True, 2)])    # frames 1 .. 3 on the device under this clip's names
    seq[3][...] = synth.flow_pair(w, h, seed=91)[0]                 # new pixels in frame 3 ...
    stale = render(inst, 2, ids)                                    # ... under its old name: the frame on the device is used
    assert np.array_equal(stale, outs[(True, 2)])
    ids[3] = "ofxcv-test-seq-3-v2"                                  # renamed, as a host does when pixels change
    fresh = render(inst, 2, ids)
    assert np.array_equal(fresh, render(inst, 2, None)) and not np.array_equal(fresh, stale)
    # a second instance whose host hands out the SAME identifiers for other pixels (identifiers that are only unique per clip): the names carry the
    # clip they came from, so it gets its own frames
    other = pl.instance()
    seq_b = [synth.flow_pair(w, h, seed=60 + i)[0].copy() for i in range(5)]
    got_b = render(other, 2, ids, seq_b)
    assert np.array_equal(got_b, render(other, 2, None, seq_b)) and not np.array_equal(got_b, fresh)
    pl.destroy(other)
    pl.destroy(inst)
