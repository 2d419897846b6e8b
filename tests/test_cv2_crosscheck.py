"""Opportunistic cross-check of the CPU oracle against a real OpenCV, if the box has one.

No OpenCV exists in the build image (no cv2 module, no libopencv_*): when that is also true on the GPU box this file
records an explicit "cv2 absent" line and skips -- the oracle then stays pinned by the analytic known-answer tests only
("parity unpinned").  When cv2 is importable, the oracle's FAITHFUL Farneback, Telea inpaint and pyramid
mean-shift restatements are diffed against cv2 on the golden inputs and the result is printed into the test log.  Nothing
of OpenCV is ever shipped with the repository.
"""
import ctypes.util
import glob
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _system_opencv_libs():
    found = []
    try:
        out = subprocess.run(["ldconfig", "-p"], capture_output=True, text=True).stdout
        found += [l.split("=>")[-1].strip() for l in out.splitlines() if "opencv" in l]
    except Exception:
        pass
    for pat in ("/usr/lib/x86_64-linux-gnu/libopencv_video.so*", "/usr/local/lib/libopencv_video.so*", "/opt/conda/lib/libopencv_video.so*"):
        found += glob.glob(pat)
    if ctypes.util.find_library("opencv_video"):
        found.append(ctypes.util.find_library("opencv_video"))
    return sorted(set(found))


def test_probe_reports_what_the_box_has():
    libs = _system_opencv_libs()
    try:
        import cv2
        print("cv2 present: version %s; system OpenCV libraries: %s" % (cv2.__version__, libs or "none"))
    except ImportError:
        print("cv2 absent on this box (ModuleNotFoundError); system OpenCV libraries: %s -> the oracle cannot be diffed against "
              "OpenCV here, parity stays pinned by known-answer tests only" % (libs or "none"))


def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


def test_oracle_farneback_stages_vs_cv2(oracle):
    """Stage by stage, so that the day a cv2 exists the log says WHICH stage and WHICH variant of the three generation switches
    (getGaussianKernel 2.4 / 3.x vs 4.x; the association of cv::resize's exact-2x INTER_AREA rewrite; separable filters with separate roundings or
    fused multiply-adds) agrees with it:
    getGaussianKernel, GaussianBlur on a float image, resize at 2x, 4x and a non-integer factor.  (FarnebackPolyExp and
    FarnebackUpdateFlow_Blur are internal to cv2; past the pyramid image the next observable stage is the whole call, below.)
    Exact counts are printed; the variant that matches the cv2 at hand must match at EVERY sample."""
    cv2 = pytest.importorskip("cv2", reason="cv2 absent: oracle vs OpenCV comparison not possible on this box")
    from openfx_opencv_amd import synth
    a, _ = synth.flow_pair(640, 480)
    ga = oracle.to_byte_grayscale(a)
    f = ga.astype(np.float32)
    report = {}
    try:
        for gauss in (3, 4):
            oracle.set_gaussian_kernel_generation(gauss)
            for n, sigma in ((3, 0.5), (9, 1.5), (19, 3.5)):
                ref = cv2.getGaussianKernel(n, sigma, cv2.CV_32F).ravel()
                report[("getGaussianKernel", gauss, n)] = int((ref != oracle.gaussian_kernel(n, sigma)).sum())
                ref = cv2.GaussianBlur(f, (n, n), sigma, sigmaY=sigma, borderType=cv2.BORDER_REFLECT_101)
                for fc in (0, 1):  # the separable filters with separate roundings (2.4 / 3.x, scalar builds) or fused multiply-adds (4.x AVX2 / NEON)
                    oracle.set_filter_contraction(fc)
                    report[("GaussianBlur", gauss, n, "contraction %d" % fc)] = int((ref != oracle.gaussian_blur(f, n, sigma)).sum())
                oracle.set_filter_contraction(0)
        oracle.set_gaussian_kernel_generation(3)
        blurred = oracle.gaussian_blur(f, 3, 0.5)
        for rz in (0, 1, 2):
            oracle.set_resize_generation(rz)
            for dw, dh in ((320, 240), (160, 120), (213, 160)):
                ref = cv2.resize(blurred, (dw, dh), interpolation=cv2.INTER_LINEAR)
                for fc in (0, 1):
                    oracle.set_filter_contraction(fc)
                    report[("resize", rz, dw, "contraction %d" % fc)] = int((ref != oracle.resize_linear(blurred, dw, dh)).sum())
                oracle.set_filter_contraction(0)
    finally:
        oracle.set_gaussian_kernel_generation(3)
        oracle.set_resize_generation(0)
        oracle.set_filter_contraction(0)
    for k in sorted(report, key=str):
        print("cv2 %s stage diff %s: %d differing samples" % (cv2.__version__, k, report[k]))
    assert min(report[("getGaussianKernel", g, 9)] + report[("getGaussianKernel", g, 19)] for g in (3, 4)) == 0, report
    fcs = ("contraction 0", "contraction 1")
    assert min(report[("GaussianBlur", g, 9, c)] + report[("GaussianBlur", g, 19, c)] for g in (3, 4) for c in fcs) == 0, report
    assert min(report[("resize", rz, 320, c)] for rz in (0, 1, 2) for c in fcs) == 0, report
    assert min(report[("resize", 0, 160, c)] + report[("resize", 0, 213, c)] for c in fcs) == 0, report


def test_oracle_farneback_vs_cv2(oracle):
    """The oracle restates the 2.4 / 3.x sources; 4.x normalises its Gaussian taps in double (orc_set_gaussian_kernel_generation), and
    cv::resize's exact-2x rewrite exists in three associations (orc_set_resize_generation).  All six variants are diffed and the
    exact counts printed: at least one variant must agree with the cv2 at hand at every sample (and the log says which)."""
    cv2 = pytest.importorskip("cv2", reason="cv2 absent: oracle vs OpenCV comparison not possible on this box")
    g = _golden("farneback_96x72.npz")
    from openfx_opencv_amd import synth
    a, b = synth.flow_pair(640, 480)
    ga, gb = oracle.to_byte_grayscale(a), oracle.to_byte_grayscale(b)
    cases = [("96x72 golden pair", g["gray_a"], g["gray_b"]), ("640x480", ga, gb)]
    results = {}
    try:
        for gen in (3, 4):
            for rz in (0, 1, 2):
                for fc in (0, 1):
                    oracle.set_gaussian_kernel_generation(gen)
                    oracle.set_resize_generation(rz)
                    oracle.set_filter_contraction(fc)
                    for name, x, y in cases:
                        ref = cv2.calcOpticalFlowFarneback(x, y, None, 0.5, 3, 3, 15, 5, 1.1, 0)
                        mine = oracle.calc_optical_flow_farneback(x, y, blur_mode=oracle.BLUR_FAITHFUL)
                        err = np.abs(ref - mine)
                        bad = int((err > 1e-4 * np.maximum(1, np.abs(ref))).sum())
                        results[(gen, rz, fc, name)] = bad
                        print("cv2 %s calcOpticalFlowFarneback vs oracle FAITHFUL (getGaussianKernel generation %d, resize generation %d, filter contraction %d), %s: "
                              "max |err| %.3g, outside 1e-4: %d of %d, bit-identical %d"
                              % (cv2.__version__, gen, rz, fc, name, err.max(), bad, err.size, int((ref == mine).sum())))
    finally:
        oracle.set_gaussian_kernel_generation(3)
        oracle.set_resize_generation(0)
        oracle.set_filter_contraction(0)
    assert any(all(results[(gen, rz, fc, name)] == 0 for name, _, _ in cases) for gen in (3, 4) for rz in (0, 1, 2) for fc in (0, 1)), results


def test_oracle_inpaint_vs_cv2(oracle):
    cv2 = pytest.importorskip("cv2", reason="cv2 absent")
    i = _golden("inpaint_96x72.npz")
    rgb = np.ascontiguousarray(i["frame"][..., :3])
    ref = cv2.inpaint(rgb, i["mask"], 3.0, cv2.INPAINT_TELEA)
    diff = (ref != i["out"]).any(axis=2)
    print("cv2 %s inpaint(TELEA) vs oracle on the 96x72 golden frame: %d differing pixels of %d hole pixels, max level diff %d"
          % (cv2.__version__, diff.sum(), (i["mask"] > 0).sum(), np.abs(ref.astype(int) - i["out"].astype(int)).max()))
    assert diff.sum() == 0   # all-integer output: exact


def test_oracle_mean_shift_vs_cv2(oracle):
    cv2 = pytest.importorskip("cv2", reason="cv2 absent")
    m = _golden("meanshift_96x72.npz")
    ref = cv2.pyrMeanShiftFiltering(m["img"], 10, 20, maxLevel=2)
    diff = (ref != m["out"]).any(axis=2)
    print("cv2 %s pyrMeanShiftFiltering(10, 20, 2) vs oracle on the 96x72 golden image: %d differing pixels" % (cv2.__version__, diff.sum()))
    assert diff.sum() == 0   # all-integer path: exact
