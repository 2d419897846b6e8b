"""Sanitizer builds of the host side (SURVEY section 5's plan; the reference has only assert()s, VectorGenerator.cpp:400-401): the three
plugins' host code, every translation unit of libofxcv_hip compiled --cuda-host-only (contexts, named-frame cache, submission queue, level
plan, launch sequence, Telea front march) and the mock host, built with ROCm's clang under AddressSanitizer + UndefinedBehaviorSanitizer and
under ThreadSanitizer against tests/sanitize/hip_stub.cpp (host memory as the device, no kernel runs), then driven by tests/sanitize/san_driver.cpp:
eight threads render consecutive named frames on ONE VectorGenerator instance (the submission queue coalesces them), four threads render
inpaint / segment instances, plugins are unloaded.  Clean = exit status 0 and no sanitizer report in the output."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["asan", "tsan"])
def test_host_side_is_clean_under_sanitizers(mode):
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang not installed")
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "sanitize"), mode], capture_output=True, text=True)
    text = out.stdout + out.stderr
    print(text[-3000:])
    assert out.returncode == 0, text[-3000:]
    assert "san_driver: 0 failed expectations" in text and "0 failed" in text
    for marker in ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "ERROR: LeakSanitizer"):
        assert marker not in text, text[-3000:]
