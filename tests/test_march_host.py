"""Host logic of the Telea inpaint: the component-parallel front march (openfx-opencv_amd/csrc/telea_march.h) must give, on
random hole masks (separate blobs, blobs cut by one-pixel walls so that band seeds are shared between components, both
methods' set-ups, second calls on one state), exactly the serial march's fill order and distance map.  Pure C++ on the CPU:
built with g++ and run here; the -m gpu tests then check the same order against the oracle through the C ABI."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parallel_march_equals_serial_march(tmp_path):
    exe = str(tmp_path / "test_march")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "openfx-opencv_amd", "csrc"),
                           os.path.join(ROOT, "tests", "march", "test_march.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "0 failed" in out.stdout
