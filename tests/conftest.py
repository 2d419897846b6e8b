import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so, built on demand with gcc).  Checker only."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def ofxcv():
    import openfx_opencv_amd
    return openfx_opencv_amd


@pytest.fixture(scope="session")
def gpu_ctx(ofxcv):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    ctx = ofxcv.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def direct_ctx(ofxcv):
    """a context in the direct-window mode (each 3x3 box window summed on its own in f64: the fast opt-in path, bit-identical
    to the oracle's DIRECT evaluation; the default mode reproduces OpenCV's running-sum order instead)"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    ctx = ofxcv.Context(0)
    ctx.set_option("farneback.opencv_rounding", 0)
    yield ctx
    ctx.close()
