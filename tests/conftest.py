import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from surfacenet_amd import _lib
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0 and os.path.exists(_lib.LIB_PATH)
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    if not _gpu_available():
        pytest.fail("GPU test selected but no MI355X / libsurfacenet_hip.so available: the HIP path has no fallback")
