"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/surfacenet_hip.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from surfacenet_amd import _lib
    return _lib


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "surfacenet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sn_[a-z0-9_]+)\s*\(", txt)))


def test_header_binding_and_library_agree(built):
    hdr = header_symbols()
    assert hdr == sorted(built.ABI_SYMBOLS)
    lib = built.load()
    for name in hdr:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", built.LIB_PATH]).decode()
    exported = set(re.findall(r" T (sn_[a-z0-9_]+)", out))
    assert set(hdr) <= exported


def test_version_and_no_gpu_fails_loudly(built):
    lib = built.load()
    assert lib.sn_version() == 1
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        have_gpu = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        have_gpu = False
    if have_gpu:
        pytest.skip("a GPU is present")
    import surfacenet_amd
    with pytest.raises(surfacenet_amd.SurfaceNetHipError, match="no CPU fallback"):
        surfacenet_amd.Context(32, 4)
    # the drop-in modules fail the same way: nothing routes to a CPU implementation
    import numpy as np
    from surfacenet_amd import CVC, runtime
    runtime.reset()
    with pytest.raises(surfacenet_amd.SurfaceNetHipError):
        CVC.gen_coloredCubes(np.zeros((1, 1, 2), np.int64), np.zeros((1, 3), np.float32), np.ones(1, np.float32),
                             np.zeros((2, 3, 4)), [np.zeros((4, 4, 3), np.uint8)] * 2, 32)


def test_code_object_is_gfx950_only(built):
    out = subprocess.run(["strings", built.LIB_PATH], capture_output=True, text=True).stdout
    targets = set(re.findall(r"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", out))
    assert targets == {"gfx950"}, targets
