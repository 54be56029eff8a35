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
    # the header IS the dynamic symbol table: the library is built with -fvisibility=hidden, so nothing else - C++ internals (pack_conv, rccl
    # glue), kernel host stubs, the sn_debug_* test hooks - may be exported by the product .so
    out = subprocess.check_output(["nm", "-D", "--defined-only", built.LIB_PATH]).decode()
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == hdr, sorted(set(exported) ^ set(hdr))
    dbg = os.path.join(os.path.dirname(built.LIB_PATH), "libsurfacenet_hip_dbg.so")       # the test-only twin: the same ABI + the hooks
    out = subprocess.check_output(["nm", "-D", "--defined-only", dbg]).decode()
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert set(hdr) <= set(exported) and set(exported) - set(hdr) == {"sn_debug_tensor", "sn_debug_mx6_encode", "sn_debug_timing", "sn_debug_trace", "sn_debug_pack_host"}


def test_version_and_no_gpu_fails_loudly(built):
    lib = built.load()
    assert lib.sn_version() == 2          # include/surfacenet_hip.h SN_ABI_VERSION (2 since round 6: + sn_mfma_probe, sn_set_conv4_fp8)
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


def _dec6(code, fmt):
    """fp6 e2m3 (fmt 2, bias 1) / bf6 e3m2 (fmt 3, bias 3) code -> value (OCP MX formats; tools/probe/fp6_probe.hip dec6)."""
    mb, bias = (3, 1) if fmt == 2 else (2, 3)
    s, E, M = code >> 5, (code & 31) >> mb, code & ((1 << mb) - 1)
    v = M * 2.0 ** (1 - bias - mb) if E == 0 else ((1 << mb) + M) * 2.0 ** (E - bias - mb)
    return -v if s else v


@pytest.mark.parametrize("fmt", [2, 3])
def test_host_mx6_encoder_matches_the_format(built, fmt):
    """The weight packer's 6-bit encoder (sn_api.hip mx6_encode): exact on every representable value, round-to-nearest-even on the
    midpoints, saturating, monotone - the properties the device-side conversions have (fp6_probe) and the CPU model assumes."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(os.path.dirname(built.LIB_PATH), "libsurfacenet_hip_dbg.so"))      # hooks live in the test-only twin
    enc = lib.sn_debug_mx6_encode
    enc.restype = ctypes.c_int
    enc.argtypes = [ctypes.c_float, ctypes.c_int]
    vals = [_dec6(c, fmt) for c in range(32)]
    assert vals == sorted(vals) and vals[0] == 0.0
    for c in range(64):
        if c == 32:
            continue                                         # -0
        assert enc(_dec6(c, fmt), fmt) == c, c
    for c in range(31):                                      # midpoints: ties to the even code; just off the midpoint: nearest
        lo, hi = vals[c], vals[c + 1]
        mid = (lo + hi) / 2
        assert enc(mid, fmt) == (c if c % 2 == 0 else c + 1), (c, mid)
        assert enc(mid - (hi - lo) / 64, fmt) == c and enc(mid + (hi - lo) / 64, fmt) == c + 1
        assert enc(-mid, fmt) == 32 + (c if c % 2 == 0 else c + 1)
    top = vals[31]
    assert top == (7.5 if fmt == 2 else 28.0)
    assert enc(top * 1.01, fmt) == 31 and enc(1e30, fmt) == 31 and enc(-1e30, fmt) == 63
    assert enc(vals[1] / 4, fmt) == 0 and enc(float("nan"), fmt) == 0
    assert enc(1.0, 0) == -1


def _packed_halfs(cin, ks, taps, cs8, split, nf, nsplit):
    """Length of pack_conv_host's weight stream, restated: slabs of cs8 channel groups; f16 / f16x3 streams hold K-chunks of 4 (tap, group)
    units (1 or 2 planes of nf KiB), f16m8 streams pieces of 8 units (4 * nf KiB); bridged layers (f16x3 / f16m8, uniform slabs, units per slab
    not a multiple of 4 / 8, 3x3(x3) taps) run a slab's units on from where the slab before stopped and pad only the last slab."""
    groups = -(-cin // 8)
    slabs = [min(cs8, groups - g) for g in range(0, groups, cs8)]
    um = 8 if split >= 2 else 4
    bridged = ks == 3 and split in (1, 2, 3) and len(slabs) >= 2 and all(c == cs8 for c in slabs) and (taps * cs8) % um != 0
    total = 0
    for si, c8n in enumerate(slabs):
        gu, o, b = taps * c8n, 0, 0
        if bridged:
            o = (si * ((um - gu % um) % um)) % um
            b = 0 if si + 1 == len(slabs) else (um - (gu - o) % um) % um
        total += -(-(gu - o + b) // um) * (um // 4)                      # in K-chunks
    per_chunk = nf * 1024 if split >= 2 else nf * 512 * (2 if split == 1 else 1)        # halfs per chunk (f16m8: 4 nf KiB per 2-chunk piece)
    return total * per_chunk * nsplit, bridged, total


@pytest.mark.parametrize("cin,cout,ks,k2d,nf,nsplit,cs8,split,chunks", [
    (32, 32, 3, 0, 2, 1, 1, 1, 27),        # conv1_2, f16x3: 4 slabs of 27 units = 27 chunks (28 padded)
    (300, 300, 3, 0, 5, 4, 1, 1, 257),     # conv4_2 on three fp16 MFMAs (f16x3p): 38 slabs -> ceil(38 * 27 / 4)
    (300, 300, 3, 0, 5, 4, 1, 3, 258),     # conv4_2, f16m8e (fp8 codes): 38 slabs = ceil(38 * 27 / 8) = 129 pieces = 258 chunks
    (64, 100, 3, 0, 7, 1, 1, 2, 54),       # merge_conv_a, f16m8: 8 slabs = 27 pieces (32 padded) = 54 chunks
    (100, 100, 3, 0, 7, 1, 1, 2, 88),      # merge_conv_b: 13 slabs = 44 pieces (52 padded)
    (64, 64, 3, 1, 4, 1, 2, 1, 18),        # similarityNet 64 -> 64, f16x3: 4 two-group slabs = 18 chunks (20 padded)
    (64, 64, 3, 1, 4, 1, 4, 0, 18),        # ... its f16 mode: 4-group slabs, 9 chunks each, no bridge needed
    (6, 32, 3, 0, 2, 1, 1, 1, 7),          # conv1_1: one slab, padded as ever
    (80, 16, 1, 0, 1, 1, 5, 1, 4),         # a 1x1x1 side conv: never bridged (two 5-group slabs = 1.25 chunks each, padded to 2 + 2)
])
def test_packed_weight_stream_length_with_bridge_chunks(built, cin, cout, ks, k2d, nf, nsplit, cs8, split, chunks):
    import ctypes
    import numpy as np
    dbg = ctypes.CDLL(os.path.join(os.path.dirname(built.LIB_PATH), "libsurfacenet_hip_dbg.so"))
    pack = dbg.sn_debug_pack_host
    pack.restype = ctypes.c_int
    pack.argtypes = [ctypes.c_int] * 9 + [ctypes.c_void_p] * 6
    taps = ks * ks * (1 if k2d else ks)
    rs = np.random.RandomState(1)
    W = rs.randn(cout, cin, taps).astype(np.float32)
    one = np.ones(cout, np.float32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    out = (ctypes.c_ulonglong * 4)()
    assert pack(cin, cout, ks, 1, k2d, nf, nsplit, cs8, split, P(W), P(one), P(one), P(one), P(one), out) == 0
    halfs, bridged, total = _packed_halfs(cin, ks, taps, cs8, split, nf, nsplit)
    assert total == chunks and bridged == (ks == 3 and split != 0 and cin > 8), (total, chunks, bridged)
    assert out[0] == halfs, (out[0], halfs, bridged)


def test_ab_switches_exist_in_the_test_twin_only():
    """The schedule / packing A/B switches (SN_NO_BRIDGE, SN_NO_EPI_FUSION, SN_M8_TAIL, SN_MX_S_ACT ...) are compiled into the test-only twin
    library alone: the product library does not even contain their names, so no environment variable can change its K order or code format
    (ADVICE r3)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prod = open(os.path.join(root, "surfacenet_amd", "libsurfacenet_hip.so"), "rb").read()
    dbg = open(os.path.join(root, "surfacenet_amd", "libsurfacenet_hip_dbg.so"), "rb").read()
    for name in (b"SN_NO_BRIDGE", b"SN_SIMIL_NO_BRIDGE", b"SN_NO_EPI_FUSION", b"SN_UPSAMPLE_PER_VOXEL", b"SN_M8_TAIL", b"SN_MX_S_ACT", b"SN_MX_S_CAT", b"SN_C4_M8"):
        assert name in dbg, name
        assert name not in prod, name
