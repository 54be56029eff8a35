"""CPU: the address-sanitizer twin of the library's HOST code (`make asan`: -fsanitize=address on the host side, device code unchanged, debug
hooks on) runs what can run without a GPU — the fragment packer for every layer shape of both networks in every precision mode, the 6-bit
encoder, the context's failure path — under LD_PRELOAD of the sanitizer runtime, in a subprocess. Any heap / stack / global overflow or
use-after-free in that code aborts the subprocess with an AddressSanitizer report (the sanitizer build the reference lacks)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "surfacenet_amd", "csrc")
LIB = os.path.join(ROOT, "surfacenet_amd", "libsurfacenet_hip_asan.so")

DRIVER = r'''
import ctypes, sys
import numpy as np
lib = ctypes.CDLL(sys.argv[1])
lib.sn_last_error.restype = ctypes.c_char_p
assert lib.sn_version() == 2
lib.sn_create.restype = ctypes.c_void_p
ctx = lib.sn_create(0, 32, 8)                       # no GPU here: must fail cleanly (message, no crash); on a GPU box it succeeds
if ctx:
    lib.sn_destroy.argtypes = [ctypes.c_void_p]; lib.sn_destroy(ctx)
else:
    assert lib.sn_last_error()
enc = lib.sn_debug_mx6_encode
enc.restype = ctypes.c_int; enc.argtypes = [ctypes.c_float, ctypes.c_int]
for fmt in (2, 3):
    for v in np.concatenate([np.linspace(-40, 40, 2001), [0.0, 1e-30, -1e30, np.inf, -np.inf, np.nan]]):
        assert 0 <= enc(float(v), fmt) < 64
pack = lib.sn_debug_pack_host
pack.restype = ctypes.c_int
pack.argtypes = [ctypes.c_int] * 9 + [ctypes.c_void_p] * 6
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
rs = np.random.RandomState(0)
# (cin, cout, ks, dil, k2d, nf, nsplit): every conv layer family of SurfaceNet (sn_api.hip tile_for) and of the similarityNet (sn_simil.hip)
net = [(6, 32, 3, 1, 0, 2, 1), (32, 32, 3, 1, 0, 2, 1), (32, 16, 1, 1, 0, 1, 1), (32, 80, 3, 1, 0, 5, 1), (80, 80, 3, 1, 0, 5, 1), (80, 16, 1, 1, 0, 1, 1),
       (80, 160, 3, 1, 0, 5, 2), (160, 160, 3, 1, 0, 5, 2), (160, 16, 1, 1, 0, 1, 1), (160, 300, 3, 2, 0, 5, 4), (300, 300, 3, 2, 0, 5, 4), (300, 16, 1, 1, 0, 1, 1),
       (64, 100, 3, 1, 0, 7, 1), (100, 100, 3, 1, 0, 7, 1)]
sim = [(3, 64, 3, 1, 1, 4, 1), (64, 64, 3, 1, 1, 4, 1), (64, 128, 3, 1, 1, 4, 2), (256, 512, 3, 1, 1, 4, 8), (512, 512, 3, 1, 1, 8, 4)]
n_calls = 0
for cin, cout, ks, dil, k2d, nf, nsplit in net + sim:
    taps = ks * ks * (1 if k2d else ks)
    W = (rs.randn(cout, cin, taps) * 10.0 ** rs.uniform(-6, 4)).astype(np.float32)          # any magnitude: the packer renormalises by powers of two
    bn = [rs.uniform(0.5, 1.5, cout).astype(np.float32) for _ in range(4)]
    for split in (0, 1, 2, 3):
        if split >= 2 and (ks == 1 or k2d):
            continue                                                                          # f16m8 / f16m8e (fp8 codes: the dilated layers) exist for the 3x3x3 kernels only
        for cs8max in ((5,) if ks == 1 else ((2, 4) if k2d else (1, 2))):              # the slab widths the kernels are instantiated with
            out = (ctypes.c_ulonglong * 4)()
            rc = pack(cin, cout, ks, dil, k2d, nf, nsplit, cs8max, split, P(W), P(bn[0]), P(bn[1]), P(bn[2]), P(bn[3]), out)
            assert rc == 0, (cin, cout, ks, split, cs8max, lib.sn_last_error())
            again = (ctypes.c_ulonglong * 4)()
            assert pack(cin, cout, ks, dil, k2d, nf, nsplit, cs8max, split, P(W), P(bn[0]), P(bn[1]), P(bn[2]), P(bn[3]), again) == 0
            assert list(out) == list(again) and out[0] > 0 and out[1] == nsplit * nf * 16 + 16           # deterministic, sized as the kernels expect
            n_calls += 1
bad = np.full((16, 8, 27), np.nan, np.float32)                                               # a non-finite weight is refused with a message, not packed
one = np.ones(16, np.float32)
assert pack(8, 16, 3, 1, 0, 1, 1, 1, 1, P(bad), P(one), P(one), P(one), P(one), (ctypes.c_ulonglong * 4)()) != 0 and b"non-finite" in lib.sn_last_error()
print("ASAN-DRIVER-OK", n_calls)
'''


@pytest.fixture(scope="module")
def asan_lib():
    subprocess.check_call(["make", "-j6", "-C", CSRC, "asan"], stdout=subprocess.DEVNULL)
    rt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()
    if not os.path.exists(rt):
        pytest.skip("no address-sanitizer runtime in this toolchain")
    return rt


def test_host_code_under_address_sanitizer(asan_lib):
    env = dict(os.environ, LD_PRELOAD=asan_lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1")
    p = subprocess.run([sys.executable, "-c", DRIVER, LIB], env=env, capture_output=True, text=True, timeout=900)
    assert "AddressSanitizer" not in p.stderr, p.stderr[-3000:]
    assert p.returncode == 0 and "ASAN-DRIVER-OK" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-2000:])
    assert int(p.stdout.split("ASAN-DRIVER-OK")[1].split()[0]) >= 80          # packer runs: layer shapes x precision modes x slab widths
