"""GPU (-m gpu): the HIP path, called through the C ABI, against the golden vectors and the oracle."""
import numpy as np
import pytest

import golden_util

pytestmark = pytest.mark.gpu

CASES = golden_util.cvc_cases()


@pytest.fixture(scope="module")
def sn(gpu_required):
    import surfacenet_amd
    return surfacenet_amd


def _ctx_for_case(sn, c, max_samples=16):
    ctx = sn.Context(cube_D=int(c["s"]), max_samples=max_samples)
    ctx.set_cameras(c["P"])
    ctx.set_images(golden_util.case_images(c))
    return ctx


@pytest.mark.parametrize("name", sorted(CASES))
def test_cvc_warp_bit_exact_vs_reference_golden(sn, name):
    c = CASES[name]
    with _ctx_for_case(sn, c) as ctx:
        out = ctx.cvc(c["pairs"], c["xyz"], c["resol"])
    assert out.dtype == np.float32 and out.shape == c["out_u8"].shape
    assert np.array_equal(out, c["out_u8"].astype(np.float32))


def test_cvc_preprocess_golden_and_chunking(sn):
    c = CASES["dtu_s8_vp1"]
    with _ctx_for_case(sn, c, max_samples=2) as ctx:      # 3 samples through a 2-sample workspace -> chunked
        out = ctx.cvc(c["pairs"], c["xyz"], c["resol"], mean=golden_util.MEAN6)
    assert np.array_equal(out, c["pre_f32"])


def test_cvc_bad_view_index(sn):
    c = CASES["dtu_s8_vp1"]
    bad = c["pairs"].copy(); bad[0, 0, 0] = 4
    with _ctx_for_case(sn, c) as ctx:
        with pytest.raises(sn.SurfaceNetHipError):
            ctx.cvc(bad, c["xyz"], c["resol"])
        neg = c["pairs"].copy(); neg[0, 0, 0] = -4     # numpy-style negative index == view 0
        a = ctx.cvc(neg, c["xyz"], c["resol"])
        pos = c["pairs"].copy(); pos[0, 0, 0] = 0
        assert np.array_equal(a, ctx.cvc(pos, c["xyz"], c["resol"]))


def _net_case(s, n, n_vp, seed):
    from surfacenet_amd import weights
    values = weights.synthetic_param_values(seed)
    rs = np.random.RandomState(seed + 10)
    X = rs.randint(0, 256, (n * n_vp, 6, s, s, s)).astype(np.float32) - golden_util.MEAN6[None, :, None, None, None]
    w = (rs.rand(n, n_vp) + 0.1).astype(np.float32)
    return values, X, w


@pytest.mark.parametrize("s,n,n_vp", [(8, 3, 1), (16, 2, 2), (32, 1, 3)])
def test_forward_vs_oracle(sn, s, n, n_vp):
    from oracle import net_oracle
    values, X, w = _net_case(s, n, n_vp, seed=s)
    with sn.Context(cube_D=s, max_samples=4) as ctx:       # forces chunking for n*n_vp > 4
        ctx.load_param_values(values)
        fused, unfused = ctx.forward(X, w if n_vp > 1 else None, n_vp=n_vp)
    assert fused.shape == (n, 1, s, s, s) and unfused.shape == (n, n_vp, s, s, s)
    f64, u64 = net_oracle.forward_torch(X, values, w=w, n_vp=n_vp)
    f16, u16 = net_oracle.forward_torch(X, values, w=w, n_vp=n_vp, quant="fp16")
    e_emu = np.abs(unfused - u16).max()
    e_ref = np.abs(unfused - u64).max()
    print("s=%d Linf vs fp16-emulating oracle %.3e, vs fp64 oracle %.3e, fused %.3e" % (s, e_emu, e_ref, np.abs(fused - f64).max()))
    assert e_emu < 3e-4          # same arithmetic, different summation order / rare rounding flips
    assert e_ref < 5e-3
    assert np.abs(fused - f16).max() < 3e-4
    if n_vp == 1:
        assert np.array_equal(fused, unfused)


def test_cvc_forward_fused_path(sn):
    from oracle import cvc_oracle, net_oracle
    from surfacenet_amd import weights
    s, n, n_vp = 16, 3, 2
    sc = golden_util.synthetic_scene(n, n_vp, s=s, seed=5, hw=(600, 800))
    sc["xyz"][1] = [-150.0, -102.0, 638.0]
    values = weights.synthetic_param_values(1)
    with sn.Context(cube_D=s, max_samples=4) as ctx:
        ctx.load_param_values(values)
        ctx.set_cameras(sc["cams"]); ctx.set_images(sc["imgs"])
        fused, unfused, cvc = ctx.cvc_forward(sc["pairs"], sc["xyz"], sc["resol"], sc["w"], return_cvc=True)
        f2, u2 = ctx.forward(cvc, sc["w"], n_vp=n_vp)
    ref_cvc = cvc_oracle.gen_coloredCubes(sc["pairs"], sc["xyz"], sc["resol"], sc["cams"], sc["imgs"], s, mean6=golden_util.MEAN6)
    assert np.array_equal(cvc, ref_cvc)
    assert np.array_equal(fused, f2) and np.array_equal(unfused, u2)      # fused entry == 3-call protocol
    f16, u16 = net_oracle.forward_torch(ref_cvc, values, w=sc["w"], n_vp=n_vp, quant="fp16")
    assert np.abs(unfused - u16).max() < 3e-4 and np.abs(fused - f16).max() < 3e-4
