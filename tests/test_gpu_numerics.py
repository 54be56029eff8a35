"""GPU (-m gpu): the dynamic range the parity claim rests on. The default arithmetic stores weights and activations as hi+lo pairs
of fp16 numbers (fp16's exponent range); sn_load_weights therefore renormalises every layer by exact powers of two (pack_conv) and
the conv epilogues raise a status bit when a value still leaves the fp16 range. These tests take a BN-calibrated net and apply
transformations that leave the network FUNCTION unchanged (so the fp64 oracle is the same reference) but move weights / activations
across many orders of magnitude; the HIP result must stay within the parity tolerance, or fail loudly (never silently)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 2e-4             # north-star bar 1e-3; same tolerance as the tame nets of test_gpu_parity.py
CONSUMERS = {"conv1_1": ["conv1_2"], "conv1_2": ["conv1_3"], "conv1_3": ["side_op1", "conv2_1"], "conv2_1": ["conv2_2"], "conv2_2": ["conv2_3"],
             "conv2_3": ["side_op2", "conv3_1"], "conv3_1": ["conv3_2"], "conv3_2": ["conv3_3"], "conv3_3": ["side_op3", "conv4_1"],
             "conv4_1": ["conv4_2"], "conv4_2": ["conv4_3"], "conv4_3": ["side_op4"], "merge_conv_a": ["merge_conv_b"], "merge_conv_b": ["merge_conv3"]}
CIN_FIRST = {"conv4_1", "conv4_2", "conv4_3", "side_op4"}           # DilatedConv3DLayer stores W as (C_in, C_out, k,k,k) (nets/layers.py:200-213)


@pytest.fixture(scope="module")
def sn(gpu_required):
    import surfacenet_amd
    return surfacenet_amd


def _index():
    from surfacenet_amd import weights
    return {(layer, p): i for i, (layer, p, _) in enumerate(weights.PARAM_LAYOUT)}


def _case(seed=1, s=16, n=2, n_vp=2):
    import synth
    values = [np.array(v) for v in synth.calibrated_params(seed)]
    X = synth.random_cvc(n * n_vp, s, seed + 20)
    w = (np.random.RandomState(seed).rand(n, n_vp) + 0.1).astype(np.float32)
    return values, X, w, s, n, n_vp


def _scale_channels(values, layer, f):
    """y_c -> f_c * y_c for ReLU layer `layer` (gamma, beta scaled), undone in the weights of its consumers: same function."""
    ix = _index()
    f = np.asarray(f, dtype=np.float64)
    for p in ("gamma", "beta"):
        values[ix[(layer, p)]] = (values[ix[(layer, p)]].astype(np.float64) * f).astype(np.float32)
    for cons in CONSUMERS[layer]:
        W = values[ix[(cons, "W")]].astype(np.float64)
        shape = [1] * W.ndim
        shape[0 if cons in CIN_FIRST else 1] = -1
        values[ix[(cons, "W")]] = (W / f.reshape(shape)).astype(np.float32)


def _scale_weights(values, layer, g):
    """W -> g * W with the BN statistics of the layer following (mean *= g, inv_std /= g): same function."""
    ix = _index()
    values[ix[(layer, "W")]] = (values[ix[(layer, "W")]].astype(np.float64) * g).astype(np.float32)
    values[ix[(layer, "mean")]] = (values[ix[(layer, "mean")]].astype(np.float64) * g).astype(np.float32)
    values[ix[(layer, "inv_std")]] = (values[ix[(layer, "inv_std")]].astype(np.float64) / g).astype(np.float32)


def _run(sn, values, X, w, s, n_vp, precision="f16x3"):
    with sn.Context(cube_D=s, max_samples=4, precision=precision) as ctx:
        ctx.load_param_values(values)
        return ctx.forward(X, w, n_vp=n_vp)


def _oracle(values, X, w, n_vp):
    from oracle import net_oracle
    return net_oracle.forward_torch(X, values, w=w, n_vp=n_vp)


def test_bn_scales_spanning_six_decades(sn):
    """Per-channel BatchNorm scales from 1e-3 to 1e3 in EVERY ReLU layer (activations of one tensor span six decades)."""
    values, X, w, s, n, n_vp = _case(1)
    f64, u64 = _oracle(values, X, w, n_vp)                                   # the function before the transformation
    rs = np.random.RandomState(7)
    ix = _index()
    for layer in CONSUMERS:
        cout = values[ix[(layer, "gamma")]].shape[0]
        _scale_channels(values, layer, 10.0 ** rs.uniform(-3, 3, cout))
    f64b, u64b = _oracle(values, X, w, n_vp)
    assert np.abs(u64b - u64).max() < 1e-6                                   # (fp32 storage of the transformed parameters)
    fused, unfused = _run(sn, values, X, w, s, n_vp)
    err = np.abs(unfused - u64b).max()
    print("BN scales 1e-3..1e3: L_inf %.3e" % err)
    assert err < TOL and np.abs(fused - f64b).max() < TOL


def test_tiny_and_huge_weights(sn):
    """Weights of 1e-6 (conv1_x, merge_conv_a), 1e+4 (conv2_x, conv4_x) with the BN statistics that go with them."""
    values, X, w, s, n, n_vp = _case(2)
    for layer, g in (("conv1_1", 1e-6), ("conv1_2", 1e-6), ("conv1_3", 3e-7), ("conv2_1", 1e4), ("conv2_2", 2e4), ("conv4_2", 1e4),
                     ("merge_conv_a", 1e-6), ("merge_conv_b", 5e3), ("side_op3", 1e-5)):
        _scale_weights(values, layer, g)
    f64, u64 = _oracle(values, X, w, n_vp)
    assert u64.std() > 0.05
    fused, unfused = _run(sn, values, X, w, s, n_vp)
    err = np.abs(unfused - u64).max()
    print("weights 1e-6 / 1e4: L_inf %.3e" % err)
    assert err < TOL and np.abs(fused - f64).max() < TOL


def test_dead_and_saturated_channels(sn):
    """All-zero channels (gamma = beta = 0), channels that never fire (beta << 0) and saturated sigmoid side outputs."""
    values, X, w, s, n, n_vp = _case(0)
    ix = _index()
    for layer, dead in (("conv1_2", [0, 5, 31]), ("conv2_2", list(range(0, 80, 7))), ("conv4_1", list(range(3, 300, 11))), ("merge_conv_a", [1, 50, 99])):
        for p in ("gamma", "beta"):
            values[ix[(layer, p)]][dead] = 0.0
    values[ix[("conv3_2", "beta")]][::5] = -50.0                            # ReLU never fires
    values[ix[("side_op1", "gamma")]][:4] *= 60.0                           # sigmoid saturates at 0 / 1
    values[ix[("side_op2", "beta")]][:3] = 40.0
    values[ix[("side_op4", "beta")]][-3:] = -40.0
    f64, u64 = _oracle(values, X, w, n_vp)
    fused, unfused = _run(sn, values, X, w, s, n_vp)
    err = np.abs(unfused - u64).max()
    print("dead / saturated channels: L_inf %.3e" % err)
    assert np.isfinite(unfused).all() and err < TOL and np.abs(fused - f64).max() < TOL


@pytest.mark.parametrize("precision", ["f16x3", "f16x3p", "f16m8", "f16"])
def test_overflow_fails_loudly(sn, precision):
    """BatchNorm statistics that do not match the data (inv_std 1e7 too large in one channel): the activation leaves the fp16 range of
    its storage format. The result would be inf / NaN downstream: the call must raise, naming the layer - never return numbers."""
    values, X, w, s, n, n_vp = _case(1)
    ix = _index()
    values[ix[("conv1_2", "inv_std")]][3] *= 1e7
    with pytest.raises(sn.SurfaceNetHipError, match="conv1_2"):
        _run(sn, values, X, w, s, n_vp, precision=precision)
    values, X, w, s, n, n_vp = _case(1)
    values[ix[("merge_conv_b", "mean")]][10] = np.float32(-3e38)            # fp32 overflow inside the folded shift -> rejected at load time
    values[ix[("merge_conv_b", "inv_std")]][10] = np.float32(1e30)
    with pytest.raises(sn.SurfaceNetHipError):
        _run(sn, values, X, w, s, n_vp, precision=precision)
    # the context stays usable after a flagged call
    values, X, w, s, n, n_vp = _case(1)
    with sn.Context(cube_D=s, max_samples=4, precision=precision) as ctx:
        bad = [np.array(v) for v in values]
        bad[ix[("conv2_3", "inv_std")]][0] *= 1e8
        ctx.load_param_values(bad)
        with pytest.raises(sn.SurfaceNetHipError, match="conv2_3"):
            ctx.forward(X, w, n_vp=n_vp)
        ctx.load_param_values(values)
        fused, unfused = ctx.forward(X, w, n_vp=n_vp)
        assert np.isfinite(unfused).all()


def test_interpolation_stencil_pinned_on_device(sn):
    """sn_load_weights accepts exactly the reference's __W_5D__ arrays (tests/golden/w5d_cases.npz, produced by executing
    nets/layers.py:361-372) for the three upsamplers and refuses anything else (the kernel implements their closed form)."""
    import os
    from surfacenet_amd import weights
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "w5d_cases.npz"))
    values = [np.array(v) for v in weights.synthetic_param_values(3)]
    ix = _index()
    values[ix[("side_op2_deconv", "W")]] = G["f2_W"].copy()
    values[ix[("side_op3_deconv", "W")]] = G["f4_W"].copy()
    values[ix[("side_op4_deconv", "W")]] = G["f4_W"].copy()
    with sn.Context(cube_D=8, max_samples=2) as ctx:
        ctx.load_param_values(values)
        values[ix[("side_op3_deconv", "W")]][0, 0, 2, 2, 1] += 1e-3
        with pytest.raises(sn.SurfaceNetHipError, match="interpolation kernel"):
            ctx.load_param_values(values)


def _ma_saturation(ctx, n_samples, s):
    """Fraction of merge_conv_a's stored outputs (the one ReLU tensor that is kept as 6-bit codes, premultiplier 2^0) whose fp16 value exceeds
    the e2m3 range 7.5, i.e. whose hi code saturates - read back through the test-only twin library's sn_debug_tensor."""
    import ctypes
    import os
    from surfacenet_amd import _lib
    dbg = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libsurfacenet_hip_dbg.so"))
    dbg.sn_debug_tensor.restype = ctypes.c_int
    dbg.sn_debug_tensor.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
    buf = np.empty((n_samples, 13, s ** 3, 8), dtype=np.float16)                 # hi plane, [sample][group of 8 of the 104 channels][voxel][8]
    assert dbg.sn_debug_tensor(ctx._h, b"ma", buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
    real = buf.transpose(0, 1, 3, 2).reshape(n_samples, 104, -1)[:, :100]
    return float((real > 7.5).mean()), float(real.max())


def _structured_inputs(s, seed):
    """Network inputs with the statistics real views have and noise lacks: constant regions, a step edge, a smooth field with a heavy tail,
    and cubes that no view sees (CVC.py:42-46 leaves zeros, i.e. -mean after the preprocess)."""
    import golden_util
    mean = golden_util.MEAN6[None, :, None, None, None]
    rs = np.random.RandomState(seed)
    out = {}
    out["all_out_of_view"] = np.zeros((4, 6, s, s, s), np.float32) - mean
    const = np.empty((4, 6, s, s, s), np.float32)
    const[:] = np.asarray([200, 30, 90, 198, 33, 92], np.float32)[None, :, None, None, None]
    out["constant_colour"] = const - mean
    edge = np.empty((4, 6, s, s, s), np.float32)
    edge[:, :, : s // 2] = np.asarray([250, 250, 250, 5, 5, 5], np.float32)[None, :, None, None, None]
    edge[:, :, s // 2:] = np.asarray([0, 10, 20, 255, 240, 230], np.float32)[None, :, None, None, None]
    edge[2:] = np.swapaxes(edge[2:], 2, 4)                                       # the edge along z for two of the samples
    out["step_edge"] = edge - mean
    coarse = rs.rand(4, 6, s // 4, s // 4, s // 4).astype(np.float32)
    smooth = np.repeat(np.repeat(np.repeat(coarse, 4, axis=2), 4, axis=3), 4, axis=4) * 120 + 60
    tail = rs.rand(*smooth.shape) < 0.01
    smooth[tail] = np.where(rs.rand(int(tail.sum())) < 0.5, 0.0, 255.0)          # 1 % saturated / black pixels
    out["smooth_heavy_tail"] = np.rint(smooth).astype(np.float32) - mean
    return out


def test_structured_inputs_and_code_saturation(sn):
    """The default mode keeps merge_conv_a's input and output as 6-bit e2m3 codes with static premultipliers: values above
    7.5 * 2^-s saturate and lose (only) their own correction term. Every other parity test feeds i.i.d. noise; this one feeds the inputs real
    scenes produce, and a net whose merge_conv_a outputs are pushed past the code range in a few % of the voxels (BatchNorm statistics that
    under-estimate the spread 3.2-fold - function unchanged, so the fp64 oracle is the same reference). Records the saturated fraction."""
    values, _, w, s, n, n_vp = _case(2)
    ix = _index()
    for name, X in _structured_inputs(s, 5).items():
        with sn.Context(cube_D=s, max_samples=4) as ctx:
            ctx.load_param_values(values)
            try:
                fused, unfused = ctx.forward(X, w, n_vp=n_vp)
            except sn.SurfaceNetHipError as e:
                print("   %-18s loud failure: %s" % (name, str(e)[:120]))
                continue
            sat, mx = _ma_saturation(ctx, 4, s)
        f64, u64 = _oracle(values, X, w, n_vp)
        e_u = float(np.abs(unfused - u64).max())
        print("   %-18s L_inf vs fp64 oracle %.3e   merge_conv_a outputs above the 6-bit range: %.4f %% (max stored %.1f)   probabilities %.3f .. %.3f"
              % (name, e_u, 100 * sat, mx, u64.min(), u64.max()))
        assert e_u < TOL and np.abs(fused - f64).max() < TOL
    import synth
    X = synth.random_cvc(4, s, 31)
    for spread in (3.2, 10.0):
        # BatchNorm statistics of merge_conv_a that under-estimate the spread of its pre-activations (function unchanged: same fp64 oracle)
        wide = [np.array(v) for v in values]
        wide[ix[("merge_conv_a", "inv_std")]] *= np.float32(spread)
        f64, u64 = _oracle(wide, X, w, n_vp)
        with sn.Context(cube_D=s, max_samples=4) as ctx:
            ctx.load_param_values(wide)
            assert ctx.numeric_status() == []
            fused, unfused = ctx.forward(X, w, n_vp=n_vp)
            sat, mx = _ma_saturation(ctx, 4, s)
            warn = ctx.numeric_status()                               # the PRODUCT library's warning word (no test hook)
            e_static = float(np.abs(unfused - u64).max())
            cal = ctx.calibrate(4, max_sat_fraction=1e-3)             # premultipliers from the activations that forward left behind
            assert ctx.numeric_status() == []
            fused2, unfused2 = ctx.forward(X, w, n_vp=n_vp)
            warn2 = ctx.numeric_status()
            e_cal = float(np.abs(unfused2 - u64).max())
        print("   merge_conv_a spread x%.1f: static exponents: L_inf %.3e, %.2f %% of its outputs above the 6-bit range (max stored %.1f), warning %s | "
              "calibrated (s_act %d -> %d, s_cat %d -> %d; saturated %.3f %% -> %.3f %%): L_inf %.3e, warning %s"
              % (spread, e_static, 100 * sat, mx, warn, cal["s_act_before"], cal["s_act"], cal["s_cat_before"], cal["s_cat"], 100 * cal["sat_act_before"],
                 100 * cal["sat_act"], e_cal, warn2))
        assert sat > 0.005, "the stress case must really saturate codes"
        assert warn == ["merge_conv_a"], "saturating codes must be visible through the product library"
        # (the calibration's fractions are of the NON-ZERO stored values - ReLU zeroes about half - the test hook's of all of them)
        assert sat <= cal["sat_act_before"] <= 3 * sat and abs(cal["max_act"] - mx) < 0.51 and cal["s_act"] < cal["s_act_before"] and cal["sat_act"] <= 1e-3
        if spread < 5:
            assert e_static < 1e-3                                    # uncalibrated: graceful, still inside the north-star bar (measured 2.0e-4) ...
            assert e_cal < TOL, "with data-driven premultipliers the x3.2 stress case meets the default mode's own tolerance (measured 1.6e-4)"
        else:
            # x10: 29 % of the codes saturate under the static exponent and the result leaves the bar (measured 1.35e-3) - not silently: the
            # warning above names the layer; calibrated it is back inside (measured 3.8e-4, 0.03 % of the values still beyond the range)
            assert e_cal < 1e-3


def test_nan_and_negative_overflow_fail_loudly(sn):
    """A NaN input voxel, and an accumulator driven to -inf (a BatchNorm scale of -1e37 in fp32): ReLU maps both to a clean 0, so the status
    check has to look at the pre-activation. The call must raise, never return numbers."""
    values, X, w, s, n, n_vp = _case(1)
    Xn = X.copy()
    Xn[1, 3, 4, 5, 6] = np.nan
    with pytest.raises(sn.SurfaceNetHipError, match="conv1_1"):
        _run(sn, values, Xn, w, s, n_vp)
    ix = _index()
    bad = [np.array(v) for v in values]
    bad[ix[("conv3_2", "gamma")]][7] = np.float32(-1.0)
    bad[ix[("conv3_2", "mean")]][7] = np.float32(0.0)
    bad[ix[("conv3_2", "inv_std")]][7] = np.float32(3e38)       # folded scale finite (-3e38); accumulator * scale = -+inf: -inf -> ReLU -> 0
    with pytest.raises(sn.SurfaceNetHipError, match="conv3_2"):
        _run(sn, bad, X, w, s, n_vp)


def test_non_finite_batchnorm_constants_are_rejected_at_load(sn):
    """The stored-value tracker of the store epilogues is a packed MAX, which drops a quiet NaN: a stored NaN whose accumulator was
    finite - sigmoid(NaN) from a NaN folded scale / shift of a side convolution - would not raise the status bit. It cannot get that far: every
    non-finite folded BatchNorm constant fails sn_load_weights, naming the layer."""
    values, X, w, s, n, n_vp = _case(1)
    ix = _index()
    for layer, param, val in (("side_op1", "gamma", np.nan), ("side_op3", "beta", np.nan), ("conv2_2", "inv_std", np.inf), ("merge_conv3", "mean", np.nan)):
        bad = [np.array(v) for v in values]
        bad[ix[(layer, param)]][0] = np.float32(val)
        with sn.Context(cube_D=s, max_samples=4) as ctx:
            with pytest.raises((sn.SurfaceNetHipError, ValueError), match=layer):      # (weights.to_blob rejects a non-positive / non-finite inv_std itself)
                ctx.load_param_values(bad)


def test_calibrate_refuses_stale_or_foreign_activations(sn):
    """sn_calibrate_dev scans what the LAST forward call left in the workspace: it must refuse a fresh context (a zeroed workspace would
    set both exponents to +6), more samples than that call ran (older data), and the all-MX mode, whose single activation exponent also belongs to
    tensors the pooling / upsampling kernels decode with the static one."""
    values, X, w, s, n, n_vp = _case(1)
    with sn.Context(cube_D=s, max_samples=8) as ctx:
        ctx.load_param_values(values)
        with pytest.raises(sn.SurfaceNetHipError, match="no forward call"):
            ctx.calibrate(2)
        ctx.forward(X[:2], None, n_vp=1)
        with pytest.raises(sn.SurfaceNetHipError, match="ran 2 samples"):
            ctx.calibrate(4)
        cal = ctx.calibrate(2)
        assert -6 <= cal["s_act"] <= 6 and -6 <= cal["s_cat"] <= 6
        ctx.load_param_values(values)                                  # new weights: the workspace is stale again
        with pytest.raises(sn.SurfaceNetHipError, match="no forward call"):
            ctx.calibrate(2)
    with sn.Context(cube_D=s, max_samples=8, precision="f16m8") as ctx:
        ctx.load_param_values(values)
        ctx.forward(X[:2], None, n_vp=1)
        with pytest.raises(sn.SurfaceNetHipError, match="default precision mode"):
            ctx.calibrate(2)


@pytest.mark.parametrize("spread", [80.0, 300.0])
def test_fp8_code_planes_warn_where_their_lo_codes_can_first_saturate(sn, spread):
    """The conv4 chain's tensors carry fp8 e4m3 code planes (round 5): nothing to calibrate, but a stored value beyond 256 * 2^-s can saturate its lo code
    (lo * 2^12 up to 512 > 448) and loses its own correction term - the layer's WARNING bit must say so (limit 256, not the hi code's 448: ADVICE r5),
    the error word must stay clear (the values are far inside fp16), the drop-in guard names the layer ONCE and offers the opt-out, and
    `conv4_fp8=False` (sn_set_conv4_fp8) runs the same net without the plane, hence without the warning and at the default tolerance. BatchNorm statistics
    of conv4_1 that under-estimate its spread `spread`-fold push its outputs there (the same values go to the fp64 oracle). x80: stored values of 256 .. 448
    (lo codes at risk only) - the result stays inside the bar; x300: values beyond 448, where the HI codes saturate too and the correction term
    `w_lo * x_hi` is wrong by the clipped amount: outside the bar (measured 1.2e-2) - which is exactly what the warning is for."""
    import warnings
    from surfacenet_amd.context import NumericsGuard
    values, X, w, s, n, n_vp = _case(1)
    ix = _index()
    values[ix[("conv4_1", "inv_std")]] = values[ix[("conv4_1", "inv_std")]] * np.float32(spread)
    f64, u64 = _oracle(values, X, w, n_vp)
    with sn.Context(cube_D=s, max_samples=4) as ctx:
        ctx.load_param_values(values)
        assert ctx.numeric_status() == []
        fused, unfused = ctx.forward(X, w, n_vp=n_vp)                     # no SN_ERR_RANGE: stored values of a few hundred are ordinary fp16 numbers
        names = ctx.numeric_status()
        assert "conv4_1" in names, names
        guard = NumericsGuard(ctx)
        ctx.forward(X, w, n_vp=n_vp)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            assert guard.check("test") is None                            # nothing to calibrate, nothing to redo
            ctx.forward(X, w, n_vp=n_vp)
            assert guard.check("test") is None
        msgs = [str(r.message) for r in rec if "fp8" in str(r.message)]
        assert len(msgs) == 1 and "conv4_1" in msgs[0] and "conv4_fp8=False" in msgs[0], [str(r.message) for r in rec]
    e = float(np.abs(unfused - u64).max())
    with sn.Context(cube_D=s, max_samples=4, conv4_fp8=False) as ctx:
        ctx.load_param_values(values)
        fused3, unfused3 = ctx.forward(X, w, n_vp=n_vp)
        assert not [n_ for n_ in ctx.numeric_status() if n_.startswith("conv4") or n_ == "conv3_3"]
    e3 = float(np.abs(unfused3 - u64).max())
    print("conv4_1 outputs x%.0f out: L_inf vs fp64 oracle %.3e with the fp8 step (warned: %s), %.3e with conv4_fp8=False" % (spread, e, names, e3))
    assert e3 < TOL
    if spread < 100:
        assert e < 1e-3
