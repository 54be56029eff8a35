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
