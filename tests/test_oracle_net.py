"""CPU: the CNN oracle (parity UNPINNED by the reference — no Theano/Lasagne/cuDNN here, no reference tests, no weights):
two independently written formulations must agree, plus hand-derivable known-answer tests."""
import numpy as np
import pytest

from oracle import net_oracle as no
from surfacenet_amd import weights


def test_param_layout_matches_product_and_survey():
    assert len(no.PARAM_LAYOUT) == 105 and no.PARAM_LAYOUT == weights.PARAM_LAYOUT
    shapes = {(l, p): s for l, p, s in no.PARAM_LAYOUT}
    assert shapes[("conv4_1", "W")] == (160, 300, 3, 3, 3)       # (C_in, C_out, ...) — nets/layers.py:200-213
    assert shapes[("side_op4", "W")] == (300, 16, 1, 1, 1)
    assert shapes[("merge_conv_a", "W")] == (100, 64, 3, 3, 3)
    assert shapes[("side_op3_deconv", "W")] == (1, 1, 5, 5, 5)
    n_w = sum(int(np.prod(s)) for (l, p, s) in no.PARAM_LAYOUT[:98] if p == "W")
    assert n_w == 8811529                                         # weight elements incl. fixed stencils


def test_two_formulations_agree():
    vals = weights.synthetic_param_values(3)
    rs = np.random.RandomState(0)
    X = rs.randn(2, 6, 8, 8, 8) * 60
    w = np.array([[0.2, 0.9]], dtype=np.float32)
    f1, u1 = no.forward_torch(X, vals, w=w, n_vp=2)
    f2, u2 = no.forward_numpy(X, vals, w=w, n_vp=2)
    # the only intended difference: forward_torch convolves with the reference's float32 stencil (__W_5D__ returns
    # float32: 1/3 -> 0.33333334), forward_numpy uses the exact closed form -> ~1e-8 on the upsampled sigmoids
    assert np.abs(u1 - u2).max() < 1e-7 and np.abs(f1 - f2).max() < 1e-7
    f3, u3 = no.forward_torch(X.astype(np.float32), vals, w=w, n_vp=2, dtype="float32")
    assert np.abs(u3 - u1).max() < 2e-5


def test_interpolation_stencil_known_answers():
    assert np.allclose(no.w5d(3), [0.5, 1, 0.5]) and np.allclose(no.w5d(5), [1 / 3, 2 / 3, 1, 2 / 3, 1 / 3])
    assert np.allclose(weights.interpolation_kernel(5)[0, 0, 2, 2], no.w5d(5))
    d = np.zeros((1, 1, 4, 1, 1)); d[0, 0, 1] = 1.0
    assert np.allclose(no._up_axis(d, 2, 2)[0, 0, :, 0, 0], [0, .5, 1, .5, 0, 0, 0, 0])
    assert np.allclose(no._up_axis(d, 4, 2)[0, 0, :, 0, 0], [0, 0, 1 / 3, 2 / 3, 1, 2 / 3, 1 / 3, 0] + [0] * 8)
    # zero-insert + 'same' cross-correlation with the reference's kernel == the closed form
    import torch, torch.nn.functional as F
    x = torch.rand(1, 1, 4, 4, 4, dtype=torch.float64)
    for f, k in ((2, 3), (4, 5)):
        z = torch.zeros(1, 1, 4 * f, 4 * f, 4 * f, dtype=torch.float64); z[:, :, ::f, ::f, ::f] = x
        ref = F.conv3d(z, torch.from_numpy(weights.interpolation_kernel(k)).double(), padding=k // 2).numpy()
        y = x.numpy()
        for ax in (2, 3, 4):
            y = no._up_axis(y, f, ax)
        assert np.allclose(ref, y, atol=1e-7)


def test_dilated_conv_one_hot_is_a_shift():
    x = np.random.RandomState(1).rand(1, 2, 6, 6, 6)
    W = np.zeros((1, 2, 3, 3, 3)); W[0, 1, 2, 1, 0] = 1.0      # tap offset (+1, 0, -1) * dilation 2 on channel 1
    y = no._conv_np(x, W, dil=2)
    exp = np.zeros((6, 6, 6)); exp[0:4, :, 2:6] = x[0, 1, 2:6, :, 0:4]
    assert np.allclose(y[0, 0], exp)


def test_fusion_semantics():
    u = np.random.RandomState(2).rand(3, 2, 4, 4, 4)
    assert np.allclose(no.fuse(u, np.array([[1, 0]] * 3, np.float32), 2)[:, 0], u[:, 0])
    w = np.random.RandomState(3).rand(3, 2).astype(np.float32) + 0.1
    assert np.allclose(no.fuse(u, w, 2), no.fuse(u, 7.5 * w, 2), atol=1e-7)          # layers.py:330-331 renormalises
    u1 = u[:, :1]
    assert np.array_equal(no.fuse(u1, None, 1), u1)                                   # N_vp == 1: identity


def test_relative_weights_softmax():
    vals = weights.synthetic_param_values(0)
    f = np.random.RandomState(4).rand(6, 258).astype(np.float32)
    sm = no.relative_weights(f, vals, 3)
    assert sm.shape == (2, 3) and np.allclose(sm.sum(axis=1), 1.0) and (sm > 0).all()


def test_device_arithmetic_model_stays_close_to_the_oracle():
    """oracle/net_emulation.py (the CPU model of the HIP path's roundings, used by the GPU parity tests as a second, tighter reference):
    on a BN-calibrated net it must sit where the design says — far inside the 1e-3 bar, the all-MX mode behind the default."""
    import synth
    from oracle import net_emulation
    values = list(synth.calibrated_params(2))
    X = synth.random_cvc(2, 8, 18)
    f64, u64 = no.forward_torch(X, values, n_vp=1)
    e = {}
    for mode in ("f16x3", "f16m8"):
        fe, ue = net_emulation.forward_emulated(X, values, n_vp=1, mode=mode)
        assert ue.shape == u64.shape
        e[mode] = np.abs(ue - u64).max()
    assert 1e-7 < e["f16x3"] < 1e-4 and e["f16x3"] < e["f16m8"] < 5e-4, e
