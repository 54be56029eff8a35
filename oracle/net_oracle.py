"""oracle/net_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's 3D-CNN surface regressor + view-pair fusion, used only as the
checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

PARITY UNPINNED: the reference network (nets/SurfaceNet.py, nets/layers.py) hard-imports
Theano 0.9.0 / Lasagne@7992faa / cuDNN 5.1 (SurfaceNet.py:2, layers.py:7-8), none of which exist
in this image or under /root/reference, the reference ships no tests for it and no weights
(inputs/SurfaceNet_models/README.txt). This file restates the published semantics of those
libraries at the reference's call sites; it is cross-checked only against a second, independent
numpy formulation in this same file (`forward_numpy`) and hand-derivable known-answer tests
(tests/test_oracle_net.py).

Follows (reference file:line):
  nets/SurfaceNet.py:18-76    __1viewPair_SurfaceNet__ topology (layer table LAYERS below)
  nets/SurfaceNet.py:126      reshape (-1, N_vp, s,s,s)
  nets/SurfaceNet.py:341-357  fusion: weighted average for N_vp>=2, identity for N_vp==1
  nets/SurfaceNet.py:84-100   __relativeWeight_net__ (258 -> 100 BN sigmoid -> 1, grouped softmax)
  nets/layers.py:228-253      DilatedConv3DLayer: out[b,o,p] = sum_c sum_t in[b,c,p+2t] W[c,o,t]
  nets/layers.py:325-336      ChannelPool_weightedAverage: sum_p pred[n,p]*w[n,p]/sum_p w[n,p]
  nets/layers.py:363-390      Bilinear_3DInterpolation: zero-insert upscale + fixed k^3 conv
Lasagne semantics used: batch_norm() drops the conv bias and applies, at deterministic=True,
  y = act((conv(x) - mean) * (gamma * inv_std) + beta); Conv3DDNNLayer is cross-correlation
  (flip_filters=False), pad='same' = k//2 zeros; Pool3DDNNLayer((2,2,2), stride=2) = max, floor.
Parameter order = lasagne.layers.get_all_param_values([output_SurfaceNet_reshape,
  output_softmaxWeights]) (SurfaceNet.py:397-400): see PARAM_LAYOUT.
"""
import numpy as np

# (name, kind, c_in, c_out, act) ; kind: conv3 = 3x3x3 pad 1, conv1 = 1x1x1, dil3 = 3x3x3 dilation 2
# (weights stored (C_in, C_out, ...)), dil1 = 1x1x1 through the dilated layer (same storage), up = fixed
LAYERS = [
    ("conv1_1", "conv3", 6, 32, "relu"), ("conv1_2", "conv3", 32, 32, "relu"), ("conv1_3", "conv3", 32, 32, "relu"),
    ("side_op1", "conv1", 32, 16, "sigmoid"),
    ("conv2_1", "conv3", 32, 80, "relu"), ("conv2_2", "conv3", 80, 80, "relu"), ("conv2_3", "conv3", 80, 80, "relu"),
    ("side_op2", "conv1", 80, 16, "sigmoid"), ("side_op2_deconv", "up", 3, 2, None),
    ("conv3_1", "conv3", 80, 160, "relu"), ("conv3_2", "conv3", 160, 160, "relu"), ("conv3_3", "conv3", 160, 160, "relu"),
    ("side_op3", "conv1", 160, 16, "sigmoid"), ("side_op3_deconv", "up", 5, 4, None),
    ("conv4_1", "dil3", 160, 300, "relu"), ("conv4_2", "dil3", 300, 300, "relu"), ("conv4_3", "dil3", 300, 300, "relu"),
    ("side_op4", "dil1", 300, 16, "sigmoid"), ("side_op4_deconv", "up", 5, 4, None),
    ("merge_conv_a", "conv3", 64, 100, "relu"), ("merge_conv_b", "conv3", 100, 100, "relu"),
    ("merge_conv3", "conv1", 100, 1, "sigmoid"),
]
D_FEATURE, N_HIDDEN = 258, 100  # params.py:99-101


def param_layout():
    """[(layer, param, shape)] in the reference pickle's order (105 arrays)."""
    out = []
    for name, kind, a, b, _ in LAYERS:
        if kind == "up":
            out.append((name, "W", (1, 1, a, a, a)))
            continue
        k = 3 if kind in ("conv3", "dil3") else 1
        shape = (b, a, k, k, k) if kind in ("conv3", "conv1") else (a, b, k, k, k)
        out.append((name, "W", shape))
        for p in ("beta", "gamma", "mean", "inv_std"):
            out.append((name, p, (b,)))
    out.append(("feature_fc1", "W", (D_FEATURE, N_HIDDEN)))
    for p in ("beta", "gamma", "mean", "inv_std"):
        out.append(("feature_fc1", p, (N_HIDDEN,)))
    out.append(("feature_linear1", "W", (N_HIDDEN, 1)))
    out.append(("feature_linear1", "b", (1,)))
    return out


PARAM_LAYOUT = param_layout()


def params_to_dict(values):
    assert len(values) == len(PARAM_LAYOUT), (len(values), len(PARAM_LAYOUT))
    d = {}
    for (layer, p, shape), v in zip(PARAM_LAYOUT, values):
        v = np.asarray(v)
        assert tuple(v.shape) == tuple(shape), (layer, p, v.shape, shape)
        d.setdefault(layer, {})[p] = v
    return d


def w5d(size):
    """layers.py:363-374 __W_5D__ restated (returns the 1-D factor; the kernel is its outer cube)."""
    size = float(size)
    factor = (size + 1) // 2
    center = factor - 1 if size % 2 == 1 else factor - 0.5
    t = np.arange(int(size), dtype=np.float64)
    return 1 - np.abs(t - center) / factor


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ------------------------------------------------------------------------------------------------
# Formulation 1: torch CPU conv3d (the oracle proper). dtype float64 by default.
# quant = None            -> exact restatement
# quant = "fp16"          -> emulates the HIP path's storage precision: weights and every stored
#                            activation are rounded to IEEE half, products accumulate wide, BN affine +
#                            activation are applied in fp32 before the rounding
#                            (the MX-assisted modes of the HIP path are modelled in oracle/net_emulation.py)
# ------------------------------------------------------------------------------------------------
def forward_torch(X, values, w=None, n_vp=1, quant=None, dtype="float64", return_intermediates=False):
    import torch
    import torch.nn.functional as F
    td = getattr(torch, dtype)
    P = params_to_dict(values)

    def q(t):
        if quant == "fp16":
            return t.to(torch.float16).to(td)
        return t

    def tw(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(td)
        return q(t)

    def bn_act(y, p, act):
        scale = (p["gamma"].astype(np.float64) * p["inv_std"].astype(np.float64))
        shift = p["beta"].astype(np.float64) - p["mean"].astype(np.float64) * scale
        if quant == "fp16":  # the device applies fp32 scale/shift in the conv epilogue
            scale = scale.astype(np.float32); shift = shift.astype(np.float32)
            y = y.to(torch.float32)
            y = y * torch.from_numpy(scale).view(1, -1, 1, 1, 1) + torch.from_numpy(shift).view(1, -1, 1, 1, 1)
        else:
            y = y * torch.from_numpy(scale).to(td).view(1, -1, 1, 1, 1) + torch.from_numpy(shift).to(td).view(1, -1, 1, 1, 1)
        y = torch.relu(y) if act == "relu" else torch.sigmoid(y)
        return y.to(td)

    def conv(x, name, kind, act, store=True):
        p = P[name]
        W = p["W"]
        if kind in ("dil3", "dil1"):
            W = np.transpose(W, (1, 0, 2, 3, 4))  # stored (C_in, C_out, ...) — layers.py:200-213
        k = W.shape[2]
        def cv(xx, ww):
            if kind == "dil3":
                return F.conv3d(F.pad(xx, (2,) * 6), ww, dilation=2)  # PadLayer(2) + 'valid' dilated conv
            return F.conv3d(xx, ww, padding=k // 2)

        y = cv(x, tw(W))
        y = bn_act(y, p, act)
        return q(y) if store else y

    def up(x, name, f):
        k = P[name]["W"].shape[2]
        Wk = torch.from_numpy(np.ascontiguousarray(P[name]["W"])).to(td)
        B, C = x.shape[:2]
        z = torch.zeros((B, C, x.shape[2] * f, x.shape[3] * f, x.shape[4] * f), dtype=td)
        z[:, :, ::f, ::f, ::f] = x  # Upscale3DLayer(mode='dilate')
        y = F.conv3d(z.reshape(B * C, 1, *z.shape[2:]), Wk, padding=k // 2)
        return q(y.reshape(B, C, *z.shape[2:]))

    inter = {}
    x = q(torch.from_numpy(np.ascontiguousarray(X)).to(td))
    c11 = conv(x, "conv1_1", "conv3", "relu"); c12 = conv(c11, "conv1_2", "conv3", "relu"); c13 = conv(c12, "conv1_3", "conv3", "relu")
    p1 = F.max_pool3d(c13, 2, 2)
    s1 = conv(c13, "side_op1", "conv1", "sigmoid")
    c21 = conv(p1, "conv2_1", "conv3", "relu"); c22 = conv(c21, "conv2_2", "conv3", "relu"); c23 = conv(c22, "conv2_3", "conv3", "relu")
    p2 = F.max_pool3d(c23, 2, 2)
    s2 = up(conv(c23, "side_op2", "conv1", "sigmoid"), "side_op2_deconv", 2)
    c31 = conv(p2, "conv3_1", "conv3", "relu"); c32 = conv(c31, "conv3_2", "conv3", "relu"); c33 = conv(c32, "conv3_3", "conv3", "relu")
    s3 = up(conv(c33, "side_op3", "conv1", "sigmoid"), "side_op3_deconv", 4)
    c41 = conv(c33, "conv4_1", "dil3", "relu"); c42 = conv(c41, "conv4_2", "dil3", "relu"); c43 = conv(c42, "conv4_3", "dil3", "relu")
    s4 = up(conv(c43, "side_op4", "dil1", "sigmoid"), "side_op4_deconv", 4)
    cat = torch.cat([s1, s2, s3, s4], dim=1)
    ma = conv(cat, "merge_conv_a", "conv3", "relu")
    mb = conv(ma, "merge_conv_b", "conv3", "relu", store=(quant is None))  # device keeps merge_b in fp32 registers
    out = conv(mb, "merge_conv3", "conv1", "sigmoid", store=False)
    if return_intermediates:
        inter = dict(conv1_1=c11, conv1_3=c13, pool1=p1, side1=s1, conv2_3=c23, side2=s2, conv3_3=c33, side3=s3,
                     conv4_1=c41, conv4_3=c43, side4=s4, cat=cat, merge_a=ma, merge_b=mb)
        inter = {k: v.numpy() for k, v in inter.items()}
    unf = out.numpy().astype(np.float64)
    s = X.shape[-1]
    unfused = unf.reshape(-1, n_vp, s, s, s)
    fused = fuse(unfused, w, n_vp)
    if return_intermediates:
        return fused, unfused, inter
    return fused, unfused


def fuse(unfused, w, n_vp):
    """SurfaceNet.py:341-357 / layers.py:325-336."""
    if n_vp == 1:
        return unfused.copy()
    w = np.asarray(w, dtype=np.float32).astype(np.float64)  # T.matrix is floatX=float32
    cw = w / w.sum(axis=1, keepdims=True)
    return (unfused * cw[:, :, None, None, None]).sum(axis=1, keepdims=True)


# ------------------------------------------------------------------------------------------------
# Formulation 2: independent numpy (explicit shifted sums, closed-form App. D upsampler). Slow;
# small cases only. Exists so two formulations written differently must agree.
# ------------------------------------------------------------------------------------------------
def _conv_np(x, W, dil=1):
    """x (B,C,D,D,D), W (O,C,k,k,k) cross-correlation, zero 'same' padding."""
    B, C, D = x.shape[0], x.shape[1], x.shape[2]
    k = W.shape[2]
    r = (k // 2) * dil
    xp = np.zeros((B, C, D + 2 * r, D + 2 * r, D + 2 * r), dtype=np.float64)
    xp[:, :, r:r + D, r:r + D, r:r + D] = x
    y = np.zeros((B, W.shape[0], D, D, D), dtype=np.float64)
    for a in range(k):
        for b in range(k):
            for c in range(k):
                y += np.einsum("bcxyz,oc->boxyz", xp[:, :, a * dil:a * dil + D, b * dil:b * dil + D, c * dil:c * dil + D],
                               W[:, :, a, b, c].astype(np.float64))
    return y


def _up_axis(x, f, axis):
    """App. D closed form of zero-insert + [1/2,1,1/2] (f=2) or [1/3,2/3,1,2/3,1/3] (f=4) along one axis."""
    x = np.moveaxis(x, axis, -1)
    n = x.shape[-1]
    nxt = np.concatenate([x[..., 1:], np.zeros_like(x[..., :1])], axis=-1)
    y = np.zeros(x.shape[:-1] + (n * f,), dtype=np.float64)
    if f == 2:
        y[..., 0::2] = x
        y[..., 1::2] = 0.5 * (x + nxt)
    else:
        y[..., 0::4] = x
        y[..., 1::4] = (2.0 / 3.0) * x
        y[..., 2::4] = (1.0 / 3.0) * (x + nxt)
        y[..., 3::4] = (2.0 / 3.0) * nxt
    return np.moveaxis(y, -1, axis)


def forward_numpy(X, values, w=None, n_vp=1):
    P = params_to_dict(values)

    def bn_act(y, p, act):
        scale = p["gamma"].astype(np.float64) * p["inv_std"].astype(np.float64)
        y = (y - p["mean"].astype(np.float64)[None, :, None, None, None]) * scale[None, :, None, None, None] \
            + p["beta"].astype(np.float64)[None, :, None, None, None]
        return np.maximum(y, 0) if act == "relu" else _sigmoid(y)

    def conv(x, name, kind, act):
        W = P[name]["W"]
        if kind in ("dil3", "dil1"):
            W = np.transpose(W, (1, 0, 2, 3, 4))
        return bn_act(_conv_np(x, W, 2 if kind == "dil3" else 1), P[name], act)

    def pool(x):
        B, C, D = x.shape[:3]
        return x.reshape(B, C, D // 2, 2, D // 2, 2, D // 2, 2).max(axis=(3, 5, 7))

    def up(x, f):
        for ax in (2, 3, 4):
            x = _up_axis(x, f, ax)
        return x

    x = np.asarray(X, dtype=np.float64)
    c13 = conv(conv(conv(x, "conv1_1", "conv3", "relu"), "conv1_2", "conv3", "relu"), "conv1_3", "conv3", "relu")
    s1 = conv(c13, "side_op1", "conv1", "sigmoid")
    c23 = conv(conv(conv(pool(c13), "conv2_1", "conv3", "relu"), "conv2_2", "conv3", "relu"), "conv2_3", "conv3", "relu")
    s2 = up(conv(c23, "side_op2", "conv1", "sigmoid"), 2)
    c33 = conv(conv(conv(pool(c23), "conv3_1", "conv3", "relu"), "conv3_2", "conv3", "relu"), "conv3_3", "conv3", "relu")
    s3 = up(conv(c33, "side_op3", "conv1", "sigmoid"), 4)
    c43 = conv(conv(conv(c33, "conv4_1", "dil3", "relu"), "conv4_2", "dil3", "relu"), "conv4_3", "dil3", "relu")
    s4 = up(conv(c43, "side_op4", "dil1", "sigmoid"), 4)
    cat = np.concatenate([s1, s2, s3, s4], axis=1)
    mb = conv(conv(cat, "merge_conv_a", "conv3", "relu"), "merge_conv_b", "conv3", "relu")
    out = conv(mb, "merge_conv3", "conv1", "sigmoid")
    s = X.shape[-1]
    unfused = out.reshape(-1, n_vp, s, s, s)
    return fuse(unfused, w, n_vp), unfused


def relative_weights(features, values, n_vp):
    """SurfaceNet.py:84-100 / :334-338: (n*P, 258) -> softmax over groups of n_vp -> (n, n_vp)."""
    P = params_to_dict(values)
    fc, lin = P["feature_fc1"], P["feature_linear1"]
    h = np.asarray(features, dtype=np.float64) @ fc["W"].astype(np.float64)
    scale = fc["gamma"].astype(np.float64) * fc["inv_std"].astype(np.float64)
    h = _sigmoid((h - fc["mean"]) * scale + fc["beta"])
    z = (h @ lin["W"].astype(np.float64) + lin["b"].astype(np.float64)).reshape(-1, n_vp)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)
