"""Weight handling for the MI355X SurfaceNet path.

* `PARAM_LAYOUT`            the 105 arrays of the reference weight file in its order
                            (`lasagne.layers.set_all_param_values([output_SurfaceNet_reshape,
                            output_softmaxWeights], values)`, nets/SurfaceNet.py:397-400)
* `load_lasagne_pickle`     reads the reference's py2 pickle (`*.model`, params.py:91,106)
* `synthetic_param_values`  seeded random-init weights of the same architecture (the pre-trained file is
                            a Dropbox download, inputs/SurfaceNet_models/README.txt — not available offline)
* `to_blob`                 flat float32 blob + sn_param_desc table for `sn_load_weights`

BN folding and the fp16 MFMA-fragment packing are done natively inside libsurfacenet_hip.so.
"""
import pickle

import numpy as np

from . import _lib

# (name, kind, a, b): conv3/conv1 -> W (b, a, k,k,k); dil3/dil1 -> W (a, b, k,k,k) (nets/layers.py:200-213);
# up -> fixed W (1,1,a,a,a), b = upscale factor (nets/layers.py:376-390)
NET_LAYERS = [
    ("conv1_1", "conv3", 6, 32), ("conv1_2", "conv3", 32, 32), ("conv1_3", "conv3", 32, 32), ("side_op1", "conv1", 32, 16),
    ("conv2_1", "conv3", 32, 80), ("conv2_2", "conv3", 80, 80), ("conv2_3", "conv3", 80, 80), ("side_op2", "conv1", 80, 16),
    ("side_op2_deconv", "up", 3, 2),
    ("conv3_1", "conv3", 80, 160), ("conv3_2", "conv3", 160, 160), ("conv3_3", "conv3", 160, 160), ("side_op3", "conv1", 160, 16),
    ("side_op3_deconv", "up", 5, 4),
    ("conv4_1", "dil3", 160, 300), ("conv4_2", "dil3", 300, 300), ("conv4_3", "dil3", 300, 300), ("side_op4", "dil1", 300, 16),
    ("side_op4_deconv", "up", 5, 4),
    ("merge_conv_a", "conv3", 64, 100), ("merge_conv_b", "conv3", 100, 100), ("merge_conv3", "conv1", 100, 1),
]
D_VIEWPAIR_FEATURE = 258   # params.py:99
N_HIDDEN = 100             # params.py:100
BN_PARAMS = ("beta", "gamma", "mean", "inv_std")   # Lasagne BatchNormLayer creation order


def _layout():
    out = []
    for name, kind, a, b in NET_LAYERS:
        if kind == "up":
            out.append((name, "W", (1, 1, a, a, a)))
            continue
        k = 3 if kind.endswith("3") else 1
        out.append((name, "W", (b, a, k, k, k) if kind.startswith("conv") else (a, b, k, k, k)))
        out.extend((name, p, (b,)) for p in BN_PARAMS)
    out.append(("feature_fc1", "W", (D_VIEWPAIR_FEATURE, N_HIDDEN)))
    out.extend(("feature_fc1", p, (N_HIDDEN,)) for p in BN_PARAMS)
    out.append(("feature_linear1", "W", (N_HIDDEN, 1)))
    out.append(("feature_linear1", "b", (1,)))
    return out


PARAM_LAYOUT = _layout()
N_NET_PARAMS = 98


def interpolation_kernel(k):
    """The fixed 'bilinear' stencil of nets/layers.py:363-374 for kernel size k (3 or 5)."""
    factor = (k + 1) // 2
    center = factor - 1 if k % 2 == 1 else factor - 0.5
    w1 = 1.0 - np.abs(np.arange(k) - center) / factor
    return (w1[:, None, None] * w1[None, :, None] * w1[None, None, :])[None, None].astype(np.float32)


def validate(values):
    if len(values) not in (N_NET_PARAMS, len(PARAM_LAYOUT)):
        raise ValueError("expected %d or %d parameter arrays, got %d" % (N_NET_PARAMS, len(PARAM_LAYOUT), len(values)))
    for (layer, p, shape), v in zip(PARAM_LAYOUT, values):
        if tuple(np.shape(v)) != tuple(shape):
            raise ValueError("%s.%s: expected shape %s, got %s" % (layer, p, shape, np.shape(v)))
        # The four BatchNorm vectors of a layer have the same shape, so the shape check cannot tell them apart. Lasagne's BatchNormLayer
        # registers beta, gamma, mean, inv_std in that order and inv_std = 1/sqrt(var + eps) is strictly positive and finite - the one
        # property that distinguishes a slot: a file written in another order (e.g. mean and inv_std swapped) fails here, loudly.
        if p == "inv_std" and not (np.all(np.isfinite(v)) and np.all(np.asarray(v) > 0)):
            raise ValueError("%s.inv_std must be finite and > 0 (got min %g): the file's BatchNorm vectors are not in Lasagne's "
                             "(beta, gamma, mean, inv_std) order" % (layer, float(np.min(v))))


def _load_py2_pickle(path):
    """A Python-2 pickle of a flat list of numpy arrays, protocol 0 (`pickle.dump(values, f)`, ASCII - what survives the reference's
    text-mode `open(model_file)`, nets/SurfaceNet.py:398) or protocol 1 / 2: py2 `str` payloads need encoding='latin1' under Python 3,
    and numpy turns the latin-1 text of an array's data string back into bytes."""
    with open(path, "rb") as f:
        values = pickle.load(f, encoding="latin1")
    if not isinstance(values, (list, tuple)):
        raise ValueError("%s: expected a pickled list of arrays (lasagne.layers.get_all_param_values), got %s" % (path, type(values).__name__))
    return values


def load_lasagne_pickle(path):
    """Reads the reference's `*.model` file: a Python-2 pickle of a flat list of numpy arrays."""
    values = _load_py2_pickle(path)
    values = [np.asarray(v, dtype=np.float32) for v in values]
    validate(values)
    return values


def synthetic_param_values(seed=0, with_relative_weight_net=True):
    """Seeded random weights with BN statistics chosen analytically so activations stay O(1) and the
    sigmoids are not saturated (stand-in for the unavailable trained model)."""
    rs = np.random.RandomState(seed)
    values = []
    # rough second moment of each layer's input, tracked analytically
    in_ms = {"conv1_1": 75.0 ** 2}
    post_relu_ms = 0.30

    def conv_params(name, shape, fan_in, ms_in):
        bound = np.sqrt(3.0 / fan_in)          # var(W) = 1/fan_in
        W = rs.uniform(-bound, bound, size=shape).astype(np.float32)
        cout = shape[0] if name not in ("conv4_1", "conv4_2", "conv4_3", "side_op4") else shape[1]
        std = np.sqrt(ms_in)                    # std of the pre-BN conv output
        beta = rs.uniform(-0.3, 0.3, cout).astype(np.float32)
        gamma = rs.uniform(0.7, 1.3, cout).astype(np.float32)
        mean = (rs.uniform(-0.2, 0.2, cout) * std).astype(np.float32)
        inv_std = (rs.uniform(0.8, 1.25, cout) / std).astype(np.float32)
        return [W, beta, gamma, mean, inv_std]

    for name, kind, a, b in NET_LAYERS:
        if kind == "up":
            values.append(interpolation_kernel(a))
            continue
        k = 3 if kind.endswith("3") else 1
        shape = (b, a, k, k, k) if kind.startswith("conv") else (a, b, k, k, k)
        if name == "conv1_1":
            ms = in_ms["conv1_1"]
        elif name == "merge_conv_a":
            ms = 0.14         # sigmoid side outputs, partly attenuated by the upsamplers
        else:
            ms = post_relu_ms
        if name in ("conv2_1", "conv3_1"):
            ms = 0.6          # max-pooled ReLU maps are larger
        values.extend(conv_params(name, shape, a * k ** 3, ms))
    if with_relative_weight_net:
        b1 = np.sqrt(6.0 / (D_VIEWPAIR_FEATURE + N_HIDDEN))
        values.append(rs.uniform(-b1, b1, (D_VIEWPAIR_FEATURE, N_HIDDEN)).astype(np.float32))
        values.append(rs.uniform(-0.3, 0.3, N_HIDDEN).astype(np.float32))
        values.append(rs.uniform(0.7, 1.3, N_HIDDEN).astype(np.float32))
        values.append(rs.uniform(-0.2, 0.2, N_HIDDEN).astype(np.float32))
        values.append(rs.uniform(0.8, 1.25, N_HIDDEN).astype(np.float32))
        b2 = np.sqrt(6.0 / (N_HIDDEN + 1))
        values.append(rs.uniform(-b2, b2, (N_HIDDEN, 1)).astype(np.float32))
        values.append(np.zeros(1, dtype=np.float32))
    validate(values)
    return values


def to_blob(values):
    """-> (blob float32 1-D, (ParamDesc * n) array) for sn_load_weights."""
    validate(values)
    return _pack_blob(values)


def _pack_blob(values):
    descs = (_lib.ParamDesc * len(values))()
    chunks, off = [], 0
    for i, v in enumerate(values):
        v = np.ascontiguousarray(v, dtype=np.float32)
        descs[i].offset = off
        descs[i].ndim = v.ndim
        for j in range(5):
            descs[i].shape[j] = v.shape[j] if j < v.ndim else 0
        chunks.append(v.ravel())
        off += v.size
    return np.concatenate(chunks), descs


# ---- similarityNet (nets/similarityNet.py:23-77): 13 x (W (Cout,Cin,3,3), b), embedding W (5888,128), b, similarity W (1,1), b ----
SIMIL_CONVS = [("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128), ("conv3_1", 128, 256),
               ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512),
               ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512)]
D_SIMIL_FEATURE = 512 * 4 + (64 + 128 + 256 + 512) * 4     # pool5 flattened + 2x2 centre crops of pool1..4 = 5888
D_EMBEDDING = 128                                           # params.py:88
SIMIL_PARAM_SHAPES = [sh for _, ci, co in SIMIL_CONVS for sh in ((co, ci, 3, 3), (co,))] + \
    [(D_SIMIL_FEATURE, D_EMBEDDING), (D_EMBEDDING,), (1, 1), (1,)]


def validate_simil(values):
    if len(values) != len(SIMIL_PARAM_SHAPES):
        raise ValueError("similarityNet weight file must hold %d arrays, got %d" % (len(SIMIL_PARAM_SHAPES), len(values)))
    for i, (v, sh) in enumerate(zip(values, SIMIL_PARAM_SHAPES)):
        if tuple(np.shape(v)) != sh:
            raise ValueError("similarityNet param %d: expected shape %s, got %s" % (i, sh, np.shape(v)))


def load_simil_pickle(path):
    """The reference's similarityNet `*.model` file (nets/similarityNet.py:240-242): py2 pickle of a flat list of arrays."""
    values = _load_py2_pickle(path)
    values = [np.asarray(v, dtype=np.float32) for v in values]
    validate_simil(values)
    return values


def synthetic_simil_param_values(seed=0):
    """Seeded He-style weights: activations stay O(1) from conv1_1 on (inputs are pixel - mean, rms ~75), embeddings are
    O(0.3) per component and the pair similarity is not saturated. Stand-in for the unavailable trained model."""
    rs = np.random.RandomState(1000 + seed)
    values = []
    for i, (_, ci, co) in enumerate(SIMIL_CONVS):
        bound = np.sqrt(6.0 / (ci * 9)) / (75.0 if i == 0 else 1.0)
        values.append(rs.uniform(-bound, bound, size=(co, ci, 3, 3)).astype(np.float32))
        values.append(rs.uniform(-0.1, 0.1, size=(co,)).astype(np.float32))
    bound = 25.0 * np.sqrt(3.0 / D_SIMIL_FEATURE)
    values.append(rs.uniform(-bound, bound, size=(D_SIMIL_FEATURE, D_EMBEDDING)).astype(np.float32))
    values.append(rs.uniform(-0.1, 0.1, size=(D_EMBEDDING,)).astype(np.float32))
    values.append(np.asarray([[rs.uniform(0.5, 1.5)]], dtype=np.float32))
    values.append(np.asarray([rs.uniform(-2.0, -1.0)], dtype=np.float32))
    return values


def simil_to_blob(values):
    validate_simil(values)
    return _pack_blob(values)
