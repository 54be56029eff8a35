"""oracle/simil_oracle.py — TEST INFRASTRUCTURE (checker only; never imported by the product).

CPU restatement of the similarityNet / early-rejection stage:

    crop_patches      utils/image.py:92-183 cropImgPatches at pyramidRate = 1 (the call form of utils/earlyRejection.py:50)
    preprocess        utils/image.py:9-36   preprocess_patches
    embedding_torch   nets/similarityNet.py:23-58  __input_var_TO_embedding_layer__ (torch CPU conv2d, float64 by default)
    embedding_numpy   the same network in plain numpy (independent formulation, small batches)
    pair_similarity   nets/similarityNet.py:71-77  DistanceLayer(Lp=2) + DenseLayer(1, sigmoid)

PARITY of the numpy/indexing parts is PINNED: tests/golden/simil_cases.npz holds outputs of the reference's own
cropImgPatches / preprocess_patches / img_hw_cubesCorner_inScopeCheck / perspectiveProj_cubesCorner / patch2embedding /
embeddingPairs2simil / selectFromSimilarity executed in the build container (oracle/gen_golden_simil.py).
PARITY of the NETWORK is UNPINNED: its arithmetic lives in Theano/Lasagne/cuDNN (absent), the reference has neither a
test nor weights for it. Layer semantics restated from the call sites:
  * ConvLayer = lasagne.layers.dnn.Conv2DDNNLayer when cuDNN is present (similarityNet.py:6-9): W (Cout,Cin,3,3), pad=1,
    CROSS-CORRELATION (flip_filters=False is that layer's default), + b, rectify (Lasagne's default nonlinearity).
  * Pool2DLayer(2): max, stride 2. CropFeatureMapCenterLayer(r=1): rows/cols [H/2-1, H/2+1), flattened (c,h,w)
    (nets/layers.py:73-78). ConcatLayer order: pool5 flatten, crops of pool1, pool2, pool3, pool4 (similarityNet.py:49-55).
  * L2NormLayer: x / sqrt(sum x^2) per row (layers.py:38-42). embedding = DenseLayer(128, nonlinearity=None): x.W + b.
  * pair similarity: sigmoid(W * ((sum |e1-e2|^2) ** 0.5) + b) (layers.py:130-138, similarityNet.py:76).
"""
import numpy as np

CONVS = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256), (256, 512), (512, 512), (512, 512),
         (512, 512), (512, 512), (512, 512)]
POOL_AFTER = {1, 3, 6, 9, 12}          # index of the last conv of each block


def crop_patches(img, center_h, center_w, patchSize=64):
    """(n, patchSize, patchSize, c) patches of img (h,w,c): top-left = trunc(centre) - patchSize/2, coordinates clamped."""
    img = np.asarray(img)
    H, W = img.shape[:2]
    r = patchSize // 2
    h0 = np.asarray(center_h, dtype=np.float64).astype(np.int64) - r
    w0 = np.asarray(center_w, dtype=np.float64).astype(np.int64) - r
    hh = np.clip(h0[:, None] + np.arange(patchSize)[None, :], 0, H - 1)
    ww = np.clip(w0[:, None] + np.arange(patchSize)[None, :], 0, W - 1)
    return img[hh[:, :, None], ww[:, None, :], :]


def preprocess(patches, mean_BGR):
    """(n,h,w,3) RGB any dtype -> (n,3,h,w) float32 BGR - mean."""
    x = np.asarray(patches).astype(np.float32)
    x = np.transpose(x, (0, 3, 1, 2))[:, ::-1]
    return np.ascontiguousarray(x - np.asarray(mean_BGR, dtype=np.float32)[None, :, None, None])


def _features(pools):
    """pools: list of 5 arrays (n,C,H,H) -> (n,5888) concat in the reference's order."""
    n = pools[0].shape[0]
    crop = lambda p: p[:, :, p.shape[2] // 2 - 1: p.shape[2] // 2 + 1, p.shape[3] // 2 - 1: p.shape[3] // 2 + 1].reshape(n, -1)
    return np.concatenate([pools[4].reshape(n, -1), crop(pools[0]), crop(pools[1]), crop(pools[2]), crop(pools[3])], axis=1)


def embedding_torch(X, values, dtype="float64", return_pools=False):
    import torch
    import torch.nn.functional as F
    dt = getattr(torch, dtype)
    x = torch.from_numpy(np.ascontiguousarray(X)).to(dt)
    pools = []
    with torch.no_grad():
        for i in range(13):
            W = torch.from_numpy(np.asarray(values[2 * i])).to(dt)
            b = torch.from_numpy(np.asarray(values[2 * i + 1])).to(dt)
            x = F.relu(F.conv2d(x, W, b, padding=1))
            if i in POOL_AFTER:
                x = F.max_pool2d(x, 2)
                pools.append(x.numpy().copy())
    f = _features(pools).astype(np.float64)
    f = f / np.sqrt((f ** 2).sum(axis=1))[:, None]
    emb = f @ np.asarray(values[26], dtype=np.float64) + np.asarray(values[27], dtype=np.float64)
    return (emb, pools) if return_pools else emb


def embedding_numpy(X, values):
    """Independent formulation: 3x3 cross-correlation as 9 shifted tensordots, float64."""
    x = np.asarray(X, dtype=np.float64)
    pools = []
    for i in range(13):
        W = np.asarray(values[2 * i], dtype=np.float64)
        b = np.asarray(values[2 * i + 1], dtype=np.float64)
        n, c, H, Wd = x.shape
        xp = np.zeros((n, c, H + 2, Wd + 2))
        xp[:, :, 1:-1, 1:-1] = x
        y = np.zeros((n, W.shape[0], H, Wd))
        for dy in range(3):
            for dx in range(3):
                y += np.einsum("nchw,oc->nohw", xp[:, :, dy:dy + H, dx:dx + Wd], W[:, :, dy, dx], optimize=True)
        x = np.maximum(y + b[None, :, None, None], 0)
        if i in POOL_AFTER:
            n, c, H, Wd = x.shape
            x = x.reshape(n, c, H // 2, 2, Wd // 2, 2).max(axis=(3, 5))
            pools.append(x)
    f = _features(pools)
    f = f / np.sqrt((f ** 2).sum(axis=1))[:, None]
    return f @ np.asarray(values[26], dtype=np.float64) + np.asarray(values[27], dtype=np.float64)


def pair_similarity(emb_pairs, values):
    e = np.asarray(emb_pairs, dtype=np.float64).reshape(-1, 2, emb_pairs.shape[-1])
    d = np.sqrt((np.abs(e[:, 0] - e[:, 1]) ** 2).sum(axis=1, keepdims=True))
    w, b = float(np.asarray(values[28]).reshape(())), float(np.asarray(values[29]).reshape(()))
    return 1.0 / (1.0 + np.exp(-(w * d + b)))
