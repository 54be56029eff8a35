"""GPU (-m gpu): similarityNet + patch cropping + early rejection through the C ABI.
Patch cropping is bit-exact vs the reference-run goldens; the network is compared with oracle/simil_oracle.py (fp64)."""
import os

import numpy as np
import pytest

import golden_util
from oracle import simil_oracle

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "simil_cases.npz"))
MEAN_BGR = np.asarray([103.939, 116.779, 123.68]).astype(np.float32)      # params.py:130
# embeddings: |values| ~ 0.3. f16x3 operands are fp32-class: observed ~1e-6 absolute; f16 fast mode ~1e-3.
TOL_EMB_X3, TOL_EMB_F16 = 2e-5, 2e-2


@pytest.fixture(scope="module")
def sn(gpu_required):
    import surfacenet_amd
    return surfacenet_amd


def scene_images():
    H, W = (int(v) for v in G["sc_hw"])
    return [golden_util.synth_image(int(sd), H, W) for sd in G["sc_seeds"]]


def test_crop_patches_bit_exact_vs_reference_golden(sn):
    imgs = scene_images()
    with sn.Context(cube_D=8, max_samples=2) as ctx:
        ctx.set_images(imgs)
        out = ctx.crop_patches(int(G["crop_view"]), G["crop_ch"], G["crop_cw"])
        out2 = ctx.crop_patches(int(G["crop_view"]), G["crop_rh"].mean(axis=1), G["crop_rw"].mean(axis=1))
        with pytest.raises(sn.SurfaceNetHipError):
            ctx.crop_patches(3, G["crop_ch"], G["crop_cw"])
    assert out.dtype == np.uint8 and np.array_equal(out, G["crop_out"])
    assert np.array_equal(out2, G["crop_out_ranges"])


@pytest.mark.parametrize("precision,n", [("f16x3", 5), ("f16", 3), ("f16m8", 2), ("f16x3", 11)])
def test_patch2embedding_vs_oracle(sn, precision, n):
    from surfacenet_amd import weights
    values = weights.synthetic_simil_param_values(1)
    rs = np.random.RandomState(n)
    raw = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    raw[0] = 0                                                    # the all-black patch of earlyRejection.py:31
    # smooth content in some patches (pure noise is the harshest case for cancellation, images are not noise)
    raw[1] = (np.indices((64, 64)).sum(0)[:, :, None] * [1, 2, 3] % 256).astype(np.uint8)
    X = simil_oracle.preprocess(raw, MEAN_BGR)
    want = simil_oracle.embedding_torch(X, values)
    with sn.Context(cube_D=8, max_samples=2, precision=precision) as ctx:
        with pytest.raises(sn.SurfaceNetHipError):
            ctx.patch2embedding(X)                                # weights not loaded
        ctx.load_simil_param_values(values)
        got = ctx.patch2embedding(X)
        again = ctx.patch2embedding(X[::-1].copy())[::-1]         # batch position must not matter
    assert got.dtype == np.float32 and got.shape == (n, 128)
    tol = TOL_EMB_F16 if precision == "f16" else TOL_EMB_X3
    assert np.abs(got - want).max() < tol, np.abs(got - want).max()
    assert np.abs(want).max() > 0.05
    assert np.array_equal(got, again)


def test_patch2embedding_chunking_and_precision_switch(sn):
    """More patches than one workspace chunk (2048) and a precision switch in between: identical rows per patch."""
    from surfacenet_amd import weights
    values = weights.synthetic_simil_param_values(2)
    base = simil_oracle.preprocess(np.random.RandomState(5).randint(0, 256, (6, 64, 64, 3)).astype(np.uint8), MEAN_BGR)
    X = np.ascontiguousarray(np.tile(base, (342, 1, 1, 1))[:2051])
    with sn.Context(cube_D=8, max_samples=2) as ctx:
        ctx.load_simil_param_values(values)
        a = ctx.patch2embedding(X)
        lib = ctx._lib
        assert lib.sn_set_precision(ctx._h, 0) == 0               # f16
        b = ctx.patch2embedding(base)
        assert lib.sn_set_precision(ctx._h, 1) == 0               # back to f16x3: weights are re-packed
        c = ctx.patch2embedding(base)
    assert np.array_equal(a[:6], c) and np.array_equal(a[6:12], c) and np.array_equal(a[2046:2051], c[:5])
    assert np.abs(b - c).max() < TOL_EMB_F16 and not np.array_equal(b, c)


def test_pair_similarity_vs_oracle(sn):
    from surfacenet_amd import weights
    values = weights.synthetic_simil_param_values(1)
    e = (np.random.RandomState(2).randn(14, 128) * 0.3).astype(np.float32)
    e[3] = e[2]                                                   # identical pair -> sigmoid(b)
    with sn.Context(cube_D=8, max_samples=2) as ctx:
        ctx.load_simil_param_values(values)
        s = ctx.embeddingpair2simil(e)
        with pytest.raises(TypeError):
            ctx.embeddingpair2simil(e[:3])
    assert s.shape == (7, 1) and s.dtype == np.float32
    assert np.abs(s - simil_oracle.pair_similarity(e, values)).max() < 1e-6


def test_early_rejection_dropin_fused_equals_three_step_protocol(sn):
    """earlyRejection.patch2embedding with the GPU-backed callable (crop+preprocess+net fused in HBM) == the reference's
    literal protocol (cropImgPatches -> preprocess_patches -> patch2embedding_fn per batch) == oracle."""
    from surfacenet_amd import earlyRejection, runtime, similarityNet, weights
    from surfacenet_amd.viewPairSelection import k_combination_np
    runtime.reset()
    values = weights.synthetic_simil_param_values(4)
    p2e, pair_fn = similarityNet.similarityNet_inference(None, (64, 64), param_values=values)
    imgs = scene_images()
    N_views, N_cubes = G["sc_img_h"].shape[:2]
    ctr = np.stack([G["sc_ctr_h"], G["sc_ctr_w"]], axis=0)
    emb_f, ins_f = earlyRejection.patch2embedding(imgs, G["sc_img_h"], G["sc_img_w"], p2e, MEAN_BGR, N_cubes, N_views, 128, patchSize=64,
                                                  batchSize=3, cubeCenter_hw=ctr)
    plain = lambda x: p2e(x)                                       # no .sn_gpu attribute -> literal three-step path
    emb_p, ins_p = earlyRejection.patch2embedding(imgs, G["sc_img_h"], G["sc_img_w"], plain, MEAN_BGR, N_cubes, N_views, 128, patchSize=64,
                                                  batchSize=3, cubeCenter_hw=ctr)
    assert np.array_equal(ins_f, G["er_inscope"]) and np.array_equal(ins_p, G["er_inscope"])
    assert np.array_equal(emb_f, emb_p)
    # oracle for every (cube, view): in scope -> its crop, else the all-black patch
    for v in range(N_views):
        raw = simil_oracle.crop_patches(imgs[v], ctr[0, v], ctr[1, v])
        raw[~G["er_inscope"][:, v]] = 0
        want = simil_oracle.embedding_torch(simil_oracle.preprocess(raw, MEAN_BGR), values)
        assert np.abs(emb_f[:, v] - want).max() < TOL_EMB_X3
    dis = earlyRejection.embeddingPairs2simil(embeddings=emb_f, embeddingPair2simil_fn=pair_fn, inScope_cubes_vs_views=ins_f,
                                              viewPairs=k_combination_np(range(N_views), k=2), N_views=N_views, batchSize=4)
    pairs = k_combination_np(range(N_views), k=2)
    want = simil_oracle.pair_similarity(emb_f[:, pairs.flatten()].reshape(-1, 128), values).reshape(N_cubes, -1)
    assert dis.shape == (N_cubes, 3) and np.abs(dis - want).max() < 1e-6
    # the one-call all-pairs path (taken above because pair_fn is GPU-backed) == the reference's batched pair protocol
    dis_b = earlyRejection.embeddingPairs2simil(embeddings=emb_f, embeddingPair2simil_fn=lambda e: pair_fn(e), inScope_cubes_vs_views=ins_f,
                                                viewPairs=pairs, N_views=N_views, batchSize=4)
    assert np.array_equal(dis, dis_b)
    with pytest.raises(TypeError):
        p2e(np.zeros((1, 3, 64, 64)))                              # float64, as Theano would reject
    with pytest.raises(NotImplementedError):
        similarityNet.similarityNet_inference(None, (32, 32), param_values=values)
    runtime.reset()


def test_crop_embed_many_chunks_equals_crop_then_embed(sn):
    """sn_crop_embed over several internal chunks (centres up / embeddings down once per call, the call's buffer grown on demand) against
    the two-step path: crop the same patches, preprocess on the host, embed them - row for row, and again after a smaller call."""
    from surfacenet_amd import weights
    values = weights.synthetic_simil_param_values(3)
    imgs = scene_images()
    H, W = imgs[0].shape[:2]
    rs = np.random.RandomState(11)
    n = 2 * 2040 + 17
    ch = rs.uniform(-20, H + 20, n)                     # some centres outside the image: the crop clamps
    cw = rs.uniform(-20, W + 20, n)
    with sn.Context(cube_D=8, max_samples=2) as ctx:
        ctx.set_images(imgs)
        ctx.load_simil_param_values(values)
        small = ctx.crop_embed(0, ch[:5], cw[:5], MEAN_BGR)                  # allocates the per-call buffer small ...
        got = ctx.crop_embed(0, ch, cw, MEAN_BGR)                            # ... then has to grow it
        raw = ctx.crop_patches(0, ch, cw)
        want = ctx.patch2embedding(simil_oracle.preprocess(raw, MEAN_BGR))
        again = ctx.crop_embed(0, ch[100:2200], cw[100:2200], MEAN_BGR)
    assert got.shape == (n, 128) and np.array_equal(got, want)
    assert np.array_equal(small, got[:5]) and np.array_equal(again, got[100:2200])
