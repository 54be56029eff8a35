"""Pins the early-rejection host logic (surfacenet_amd/{earlyRejection,image}.py) and oracle/simil_oracle.py's
patch cropping against outputs of the reference's own functions (tests/golden/simil_cases.npz; generator:
oracle/gen_golden_simil.py), and cross-checks the two formulations of the similarityNet oracle."""
import os

import numpy as np
import pytest

import golden_util
from oracle import simil_oracle
from oracle import cvc_oracle
from surfacenet_amd import earlyRejection, image, weights
from surfacenet_amd.viewPairSelection import camera_centers, k_combination_np

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "simil_cases.npz"))


def scene_images():
    H, W = (int(v) for v in G["sc_hw"])
    return [golden_util.synth_image(int(sd), H, W) for sd in G["sc_seeds"]]


def test_preprocess_patches_matches_reference():
    doc = image.preprocess_patches(np.zeros((2, 2, 5, 3)), mean_BGR=np.array([1, 2, 3]))        # utils/image.py:24-33 doctest
    assert np.array_equal(doc, G["pre_doc"]) and doc.shape == (2, 3, 2, 5)
    out = image.preprocess_patches(G["pre_in"].astype(np.float32), mean_BGR=G["pre_mean"])
    assert out.dtype == np.float32 and np.array_equal(out, G["pre_out"])
    assert np.array_equal(simil_oracle.preprocess(G["pre_in"], G["pre_mean"]), G["pre_out"])


def _oracle_corners(P, xyz_min, cube_D_mm, return_int_hw):
    """perspectiveProj_cubesCorner (utils/camera.py:188-245) on the oracle's projection: (N_Ms, N_cubes, 8)."""
    xyz_min = np.atleast_2d(np.asarray(xyz_min))
    corners = xyz_min[:, None, :] + np.indices((2, 2, 2)).reshape((3, -1)).T[None] * cube_D_mm
    h, w = cvc_oracle.perspectiveProj(P, corners.reshape((-1, 3)), return_int_hw=return_int_hw)
    return h.reshape((-1, xyz_min.shape[0], 8)), w.reshape((-1, xyz_min.shape[0], 8))


def test_cube_corner_projection_and_scope_check_match_reference():
    """The oracle's projection against the reference-run corner / centre vectors (the product's GPU projection is checked
    against the same vectors in tests/test_gpu_dropin.py); the in-scope test is host logic of the product."""
    h, w = _oracle_corners(G["cc_doc_Ms"], G["cc_doc_pts"], 1, False)                           # camera.py:211-216 doctest
    assert np.array_equal(h, G["cc_doc_h"]) and np.array_equal(w, G["cc_doc_w"])
    assert np.allclose(w[:, :, 0], [[1.35860185, 0.9878389], [0.64522543, 0.76079278]])
    D = np.float32(G["sc_D"])
    h, w = _oracle_corners(G["sc_P"], G["sc_xyz"], D, False)
    assert np.array_equal(h, G["sc_img_h"]) and np.array_equal(w, G["sc_img_w"])
    ch, cw = cvc_oracle.perspectiveProj(G["sc_P"], G["sc_xyz"] + D / 2., return_int_hw=False)
    assert np.array_equal(ch, G["sc_ctr_h"]) and np.array_equal(cw, G["sc_ctr_w"])
    hw = tuple(int(v) for v in G["sc_hw"])
    ins = np.stack([image.img_hw_cubesCorner_inScopeCheck(hw, h[v], w[v]) for v in range(h.shape[0])])
    assert np.array_equal(ins, G["sc_inscope"])


def test_oracle_crop_matches_reference():
    img = scene_images()[int(G["crop_view"])]
    assert np.array_equal(simil_oracle.crop_patches(img, G["crop_ch"], G["crop_cw"]), G["crop_out"])
    assert np.array_equal(simil_oracle.crop_patches(img, G["crop_rh"].mean(axis=1), G["crop_rw"].mean(axis=1)), G["crop_out_ranges"])


def test_early_rejection_host_logic_matches_reference(monkeypatch):
    """patch2embedding / embeddingPairs2simil / selectFromSimilarity with the exactly-rounded stand-in callables; the GPU
    crop of image.cropImgPatches is replaced by the (reference-pinned) oracle crop so this runs without a GPU."""
    imgs = scene_images()
    monkeypatch.setattr(image, "cropImgPatches", lambda img, range_h, range_w, patchSize, pyramidRate, interp_order, cubeCenter_hw:
                        simil_oracle.crop_patches(img, cubeCenter_hw[0], cubeCenter_hw[1], patchSize))
    N_views, N_cubes = G["sc_img_h"].shape[:2]
    emb, inscope = earlyRejection.patch2embedding(imgs, G["sc_img_h"], G["sc_img_w"], golden_util.toy_embedding, G["pre_mean"], N_cubes, N_views, 128,
                                                  patchSize=64, batchSize=3, cubeCenter_hw=np.stack([G["sc_ctr_h"], G["sc_ctr_w"]], axis=0))
    assert emb.dtype == np.float32 and np.array_equal(emb, G["er_emb"]) and np.array_equal(inscope, G["er_inscope"])
    dis = earlyRejection.embeddingPairs2simil(embeddings=emb, embeddingPair2simil_fn=golden_util.toy_pair_simil, inScope_cubes_vs_views=inscope,
                                              viewPairs=k_combination_np(range(N_views), k=2), N_views=N_views, batchSize=4)
    assert np.array_equal(dis, G["er_dis"])
    for n in (1, 2, 3):
        sel = earlyRejection.selectFromSimilarity(dis, n)
        assert sel.dtype == bool and np.array_equal(sel, G["er_sel%d" % n])


def test_simil_oracle_two_formulations_and_known_answers():
    values = weights.synthetic_simil_param_values(3)
    assert [tuple(v.shape) for v in values] == weights.SIMIL_PARAM_SHAPES and weights.D_SIMIL_FEATURE == 5888
    X = simil_oracle.preprocess(np.random.RandomState(0).randint(0, 256, (2, 64, 64, 3)).astype(np.uint8), G["pre_mean"])
    a, pools = simil_oracle.embedding_torch(X, values, return_pools=True)
    b = simil_oracle.embedding_numpy(X, values)
    assert a.shape == (2, 128) and np.abs(a - b).max() < 1e-9
    assert [p.shape[1:] for p in pools] == [(64, 32, 32), (128, 16, 16), (256, 8, 8), (512, 4, 4), (512, 2, 2)]
    # centre-crop order (nets/layers.py:55-63 doctest: 6x6 map, r=2 -> rows/cols 1..4): here r=1 on a 4x4 map -> [5,6,9,10]
    f = simil_oracle._features([np.zeros((1, 1, 2, 2)) for _ in range(4)] + [np.zeros((1, 1, 2, 2))])
    assert f.shape == (1, 20)
    p = np.arange(16.0).reshape(1, 1, 4, 4)
    f = simil_oracle._features([p, p, p, p, np.arange(4.0).reshape(1, 1, 2, 2)])
    assert np.array_equal(f[0], [0, 1, 2, 3] + [5, 6, 9, 10] * 4)
    # pair similarity known answers: identical embeddings -> sigmoid(b); distance 5 (3-4-5) -> sigmoid(5w + b)
    e = np.zeros((4, 128), np.float32); e[2, 0] = 3; e[3, 1] = -4
    s = simil_oracle.pair_similarity(e, values)
    w, bb = float(values[28].reshape(())), float(values[29].reshape(()))
    assert np.allclose(s[:, 0], [1 / (1 + np.exp(-bb)), 1 / (1 + np.exp(-(5 * w + bb)))])


def test_camera_centres_doctest():
    P = np.array([[798.693916, -2438.153488, 1568.674338, -542599.034996], [-44.838945, 1433.912029, 2576.399630, -1176685.647358],
                  [-0.840873, -0.344537, 0.417405, 382.793511]])                      # utils/camera.py:91-95 doctest
    t = np.array([555.64348632032, 191.10837560939, 360.02470478273])
    assert np.allclose(camera_centers(P[None])[0], t)
    assert camera_centers([P, P]).shape == (2, 3)
    assert np.allclose(P @ np.r_[camera_centers(P)[0], 1.0], 0, atol=1e-6)
    # the drop-in's cameraPs2Ts keeps the reference's container contract (utils/camera.py:103-120): list in -> list out, array in -> array out
    from surfacenet_amd import camera
    as_list = camera.cameraPs2Ts([P, 2 * P])
    assert type(as_list) is list and len(as_list) == 2 and as_list[0].shape == (3,) and np.allclose(as_list[0], t) and np.allclose(as_list[1], t)
    as_arr = camera.cameraPs2Ts(np.stack([P, P]))
    assert isinstance(as_arr, np.ndarray) and as_arr.shape == (2, 3) and np.allclose(as_arr, t)
