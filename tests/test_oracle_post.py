"""Pins oracle/post_oracle.py (ray pooling, dense2sparse) and the host-side view-pair selection against outputs of the
reference's own functions (tests/golden/post_cases.npz, vps_cases.npz; generator: oracle/gen_golden_post.py)."""
import os

import numpy as np
import pytest

from oracle import net_oracle, post_oracle
from surfacenet_amd import viewPairSelection as vps
from surfacenet_amd import weights

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def post():
    return np.load(os.path.join(GOLD, "post_cases.npz"))


def rp_case(post, name):
    g = lambda k: post[name + "/" + k]
    thr = float(g("thresh"))
    return g("P"), g("pred32"), g("pairs"), g("xyz"), g("resol"), (None if np.isnan(thr) else thr), g("votes")


def test_ray_pool_matches_reference_goldens(post):
    for name in post["rp_names"]:
        P, pred32, pairs, xyz, resol, thr, votes = rp_case(post, str(name))
        got = post_oracle.ray_pool_1cube(P, pred32.astype(np.float16), pairs, xyz, resol, thr)
        assert got.shape == votes.shape
        assert np.array_equal(got.astype(np.uint8), votes), name


def d2s_case(post, name):
    g = lambda k: post[name + "/" + k]
    D, Dc, crop, rp_on, rp_thr = (int(v) for v in g("cfg"))
    return dict(pred32=g("pred32"), rgbf=g("rgbf"), xyz=g("xyz"), resol=g("resol"), pairs=g("pairs"), D=D, Dc=Dc or None,
                crop=bool(crop), rp_on=bool(rp_on), rp_thr=rp_thr, min_prob=float(g("min_prob")), nonempty=g("nonempty"),
                counts=g("counts"), ijk=g("ijk"), pred16=g("pred16"), rgb=g("rgb"), votes=g("votes"), xyz_new=g("xyz_new"))


def test_dense2sparse_matches_reference_goldens(post):
    P = np.load(os.path.join(GOLD, "cameras.npz"))["P_dtu"]
    for name in post["d2s_names"]:
        c = d2s_case(post, str(name))
        p16, rgb8 = post_oracle.to_sparse_inputs(c["pred32"], c["rgbf"])
        ne, ijk_l, p_l, rgb_l, v_l, xyz_new = post_oracle.dense2sparse(
            p16, rgb8, c["xyz"], c["resol"], c["pairs"], min_prob=c["min_prob"], rayPool_thresh=c["rp_thr"],
            enable_centerCrop=c["crop"], cube_Dcenter=c["Dc"], enable_rayPooling=c["rp_on"], cameraPOs=P)
        assert np.array_equal(ne, c["nonempty"])
        assert np.array_equal([len(x) for x in p_l], c["counts"])
        assert np.array_equal(np.concatenate(ijk_l), c["ijk"])
        assert np.array_equal(np.concatenate(p_l).view(np.uint16), c["pred16"].view(np.uint16))
        assert np.array_equal(np.concatenate(rgb_l), c["rgb"])
        if c["rp_on"]:
            assert np.array_equal(np.concatenate(v_l), c["votes"])
        assert np.array_equal(xyz_new, c["xyz_new"])


def test_view_pair_selection_matches_reference_goldens():
    v = np.load(os.path.join(GOLD, "vps_cases.npz"))
    for N_arg in (1, 2):          # utils/viewPairSelection.py:19-33 doctest
        a, b = vps.__argmaxN_viewPairs__(v["doc_pairs"], v["doc_w"], N_arg)
        assert np.array_equal(a, v["argmax%d/pairs" % N_arg]) and np.array_equal(b, v["argmax%d/w" % N_arg])
    assert np.array_equal(vps.k_combination_np(range(3), k=2), v["doc_pairs"])
    ang = vps.viewPairAngles_wrt_pts(v["ang_Ts"], v["ang_pts"])            # utils/camera.py:290-296 doctest
    assert ang.dtype == v["ang_out"].dtype and np.array_equal(ang, v["ang_out"])
    assert np.allclose(ang * 180 / np.pi, [[45, 45, 60], [45, 45, 90]])
    values = weights.synthetic_param_values(int(v["sel_seed"]))
    relw = lambda f, n_samples_perGroup: net_oracle.relative_weights(f, values, n_samples_perGroup)
    pairs, w = vps.viewPairSelection(v["sel_Ts"], v["sel_e"], v["sel_d"], v["sel_valid"], v["sel_centers"], relw, int(v["sel_batch"]),
                                     int(v["sel_N"]), v["sel_viewPairs"])
    assert np.array_equal(pairs, v["sel_pairs"]) and np.array_equal(w, v["sel_w"])
    assert pairs.shape == (int(v["sel_valid"].sum()), int(v["sel_N"]), 2)
    assert np.all(np.diff(w, axis=1) >= 0)                                 # ascending: the largest weight is last


def test_yield_batch_selectors():
    sel = list(vps.yield_batch_npBool(7, 3))      # utils/utils.py:131-139 doctest of gen_batch_npBool
    assert np.array_equal(np.array(sel), [[1, 1, 1, 0, 0, 0, 0], [0, 0, 0, 1, 1, 1, 0], [0, 0, 0, 0, 0, 0, 1]])
    assert len(list(vps.yield_batch_npBool(6, 100))) == 1
