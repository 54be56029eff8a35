"""CPU: pins the oracle's CVC / projection / batching restatements against golden vectors produced by the
reference's own code (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

import golden_util
from oracle import cvc_oracle

CASES = golden_util.cvc_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_c_oracle_matches_reference_golden(name):
    c = CASES[name]
    out = cvc_oracle.gen_coloredCubes(c["pairs"], c["xyz"], c["resol"], c["P"], golden_util.case_images(c), int(c["s"]))
    assert out.dtype == np.float32 and out.shape == c["out_u8"].shape
    assert np.array_equal(out, c["out_u8"].astype(np.float32))          # bit-exact (integer-valued)


@pytest.mark.parametrize("name", ["dtu_s8_vp1", "dtu_s16_vp3_edge", "mid_s16_vp2"])
def test_numpy_restatement_matches_reference_golden(name):
    c = CASES[name]
    out = cvc_oracle.gen_coloredCubes_numpy(c["pairs"], c["xyz"], c["resol"], c["P"], golden_util.case_images(c), int(c["s"]))
    assert np.array_equal(out, c["out_u8"].astype(np.float32))


@pytest.mark.parametrize("name", ["dtu_real", "mid_real"])
def test_oracle_matches_reference_on_real_dataset_pixels(name):
    """Windows of two DTU scan9 JPEGs / two Middlebury dino PNGs (decoded in the build container, stored as arrays) with the principal
    point of their P matrices shifted to the window origin; expected outputs from the reference's own CVC.py (oracle/gen_golden_real.py).
    One cube inside both windows, one crossing the window border."""
    c = golden_util.real_cases()[name]
    imgs = golden_util.case_images(c)
    assert imgs[0].shape == (128, 160, 3) and imgs[0].dtype == np.uint8 and imgs[0].std() > 10       # real pixels, not a constant
    for fn in (cvc_oracle.gen_coloredCubes, cvc_oracle.gen_coloredCubes_numpy):
        out = fn(c["pairs"], c["xyz"], c["resol"], c["P"], imgs, int(c["s"]))
        assert np.array_equal(out, c["out_u8"].astype(np.float32))
    pre = cvc_oracle.gen_coloredCubes(c["pairs"][:1], c["xyz"][:1], c["resol"][:1], c["P"], imgs, int(c["s"]), mean6=golden_util.MEAN6)
    assert np.array_equal(pre, c["pre_f32_cube0"])
    frac = (c["out_u8"][1].reshape(2, 3, -1).max(axis=1) > 0).mean()
    assert 0.2 < frac < 0.9                                   # the second cube really leaves the window


def test_preprocess_golden():
    c = CASES["dtu_s8_vp1"]
    out = cvc_oracle.gen_coloredCubes(c["pairs"], c["xyz"], c["resol"], c["P"], golden_util.case_images(c), 8, mean6=golden_util.MEAN6)
    assert np.array_equal(out, c["pre_f32"])                              # X.astype(f32) - mean, CVC.py:110-111


def test_out_of_scope_is_zero_and_edge_cases_present():
    c = CASES["dtu_s16_vp3_edge"]
    o = c["out_u8"]
    assert (o[3:] == 0).all()                 # cube 1 projects outside every image
    frac = (o[:3].reshape(3, 2, 3, -1).max(axis=2) > 0).mean()
    assert 0.05 < frac < 0.95                 # cube 0 straddles the image border


def test_view_index_out_of_range_raises():
    c = CASES["dtu_s8_vp1"]
    bad = c["pairs"].copy(); bad[0, 0, 0] = 4
    with pytest.raises(IndexError):
        cvc_oracle.gen_coloredCubes(bad, c["xyz"], c["resol"], c["P"], golden_util.case_images(c), 8)


def test_projection_goldens():
    z = np.load(os.path.join(golden_util.GOLDEN, "proj_cases.npz"))
    # doctest of camera.py:144-160
    assert np.allclose(z["doc_w_f"], np.array([[1.35860185, 0.9878389], [0.64522543, 0.76079278]]))
    h, w = cvc_oracle.perspectiveProj(z["doc_Ms"], z["doc_pts"], return_int_hw=False)
    assert np.allclose(h, z["doc_h_f"], rtol=1e-13, atol=0) and np.allclose(w, z["doc_w_f"], rtol=1e-13, atol=0)
    hi, wi = cvc_oracle.perspectiveProj(z["doc_Ms"], z["doc_pts"], return_int_hw=True)
    assert np.array_equal(hi, z["doc_h_i"]) and np.array_equal(wi, z["doc_w_i"])
    assert np.array_equal(wi, np.array([[1, 1], [1, 1]]))
    h, w = cvc_oracle.perspectiveProj(z["dtu_P"], z["dtu_pts"], return_int_hw=False)
    assert np.allclose(h, z["dtu_h_f"], rtol=1e-12, atol=0) and np.allclose(w, z["dtu_w_f"], rtol=1e-12, atol=0)
    hi, wi = cvc_oracle.perspectiveProj(z["dtu_P"], z["dtu_pts"], return_int_hw=True)
    assert np.array_equal(hi, z["dtu_h_i"]) and np.array_equal(wi, z["dtu_w_i"])


def test_projection_chain_matches_numpy_dgemm():
    """The FMA chain of cvc_oracle.c is what np.dot(3x4, 4xN) computes (checked where numpy's BLAS does so)."""
    import ctypes
    P = golden_util.cameras()["P_dtu"][0]
    rs = np.random.RandomState(3)
    pts = np.vstack([rs.rand(3, 4096) * 100 + np.array([[-50], [-50], [560]]), np.ones((1, 4096))])
    ref = np.dot(P, pts)
    out = np.empty((3, 4096))
    cvc_oracle.lib().sn_oracle_dot34(4096, P.ctypes.data_as(ctypes.c_void_p), np.ascontiguousarray(pts[:3]).ctypes.data_as(ctypes.c_void_p),
                                      out.ctypes.data_as(ctypes.c_void_p))
    mism = int((out != ref).sum())
    if mism:
        assert np.allclose(out, ref, rtol=1e-14)   # a BLAS without FMA kernels differs in the last ulp only
        pytest.skip("this numpy's BLAS does not use the FMA chain (%d last-ulp differences)" % mism)


def test_batch_selector_goldens():
    z = np.load(os.path.join(golden_util.GOLDEN, "batch_cases.npz"))
    for i in range(5):
        sel = cvc_oracle.gen_non0Batch_npBool(z["c%d/ind" % i], int(z["c%d/bs" % i]))
        assert np.array_equal(sel, z["c%d/sel" % i])


def test_color_fusion_golden():
    z = np.load(os.path.join(golden_util.GOLDEN, "color_cases.npz"))
    X = z["col_u8"].astype(np.float32) - golden_util.MEAN6[None, :, None, None, None]
    X += golden_util.MEAN6[None, :, None, None, None]                 # what the caller holds at main_reconstruct.py:150
    rgb = cvc_oracle.color_fuse(X, z["pred"], z["w"])
    assert rgb.dtype == np.uint8 and np.array_equal(rgb, z["rgb"])


@pytest.mark.parametrize("name", ["cfg1_s32", "cfg3_s64", "edge_s64"])
def test_c_oracle_at_the_configs_own_sizes(name):
    """the C oracle was pinned at s <= 32 on hand-placed cubes only. Here: BASELINE configs[1]'s own synthetic scene (full 1200x1600
    frames, s = 32), configs[3]'s (s = 64) and an s = 64 cube pair with out-of-scope voxels, against digests (sha256 over every voxel + 4,096
    sampled values + per-channel sums) of what the reference's own CVC.py returned for them (oracle/gen_golden_configs.py)."""
    sc, s, c = golden_util.cvc_config_cases()[name]
    out = cvc_oracle.gen_coloredCubes(sc["pairs"], sc["xyz"], sc["resol"], sc["cams"], sc["imgs"], s)
    golden_util.check_cvc_digest(out, c)
    if name == "edge_s64":
        assert 0.2 < float(c["inscope"]) < 0.8                   # the case really has out-of-scope voxels
