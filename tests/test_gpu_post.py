"""GPU (-m gpu): ray pooling + dense2sparse through the C ABI, bit-exact against the reference-run
goldens (tests/golden/post_cases.npz) and the oracle (oracle/post_oracle.py) on seeded batches."""
import os

import numpy as np
import pytest

from oracle import post_oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
POST = np.load(os.path.join(GOLD, "post_cases.npz"))
P_DTU = np.load(os.path.join(GOLD, "cameras.npz"))["P_dtu"]
PARAM_DT = [("xyz", np.float32, (3,)), ("ijk", np.uint32, (3,)), ("resol", np.float32)]    # utils/scene.py:55


@pytest.fixture(scope="module")
def sn(gpu_required):
    import surfacenet_amd
    return surfacenet_amd


def field(seed, D, n, zero_frac=0.0, quant=None):
    rs = np.random.RandomState(seed)
    g = np.indices((D, D, D)).astype(np.float32) / D
    out = []
    for i in range(n):
        dist = g[i % 3] - (0.5 + 0.2 * np.sin(5 * g[(i + 1) % 3] + i) * np.cos(4 * g[(i + 2) % 3]))
        p = np.exp(-(dist * D / 2.0) ** 2) * 0.9 + 0.08 * rs.rand(D, D, D)
        if quant:
            p = np.round(p * quant) / quant
        if zero_frac:
            p[rs.rand(D, D, D) < zero_frac] = 0
        out.append(p)
    return np.clip(np.stack(out), 0, 0.9999).astype(np.float32)


@pytest.mark.parametrize("name", [str(n) for n in POST["rp_names"]])
def test_ray_pool_bit_exact_vs_reference_golden(sn, name):
    g = lambda k: POST[name + "/" + k]
    thr = float(g("thresh"))
    D = g("pred32").shape[0]
    with sn.Context(cube_D=D, max_samples=4) as ctx:
        ctx.set_cameras(g("P"))
        votes = ctx.ray_pool(g("pairs")[None], g("xyz")[None], g("resol").reshape(1), g("pred32")[None], None if np.isnan(thr) else thr)
    assert votes.dtype == np.uint8 and votes.shape == (1, D, D, D)
    assert np.array_equal(votes[0], g("votes"))


@pytest.mark.parametrize("D,thr,zero_frac,quant", [(32, 0.5, 0.0, None), (32, None, 0.5, 8), (16, 0.3, 0.0, 16), (64, 0.6, 0.0, None)])
def test_ray_pool_batch_vs_oracle(sn, D, thr, zero_frac, quant):
    n, n_vp = (3, 2) if D == 64 else (7, 3)
    rs = np.random.RandomState(D)
    pred = field(D + 1, D, n, zero_frac, quant)
    pairs = rs.randint(0, 4, (n, n_vp, 2))
    pairs[0] = [[1, 1]] * n_vp                                  # one view repeated 2*n_vp times
    xyz = (rs.rand(n, 3) * [60, 60, 40] + [-30, -30, 590]).astype(np.float32)
    resol = rs.choice([0.2, 0.4, 0.8], n).astype(np.float32)
    with sn.Context(cube_D=D, max_samples=4) as ctx:
        ctx.set_cameras(P_DTU)
        votes = ctx.ray_pool(pairs, xyz, resol, pred, thr)
        again = ctx.ray_pool(pairs, xyz, resol, pred, thr)
    assert np.array_equal(votes, again)                         # atomics-based, still deterministic
    p16 = pred.astype(np.float16)
    for i in range(n):
        want = post_oracle.ray_pool_1cube(P_DTU, p16[i], pairs[i], xyz[i], resol[i], thr)
        assert np.array_equal(votes[i], want.astype(np.uint8)), i
    assert votes.max() <= 2 * n_vp
    if thr is not None:
        assert not votes[p16 <= np.float16(thr)].any()          # only selected voxels can be voted
    assert (votes[0][votes[0] > 0] == 2 * n_vp).all()           # the repeated view votes with its multiplicity


def d2s_inputs(name):
    g = lambda k: POST[name + "/" + k]
    D, Dc, crop, rp_on, rp_thr = (int(v) for v in g("cfg"))
    param = np.empty((g("xyz").shape[0],), dtype=PARAM_DT)
    param["xyz"], param["resol"], param["ijk"] = g("xyz"), g("resol"), np.arange(12).reshape(4, 3)
    return g, dict(D=D, Dc=Dc or None, crop=bool(crop), rp_on=bool(rp_on), rp_thr=rp_thr, min_prob=float(g("min_prob"))), param


@pytest.mark.parametrize("name", [str(n) for n in POST["d2s_names"]])
def test_dense2sparse_dropin_vs_reference_golden(sn, name):
    from surfacenet_amd import sparseCubes, runtime
    g, c, param = d2s_inputs(name)
    p16 = g("pred32").astype(np.float16)[:, 0]
    rgb8 = np.transpose(g("rgbf").astype(np.uint8), (0, 2, 3, 4, 1))
    ne, ijk_l, p_l, rgb_l, v_l, param_new = sparseCubes.dense2sparse(
        prediction=p16, rgb=rgb8, param=param, viewPair=g("pairs").astype(np.uint16), min_prob=c["min_prob"], rayPool_thresh=c["rp_thr"],
        enable_centerCrop=c["crop"], cube_Dcenter=c["Dc"], enable_rayPooling=c["rp_on"], cameraPOs=P_DTU, cameraTs=None)
    assert np.array_equal(ne, g("nonempty")) and np.array_equal([len(x) for x in p_l], g("counts"))
    assert ijk_l[0].dtype == np.uint8 and p_l[0].dtype == np.float16 and rgb_l[0].dtype == np.uint8
    assert np.array_equal(np.concatenate(ijk_l), g("ijk"))
    assert np.array_equal(np.concatenate(p_l).view(np.uint16), g("pred16").view(np.uint16))
    assert np.array_equal(np.concatenate(rgb_l), g("rgb"))
    if c["rp_on"]:
        assert v_l[0].dtype == np.uint8 and np.array_equal(np.concatenate(v_l), g("votes"))
    else:
        assert v_l == []
    assert np.array_equal(param_new["xyz"], g("xyz_new")) and np.array_equal(param["xyz"], g("xyz"))   # input untouched
    runtime.reset()


def test_append_dense_2sparseList_matches_oracle_s32(sn):
    """The call of main_reconstruct.py:153-160 at the reference's settings (rayPool_thresh=0, centre crop 32 -> 26)."""
    from surfacenet_amd import sparseCubes, runtime
    n, n_vp, D, Dc = 6, 2, 32, 26
    rs = np.random.RandomState(5)
    pred = field(9, D, n)[:, None]
    pred[4] = 0.2                                                # an empty cube in the middle
    rgbf = (rs.rand(n, 3, D, D, D) * 255.999).astype(np.float32)
    param = np.empty((n,), dtype=PARAM_DT)
    param["xyz"] = (rs.rand(n, 3) * [60, 60, 40] + [-30, -30, 590]).astype(np.float32)
    param["resol"] = 0.4
    param["ijk"] = rs.randint(0, 50, (n, 3))
    pairs = rs.randint(0, 4, (n, n_vp, 2))
    lists = sparseCubes.append_dense_2sparseList(
        prediction_sub=pred, rgb_sub=rgbf, param_sub=param, viewPair_sub=pairs, min_prob=0.5, rayPool_thresh=0, enable_centerCrop=True,
        cube_Dcenter=Dc, enable_rayPooling=True, cameraPOs=P_DTU, cameraTs=None, prediction_list=[], rgb_list=[], vxl_ijk_list=[],
        rayPooling_votes_list=[], cube_ijk_np=None, param_np=None, viewPair_np=None)
    p_l, rgb_l, ijk_l, v_l, cube_ijk, param_np, vp_np = lists
    p16, rgb8 = post_oracle.to_sparse_inputs(pred, rgbf)
    ne, o_ijk, o_p, o_rgb, o_v, o_xyz = post_oracle.dense2sparse(p16, rgb8, param["xyz"], param["resol"], pairs, min_prob=0.5, rayPool_thresh=0,
                                                                 enable_centerCrop=True, cube_Dcenter=Dc, enable_rayPooling=True, cameraPOs=P_DTU)
    assert ne == [0, 1, 2, 3, 5] and len(p_l) == len(ne)
    for a, b in ((ijk_l, o_ijk), (rgb_l, o_rgb), (v_l, o_v)):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert all(np.array_equal(x.view(np.uint16), y.view(np.uint16)) for x, y in zip(p_l, o_p))
    assert np.array_equal(param_np["xyz"], o_xyz[ne]) and np.array_equal(cube_ijk, param["ijk"][ne]) and np.array_equal(vp_np, pairs[ne])
    # round trip: scattering the packed lists back gives exactly the thresholded, cropped dense cube
    lo = (D - Dc) // 2
    for k, i in enumerate(ne):
        dense = np.zeros((Dc, Dc, Dc), dtype=np.float16)
        dense[tuple(ijk_l[k].T)] = p_l[k]
        crop = p16[i][lo:lo + Dc, lo:lo + Dc, lo:lo + Dc]
        assert np.array_equal(dense, np.where(crop > np.float16(0.5), crop, np.float16(0)))
    masks = sparseCubes.filter_voxels(vxl_mask_list=[], prediction_list=p_l, prob_thresh=0.7, rayPooling_votes_list=v_l, rayPool_thresh=2)
    assert all(np.array_equal(m, (p >= 0.7) & (v >= 2)) for m, p, v in zip(masks, p_l, v_l))
    runtime.reset()


def test_post_error_paths(sn):
    from surfacenet_amd import rayPooling, runtime
    pred = field(3, 8, 1)
    with sn.Context(cube_D=8, max_samples=2) as ctx:
        with pytest.raises(sn.SurfaceNetHipError):                # cameras not set
            ctx.ray_pool(np.zeros((1, 1, 2), int), np.zeros((1, 3)), np.ones(1), pred, 0.5)
        ctx.set_cameras(P_DTU)
        with pytest.raises(sn.SurfaceNetHipError):                # view id out of range
            ctx.ray_pool(np.full((1, 1, 2), 4), np.zeros((1, 3)), np.ones(1), pred, 0.5)
        # a cube touching the camera's principal plane (Z = 0 for this pinhole at the origin): pixels at infinity -> loud error
        ctx.set_cameras(np.asarray([[[1000.0, 0, 0, 0], [0, 1000.0, 0, 0], [0, 0, 1.0, 0]]]))
        with pytest.raises(sn.SurfaceNetHipError):
            ctx.ray_pool(np.zeros((1, 1, 2), int), np.asarray([[1.0, 1.0, 0.0]]), np.ones(1), pred, None)
        ctx.set_cameras(P_DTU)
        ok = ctx.ray_pool(np.zeros((1, 1, 2), int), np.asarray([[0, 0, 600.0]]), np.full(1, 0.4), pred, 0.5)    # context still usable
        assert ok.shape == (1, 8, 8, 8)
    with pytest.raises(TypeError):                               # float32 values that float16 cannot hold
        rayPooling.rayPooling_1cube_numpy(P_DTU, None, pred[0] + 1e-5, np.zeros((1, 2), int), [0, 0, 600.0], 0.4, 0.5)
    with pytest.raises(ValueError):
        rayPooling.rayPooling_1cube_numpy(P_DTU, None, pred[0, 0], np.zeros((1, 2), int), [0, 0, 600.0], 0.4, 0.5)
    v = rayPooling.rayPooling_1cube_numpy(P_DTU, None, pred[0].astype(np.float16), np.asarray([[0, 1]]), np.asarray([0, 0, 600.0]), np.float32(0.4), 0.5)
    want = post_oracle.ray_pool_1cube(P_DTU, pred[0].astype(np.float16), np.asarray([[0, 1]]), [0, 0, 600.0], 0.4, 0.5)
    assert np.array_equal(v, want)
    runtime.reset()


@pytest.mark.parametrize("n_vp,crop", [(2, True), (1, False)])
def test_sparse_loop_device_chain_equals_piecewise(sn, n_vp, crop):
    """reconstruct.SparseLoop (CVC -> CNN -> fusion -> colours -> ray pooling -> dense2sparse without leaving HBM)
    == the same steps through the host-array entry points, checked against the oracle's dense2sparse."""
    import golden_util
    import synth
    from surfacenet_amd import reconstruct
    s, n = 16, 5
    sc = golden_util.synthetic_scene(n, n_vp, s=s, seed=11, hw=(600, 800))
    values = list(synth.calibrated_params(1))
    # the synthetic net's probabilities hover around its calibration mean: threshold at their median so ~half survive
    with sn.Context(cube_D=s, max_samples=16) as ctx:
        ctx.load_param_values(values)
        ctx.set_cameras(sc["cams"]); ctx.set_images(sc["imgs"])
        fused, unfused, cvc = ctx.cvc_forward(sc["pairs"], sc["xyz"], sc["resol"], sc["w"], return_cvc=True)
        rgb = ctx.color_fuse(cvc, unfused, sc["w"])
        thr = float(np.median(fused.astype(np.float16)))
        loop = reconstruct.SparseLoop(ctx, n_vp, min_prob=thr, rayPool_thresh=0, enable_centerCrop=crop, cube_Dcenter=12 if crop else None,
                                      enable_rayPooling=True)
        got = loop.run(sc["pairs"], sc["xyz"], sc["resol"], sc["w"])
        got2 = loop.run(sc["pairs"][:2], sc["xyz"][:2], sc["resol"][:2], sc["w"][:2])      # buffers are reusable, shorter batch
        loop.close()
        small = reconstruct.SparseLoop(ctx, n_vp, max_cubes=2, min_prob=thr, rayPool_thresh=0, enable_centerCrop=crop,
                                       cube_Dcenter=12 if crop else None, enable_rayPooling=True)
        many = small.run_many(sc["pairs"], sc["xyz"], sc["resol"], sc["w"])            # 5 cubes through 2-cube batches, pipelined
        small.close()
    p16, rgb8 = post_oracle.to_sparse_inputs(fused, rgb)
    want = post_oracle.dense2sparse(p16, rgb8, sc["xyz"], sc["resol"], sc["pairs"], min_prob=thr, rayPool_thresh=0, enable_centerCrop=crop,
                                    cube_Dcenter=12 if crop else None, enable_rayPooling=True, cameraPOs=sc["cams"])
    assert got[0] == want[0] and len(want[0]) > 0
    for k in (1, 3, 4):
        assert all(np.array_equal(a, b) for a, b in zip(got[k], want[k])), k
    assert all(np.array_equal(a.view(np.uint16), b.view(np.uint16)) for a, b in zip(got[2], want[2]))
    assert np.array_equal(got[5], want[5])
    assert got2[0] == [i for i in want[0] if i < 2]
    assert many[0] == want[0] and np.array_equal(many[5], want[5])
    for k in (1, 3, 4):
        assert all(np.array_equal(a, b) for a, b in zip(many[k], want[k])), k
    assert all(np.array_equal(a.view(np.uint16), b.view(np.uint16)) for a, b in zip(many[2], want[2]))
    assert all(np.array_equal(a, b) for a, b in zip(got2[1], want[1]))


def test_empty_batches_everywhere(sn):
    """n = 0 through every host-array entry point: empty results of the right shape and dtype, no error (the reference's
    numpy code returns empty arrays in the same situations)."""
    import synth
    from surfacenet_amd import weights
    s = 8
    z = lambda *sh, dt=np.float32: np.zeros(sh, dtype=dt)
    with sn.Context(cube_D=s, max_samples=4) as ctx:
        ctx.load_param_values(list(synth.calibrated_params(0)))
        ctx.load_simil_param_values(weights.synthetic_simil_param_values(0))
        ctx.set_cameras(P_DTU)
        ctx.set_images([np.zeros((40, 50, 3), np.uint8)] * 4)
        pairs0, xyz0, r0 = z(0, 2, 2, dt=np.int64), z(0, 3), z(0)
        assert ctx.cvc(pairs0, xyz0, r0).shape == (0, 6, s, s, s)
        f, u = ctx.forward(z(0, 6, s, s, s), z(0, 2), n_vp=2)
        assert f.shape == (0, 1, s, s, s) and u.shape == (0, 2, s, s, s)
        f, u, c = ctx.cvc_forward(pairs0, xyz0, r0, z(0, 2), return_cvc=True)
        assert f.shape == (0, 1, s, s, s) and c.shape == (0, 6, s, s, s)
        assert ctx.relative_weights(z(0, 258), 3).shape == (0, 3)
        assert ctx.color_fuse(z(0, 6, s, s, s), z(0, 2, s, s, s), z(0, 2)).shape == (0, 3, s, s, s)
        assert ctx.ray_pool(pairs0, xyz0, r0, z(0, s, s, s), 0.5).shape == (0, s, s, s)
        off, ijk, p16, rgb, votes = ctx.dense2sparse(z(0, s, s, s), z(0, 3, s, s, s, dt=np.uint8), pairs0, xyz0, r0, enable_rayPooling=True)
        assert off.tolist() == [0] and ijk.shape == (0, 3) and p16.dtype == np.float16 and votes.shape == (0,)
        assert ctx.crop_patches(0, z(0, dt=np.float64), z(0, dt=np.float64)).shape == (0, 64, 64, 3)
        assert ctx.patch2embedding(z(0, 3, 64, 64)).shape == (0, 128)
        assert ctx.crop_embed(1, z(0, dt=np.float64), z(0, dt=np.float64), [1, 2, 3]).shape == (0, 128)
        assert ctx.embeddingpair2simil(z(0, 128)).shape == (0, 1)
        assert ctx.embeddings2simil(z(0, 4, 128)).shape == (0, 6)
        assert ctx.viewpair_weights(z(0, 4, 128), z(0, 6), z(0, 6)).shape == (0, 6)
        # and the context still works afterwards
        assert ctx.ray_pool(np.zeros((1, 1, 2), int), np.asarray([[0, 0, 600.0]]), np.full(1, 0.4), field(3, 8, 1), 0.5).shape == (1, s, s, s)
