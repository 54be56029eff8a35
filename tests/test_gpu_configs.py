"""GPU (-m gpu): BASELINE.json configs[2], [3] and [4] on their OWN single-GPU workloads (what one MI355X of the posed job runs).

  config 3 (index 2)  DTU scan9 bounding box, s=32, all 49 views / 1,176 view pairs, N_viewPairs4inference = 5 (params.py:165-172)
  config 4 (index 3)  s=64 cubes, batch 256 over 8 GPUs -> this GPU's shard: 32 cubes x 2 view pairs, device-resident
  config 5 (index 4)  Middlebury dino, s=32, 16 views / 120 pairs, 16 view pairs, early rejection + view-pair selection active
                      (params.py:176-182)

Calibration and cube grids are the datasets' own (surfacenet_amd/data/calibration.npz, synthetic.cube_grid: both pinned against the
reference's readers / initializeCubes in tests/test_host_logic.py); views are seeded noise and the networks BN-calibrated random nets
(no dataset pixels or trained weights travel). For each config: the scene runs through `reconstruct.reconstruct_scene` (every stage of
main_reconstruct.py:67-173 on the GPU), then sampled valid cubes are re-run dense with the pairs / weights the scene selected and
checked against the oracle — CVC bit-exact, surface probabilities within TOL — and the scene's own sparse lists must be the fp16
values of that dense result at the kept voxels."""
import numpy as np
import pytest

import golden_util

pytestmark = pytest.mark.gpu
TOL_X3 = 2e-4                         # the default mode's asserted tolerance (tests/test_gpu_parity.py); north-star bar 1e-3
MEAN_BGR = np.asarray([103.939, 116.779, 123.68]).astype(np.float32)


def _scene_through_pipeline(config, n_cubes, min_prob=0.5):
    import synth
    from surfacenet_amd import SurfaceNet, reconstruct, runtime, similarityNet, synthetic, weights
    runtime.reset()
    P, imgs, cubes, cube_D_mm, Dc, n_vp = synthetic.dataset_scene(config, 32, n_cubes)
    net_values = list(synth.calibrated_params(1))
    simil_values = weights.synthetic_simil_param_values(0)
    simil_values[28][:] = 3.0; simil_values[29][:] = -2.5          # logistic unit: synthetic pair distances fall inside / outside the accepted band
    runtime.DEFAULT_MAX_SAMPLES = 128 if n_vp <= 8 else 8 * n_vp
    p2e, pair_fn = similarityNet.similarityNet_inference(None, (64, 64), param_values=simil_values)
    relw_fn, _ = SurfaceNet.SurfaceNet_inference(n_vp, None, None, cube_D=32, param_values=net_values)
    res = reconstruct.reconstruct_scene(imgs, P, cubes, cube_D_mm, 32, n_vp, p2e, pair_fn, relw_fn, cube_Dcenter=Dc, patches_mean_bgr=MEAN_BGR,
                                        min_prob=min_prob)
    return dict(P=P, imgs=imgs, cubes=cubes, Dc=Dc, n_vp=n_vp, net_values=net_values, res=res, min_prob=min_prob)


def _check_sampled_cubes(sn, sc, picks):
    """Dense re-run of the sampled valid cubes with the scene's own selections vs the oracle, and the scene's sparse lists vs that dense result."""
    from oracle import cvc_oracle, net_oracle
    from surfacenet_amd import runtime
    res, n_vp, s, Dc = sc["res"], sc["n_vp"], 32, sc["Dc"]
    valid = np.nonzero(res["validCubes"])[0]
    vp, w = res["viewPairs4Reconstr"], res["w_viewPairs4Reconstr"]
    assert vp.shape == (len(valid), n_vp, 2) and w.shape == (len(valid), n_vp)
    assert (vp[:, :, 0] < vp[:, :, 1]).all() and vp.max() < len(sc["imgs"])
    xyz, resol = sc["cubes"]["xyz"][valid][picks], sc["cubes"]["resol"][valid][picks]
    with sn.Context(cube_D=s, max_samples=len(picks) * n_vp) as ctx:
        ctx.load_param_values(sc["net_values"]); ctx.set_cameras(sc["P"]); ctx.set_images(sc["imgs"])
        fused, unfused, cvc = ctx.cvc_forward(vp[picks], xyz, resol, w[picks], return_cvc=True)
    ref_cvc = cvc_oracle.gen_coloredCubes(vp[picks], xyz, resol, sc["P"], sc["imgs"], s, mean6=golden_util.MEAN6)
    assert np.array_equal(cvc, ref_cvc), "CVC warp differs from the oracle on the dataset's own calibration"
    f32, u32 = net_oracle.forward_torch(ref_cvc, sc["net_values"], w=w[picks], n_vp=n_vp, dtype="float32")
    e_u, e_f = float(np.abs(unfused - u32).max()), float(np.abs(fused - f32).max())
    print("   sampled cubes %s: L_inf vs fp32 oracle unfused %.3e fused %.3e (probability range %.3f .. %.3f)" % (list(picks), e_u, e_f, u32.min(), u32.max()))
    assert e_u < TOL_X3 and e_f < TOL_X3
    # the scene's sparse lists == fp16 of the dense probabilities at the voxels dense2sparse keeps (centre crop, fp16 probability > min_prob; main_reconstruct.py:154-160 passes rayPool_thresh = 0)
    lo = (s - Dc) // 2
    valid_pos = {int(c): i for i, c in enumerate(np.nonzero(res["validCubes"])[0])}
    ijk_of = {tuple(int(v) for v in row): j for j, row in enumerate(res["cube_ijk_np"])}
    checked = 0
    for k, pick in enumerate(picks):
        j = ijk_of.get(tuple(int(v) for v in sc["cubes"]["ijk"][valid][pick]))
        if j is None:
            continue                                     # cube kept no voxel
        vox, p16 = res["vxl_ijk_list"][j].astype(np.int64), res["prediction_list"][j]
        dense16 = fused[k, 0].astype(np.float16)[lo:lo + Dc, lo:lo + Dc, lo:lo + Dc]
        assert np.array_equal(dense16[vox[:, 0], vox[:, 1], vox[:, 2]], p16)
        assert len(p16) == int((dense16 > np.float16(sc["min_prob"])).sum())          # rayPool_thresh = 0: exactly the voxels above min_prob
        checked += 1
    return checked


def test_config3_dtu_scan9_all_view_pairs(gpu_required):
    """BASELINE configs[2]: an even 240-cube sample of the scan9 grid (195,360 cubes), all 49 views (1,176 view pairs), N_vp = 5."""
    import surfacenet_amd as sn
    sc = _scene_through_pipeline("dtu_scan9", 240)
    res = sc["res"]
    n_valid = int(res["validCubes"].sum())
    assert res["inScope_cubes_vs_views"].shape == (240, 49) and res["dissimilarity"].shape == (240, 1176)
    print("   scan9 sample: %d in-scope patches, %d / 240 cubes valid, %d non-empty" % (int(res["inScope_cubes_vs_views"].sum()), n_valid, len(res["prediction_list"])))
    assert 5 <= n_valid < 240, "early rejection must reject some cubes and keep some"
    picks = np.unique(np.linspace(0, n_valid - 1, 3).astype(np.int64))
    assert _check_sampled_cubes(sn, sc, picks) >= 1
    from surfacenet_amd import runtime
    runtime.reset()


def test_config5_middlebury_dino_16_view_pairs(gpu_required):
    """BASELINE configs[4] on one GPU: a 240-cube sample of the dino grid, P_mid16 calibration, 480x640 views, 120 candidate pairs,
    early rejection + view-pair selection active, 16 view pairs per cube through the CNN."""
    import surfacenet_amd as sn
    sc = _scene_through_pipeline("dino", 240)
    res = sc["res"]
    n_valid = int(res["validCubes"].sum())
    assert sc["n_vp"] == 16 and res["dissimilarity"].shape == (240, 120) and sc["imgs"][0].shape == (480, 640, 3)
    print("   dino sample: %d in-scope patches, %d / 240 cubes valid, %d non-empty" % (int(res["inScope_cubes_vs_views"].sum()), n_valid, len(res["prediction_list"])))
    assert n_valid >= 16
    # selection really selected: 16 distinct pairs per cube out of 120, weights ascending (viewPairSelection.py:38 keeps argsort order)
    vp, w = res["viewPairs4Reconstr"], res["w_viewPairs4Reconstr"]
    codes = vp[:, :, 0] * 16 + vp[:, :, 1]
    assert all(len(set(r)) == 16 for r in codes) and (np.diff(w, axis=1) >= 0).all() and (w > 0).all()
    picks = np.unique(np.linspace(0, n_valid - 1, 2).astype(np.int64))
    assert _check_sampled_cubes(sn, sc, picks) >= 1
    from surfacenet_amd import runtime
    runtime.reset()


def test_config4_s64_shard_32_cubes(gpu_required):
    """BASELINE configs[3]: s=64, batch 256 over 8 GPUs = 32 cubes x 2 view pairs per GPU and step, device-resident
    (sn_cvc_forward_dev). Size-independent properties over the whole shard + two sampled cubes against the fp32 oracle."""
    import surfacenet_amd as sn
    import synth
    from oracle import cvc_oracle, net_oracle
    s, n, n_vp = 64, 32, 2
    sc = golden_util.synthetic_scene(n, n_vp, s=s, seed=21)
    values = list(synth.calibrated_params(2))
    v = s ** 3
    with sn.Context(cube_D=s, max_samples=n * n_vp) as ctx:
        ctx.load_param_values(values); ctx.set_cameras(sc["cams"]); ctx.set_images(sc["imgs"])
        d = [ctx.upload(sc[k]) for k in ("pairs", "xyz", "resol", "w")]
        d_f, d_u, d_c = ctx.dev_alloc(n * v * 4), ctx.dev_alloc(n * n_vp * v * 4), ctx.dev_alloc(2 * n_vp * 6 * v * 4)
        ctx.cvc_forward_dev(n, n_vp, d[0], d[1], d[2], d[3], d_f, d_u)
        fused, unfused = np.empty((n, 1, s, s, s), np.float32), np.empty((n, n_vp, s, s, s), np.float32)
        ctx.d2h(fused, d_f); ctx.d2h(unfused, d_u)
        # (1) probabilities; fused = normalised weighted mean of the unfused ones (nets/layers.py:325-336)
        assert np.isfinite(unfused).all() and unfused.min() > 0.0 and unfused.max() < 1.0 and unfused.std() > 0.05
        cw = sc["w"] / sc["w"].sum(axis=1, keepdims=True)
        assert np.abs((unfused * cw[:, :, None, None, None]).sum(axis=1, keepdims=True) - fused).max() < 1e-6
        # (2) cubes are independent (what the 8-GPU sharding relies on): a 2-cube sub-batch reproduces the shard's bits, and so does a second pass
        idx = np.asarray([3, 29])
        d2 = [ctx.upload(np.ascontiguousarray(sc[k][idx])) for k in ("pairs", "xyz", "resol", "w")]
        ctx.cvc_forward_dev(2, n_vp, d2[0], d2[1], d2[2], d2[3], d_f, d_u, d_c, mean=golden_util.MEAN6)
        f2, u2, cvc = np.empty((2, 1, s, s, s), np.float32), np.empty((2, n_vp, s, s, s), np.float32), np.empty((2 * n_vp, 6, s, s, s), np.float32)
        ctx.d2h(f2, d_f); ctx.d2h(u2, d_u); ctx.d2h(cvc, d_c)
        assert np.array_equal(f2, fused[idx]) and np.array_equal(u2, unfused[idx])
        for p in d + d2 + [d_f, d_u, d_c]:
            ctx.dev_free(p)
    # (3) the two sampled cubes: CVC bit-exact, probabilities vs the fp32 oracle
    ref_cvc = cvc_oracle.gen_coloredCubes(sc["pairs"][idx], sc["xyz"][idx], sc["resol"][idx], sc["cams"], sc["imgs"], s, mean6=golden_util.MEAN6)
    assert np.array_equal(cvc, ref_cvc)
    f32, u32 = net_oracle.forward_torch(ref_cvc, values, w=sc["w"][idx], n_vp=n_vp, dtype="float32")
    e_u, e_f = float(np.abs(u2 - u32).max()), float(np.abs(f2 - f32).max())
    print("   s=64 shard, cubes %s: L_inf vs fp32 oracle unfused %.3e fused %.3e" % (list(idx), e_u, e_f))
    assert e_u < TOL_X3 and e_f < TOL_X3
