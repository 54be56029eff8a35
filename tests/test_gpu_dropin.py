"""GPU (-m gpu): the drop-in modules with the reference's own call signatures, written the way the
reference's loop body (main_reconstruct.py:132-152) calls them."""
import numpy as np
import pytest

import golden_util

pytestmark = pytest.mark.gpu
CASES = golden_util.cvc_cases()


@pytest.fixture()
def dropin(gpu_required):
    from surfacenet_amd import CVC, SurfaceNet, runtime
    runtime.reset()
    yield CVC, SurfaceNet, runtime
    runtime.reset()


def test_gen_coloredCubes_keyword_call_matches_reference_golden(dropin):
    CVC, _, _ = dropin
    for name in ("dtu_s16_vp2", "mid_s16_vp2"):                # two scenes in a row: the cached scene must be re-bound
        c = CASES[name]
        imgs = golden_util.case_images(c)
        out = CVC.gen_coloredCubes(selected_viewPairs=c["pairs"], xyz=c["xyz"], resol=c["resol"], colorize_cube_D=int(c["s"]),
                                   cameraPOs=c["P"], models_img=imgs, visualization_ON=False)
        assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and out.flags.writeable
        assert np.array_equal(out, c["out_u8"].astype(np.float32))
    with pytest.raises(IndexError):                            # numpy raises IndexError in the reference
        bad = c["pairs"].copy(); bad[0, 0, 0] = 99
        CVC.gen_coloredCubes(bad, c["xyz"], c["resol"], c["P"], imgs, int(c["s"]))
    empty = CVC.gen_coloredCubes(c["pairs"][:0], c["xyz"][:0], c["resol"][:0], c["P"], imgs, int(c["s"]))
    assert empty.shape == (0, 6, 16, 16, 16)


def test_loop_body_three_call_protocol(dropin):
    """main_reconstruct.py:134-150 verbatim (names kept), N_viewPairs4inference = 2 and 1."""
    CVC, SurfaceNet, runtime = dropin
    import synth
    from oracle import net_oracle
    c = CASES["dtu_s16_vp2"]
    cube_D = int(c["s"])
    images_list, cameraPOs_np = golden_util.case_images(c), c["P"]
    values = list(synth.calibrated_params(0))
    MEAN = golden_util.MEAN6
    for N_viewPairs4inference in (2, 1):
        viewPair_relativeImpt_fn, nViewPair_SurfaceNet_fn = SurfaceNet.SurfaceNet_inference(
            N_viewPairs4inference, model_file=None, layerNameList_2_load=["output_SurfaceNet_reshape", "output_softmaxWeights"],
            cube_D=(cube_D if N_viewPairs4inference == 2 else None), param_values=values)      # None: cube size inferred from X.shape
        pairs = c["pairs"][:, :N_viewPairs4inference]
        w = (np.random.RandomState(3).rand(pairs.shape[0], N_viewPairs4inference) + 0.1).astype(np.float32)
        _CVCs1_sub = CVC.gen_coloredCubes(selected_viewPairs=pairs, xyz=c["xyz"], resol=c["resol"], colorize_cube_D=cube_D,
                                          cameraPOs=cameraPOs_np, models_img=images_list, visualization_ON=False)
        _, _CVCs2_sub = CVC.preprocess_augmentation(None, _CVCs1_sub, mean_rgb=MEAN[None, :, None, None, None], augment_ON=False, crop_ON=False)
        surfacePrediction, unfused_predictions = nViewPair_SurfaceNet_fn(_CVCs2_sub) if N_viewPairs4inference == 1 \
            else nViewPair_SurfaceNet_fn(_CVCs2_sub, w)
        keep = _CVCs2_sub.copy()
        _CVCs2_sub += MEAN[None, :, None, None, None]                   # :150 mutates the array in place
        f64, u64 = net_oracle.forward_torch(keep, values, w=w, n_vp=N_viewPairs4inference)
        assert surfacePrediction.shape == (pairs.shape[0], 1, cube_D, cube_D, cube_D)
        assert np.abs(surfacePrediction - f64).max() < 2e-4 and np.abs(unfused_predictions - u64).max() < 2e-4
        if N_viewPairs4inference == 1:
            assert unfused_predictions is surfacePrediction or np.array_equal(unfused_predictions, surfacePrediction)
            with pytest.raises(TypeError):
                nViewPair_SurfaceNet_fn(_CVCs2_sub, w)
        else:
            with pytest.raises(TypeError):
                nViewPair_SurfaceNet_fn(_CVCs2_sub.astype(np.float64), w)      # Theano rejects non-float32 input
            sm = viewPair_relativeImpt_fn(np.random.RandomState(1).rand(6, 258).astype(np.float32), n_samples_perGroup=3)
            assert sm.shape == (2, 3) and np.allclose(sm.sum(axis=1), 1, atol=1e-5)


def test_hot_loop_matches_three_call_protocol(dropin):
    """reconstruct.hot_loop (one fused call per batch) == the three-call protocol, batch partition of utils.gen_non0Batch_npBool."""
    _, _, _ = dropin
    import surfacenet_amd
    import synth
    from surfacenet_amd import reconstruct
    s, n_all, n_vp = 16, 9, 2
    validCubes = np.array([1, 1, 0, 1, 1, 1, 0, 1, 1], dtype=bool)
    sc = golden_util.synthetic_scene(n_all, n_vp, s=s, seed=2, hw=(600, 800))
    cubes_param_np = np.zeros(n_all, dtype=[("xyz", np.float32, (3,)), ("ijk", np.uint32, (3,)), ("resol", np.float32)])
    cubes_param_np["xyz"], cubes_param_np["resol"] = sc["xyz"], sc["resol"]
    viewPairs4Reconstr, w4 = sc["pairs"][validCubes], sc["w"][validCubes]          # indexed by valid cube, as in the reference
    values = list(synth.calibrated_params(1))
    with surfacenet_amd.Context(cube_D=s, max_samples=8) as ctx:
        ctx.load_param_values(values); ctx.set_cameras(sc["cams"]); ctx.set_images(sc["imgs"])
        seen = np.zeros(n_all, dtype=int)
        for _batch, pred, unf, cvc_raw in reconstruct.hot_loop(ctx, validCubes, viewPairs4Reconstr, w4, cubes_param_np, batch_size=3):
            sel = _batch[validCubes]
            seen += _batch
            raw = ctx.cvc(viewPairs4Reconstr[sel], cubes_param_np["xyz"][_batch], cubes_param_np["resol"][_batch])
            assert np.abs(cvc_raw - raw).max() < 1e-4                                 # (raw - mean) + mean, fp32
            f, u = ctx.forward(raw - golden_util.MEAN6[None, :, None, None, None], w4[sel], n_vp=n_vp)
            assert np.array_equal(f, pred) and np.array_equal(u, unf)
        assert np.array_equal(seen, validCubes.astype(int))                           # every valid cube exactly once


def test_camera_projection_dropin_bit_exact_vs_reference_golden(dropin):
    """camera.perspectiveProj / perspectiveProj_cubesCorner (utils/camera.py:123-245) on the GPU against vectors produced by the
    reference's own functions (tests/golden/simil_cases.npz, proj_cases.npz), incl. the doctest inputs of camera.py:144-160,211-227."""
    import os
    from surfacenet_amd import camera
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "simil_cases.npz"))
    h, w = camera.perspectiveProj_cubesCorner(G["cc_doc_Ms"], G["cc_doc_pts"], cube_D_mm=1, return_int_hw=False)
    assert h.dtype == np.float64 and np.array_equal(h, G["cc_doc_h"]) and np.array_equal(w, G["cc_doc_w"])
    assert np.allclose(w[:, :, 0], [[1.35860185, 0.9878389], [0.64522543, 0.76079278]])
    hi, wi = camera.perspectiveProj_cubesCorner(G["cc_doc_Ms"][1], G["cc_doc_pts"][0], cube_D_mm=1, return_int_hw=True)
    assert hi.shape == (1, 1, 8) and hi.dtype == np.int64 and np.array_equal(hi, np.round(G["cc_doc_h"][1:2, 0:1]).astype(np.int64))
    D = np.float32(G["sc_D"])
    h, w = camera.perspectiveProj_cubesCorner(G["sc_P"], G["sc_xyz"], cube_D_mm=D, return_int_hw=False)
    assert np.array_equal(h, G["sc_img_h"]) and np.array_equal(w, G["sc_img_w"])
    ch, cw = camera.perspectiveProj(G["sc_P"], G["sc_xyz"] + D / 2., return_int_hw=False)
    assert np.array_equal(ch, G["sc_ctr_h"]) and np.array_equal(cw, G["sc_ctr_w"])
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "proj_cases.npz"))
    for Ms, pts, pre in ((z["doc_Ms"], z["doc_pts"], "doc"), (z["dtu_P"], z["dtu_pts"], "dtu")):
        names = [k for k in z.files if k.startswith(pre)]
        hf, wf = camera.perspectiveProj(Ms, pts, return_int_hw=False)
        hq, wq, dq = camera.perspectiveProj(Ms, pts, return_int_hw=True, return_depth=True)
        assert hq.dtype == np.int64 and dq.shape == hf.shape
        assert np.array_equal(hq, np.round(hf).astype(np.int64)) and np.array_equal(wq, np.round(wf).astype(np.int64))
        for k in names:                                                     # whichever float / int vectors the fixture holds
            if k.endswith("_h_f"): assert np.array_equal(hf, z[k])
            if k.endswith("_w_f"): assert np.array_equal(wf, z[k])
            if k.endswith("_h_i"): assert np.array_equal(hq, z[k])
            if k.endswith("_w_i"): assert np.array_equal(wq, z[k])
    one_h, one_w = camera.perspectiveProj(z["doc_Ms"][0], z["doc_pts"][0], return_int_hw=False)     # (3,4) x (3,) -> (1,)
    assert one_h.shape == (1,) and one_h[0] == z["doc_h_f"][0, 0] and one_w[0] == z["doc_w_f"][0, 0]
    with pytest.raises(ValueError):
        camera.perspectiveProj(np.zeros((4, 4)), np.zeros((2, 3)))
    with pytest.raises(ValueError):
        camera.perspectiveProj(np.zeros((3, 4)), np.zeros((2, 2)))


def test_scene_cache_sees_in_place_changes_and_recycled_lists(dropin):
    """runtime.bind_scene keys the uploaded images on content, not only on id(): an image changed in place, or a new list that
    happens to reuse the ids of a freed one, must be uploaded again."""
    CVC, _, runtime = dropin
    c = CASES["dtu_s16_vp2"]
    imgs = [im.copy() for im in golden_util.case_images(c)]
    kw = dict(selected_viewPairs=c["pairs"], xyz=c["xyz"], resol=c["resol"], colorize_cube_D=int(c["s"]), cameraPOs=c["P"])
    a = CVC.gen_coloredCubes(models_img=imgs, **kw)
    assert np.array_equal(a, c["out_u8"].astype(np.float32))
    for im in imgs:
        im[...] = 255 - im                                              # same objects, same ids, new pixels
    b = CVC.gen_coloredCubes(models_img=imgs, **kw)
    inv = CVC.gen_coloredCubes(models_img=[255 - im for im in golden_util.case_images(c)], **kw)
    assert np.array_equal(b, inv) and not np.array_equal(a, b)


@pytest.mark.parametrize("proto", [2, 0])
def test_inference_entry_points_read_python2_model_files(dropin, tmp_path, proto):
    """`SurfaceNet_inference(N_viewPairs4inference, model_file, layerNameList_2_load)` (nets/SurfaceNet.py:385-402) and
    `similarityNet_inference(model_file, imgPatch_hw_size)` (nets/similarityNet.py:229-244) on `.model` FILES in the formats Python 2
    writes them (cPickle protocol 2, and the ASCII protocol 0 that the reference's text-mode open implies; tests/py2pickle.py emits
    both opcode by opcode): file -> pickle -> sn_load_weights / sn_simil_load_weights -> forward on the GPU, against the oracle run on the
    arrays that went into the file."""
    CVC, SurfaceNet, runtime = dropin
    import py2pickle
    import synth
    from oracle import net_oracle, simil_oracle
    from surfacenet_amd import similarityNet, weights
    dumps = py2pickle.dumps_py2 if proto == 2 else py2pickle.dumps_py2_proto0
    values = list(synth.calibrated_params(2))
    model_file = tmp_path / "2D_2_3D-19-0.918_0.951.model"                    # params.py:106
    model_file.write_bytes(dumps(values))
    s, n, n_vp = 16, 2, 2
    relw_fn, net_fn = SurfaceNet.SurfaceNet_inference(n_vp, str(model_file), ["output_SurfaceNet_reshape", "output_softmaxWeights"])
    X = synth.random_cvc(n * n_vp, s, 77)
    w = (np.random.RandomState(5).rand(n, n_vp) + 0.1).astype(np.float32)
    fused, unfused = net_fn(X, w)
    f64, u64 = net_oracle.forward_torch(X, values, w=w, n_vp=n_vp)
    e = max(float(np.abs(fused - f64).max()), float(np.abs(unfused - u64).max()))
    print("protocol %d: SurfaceNet from file, L_inf vs fp64 oracle %.3e" % (proto, e))
    assert e < 2e-4
    feats = np.random.RandomState(6).randn(3 * 4, 258).astype(np.float32)
    got = relw_fn(feats, n_samples_perGroup=4)                                 # the relative-weight MLP rides in the same file (arrays 98..104)
    ref = net_oracle.relative_weights(feats, values, 4) if hasattr(net_oracle, "relative_weights") else None
    assert got.shape == (3, 4) and np.allclose(got.sum(axis=1), 1, atol=1e-5)
    if ref is not None:
        assert np.abs(got - ref).max() < 1e-5
    svals = weights.synthetic_simil_param_values(4)
    simil_file = tmp_path / "epoch33_acc_tr0.707_val0.791.model"               # params.py:91
    simil_file.write_bytes(dumps(svals))
    p2e, pair_fn = similarityNet.similarityNet_inference(str(simil_file), (64, 64))
    patches = (np.random.RandomState(8).randint(0, 256, (5, 3, 64, 64)).astype(np.float32) - np.asarray([103.939, 116.779, 123.68], np.float32)[None, :, None, None])
    emb = p2e(patches)
    ref_emb = simil_oracle.embedding_torch(patches, svals, dtype="float64")
    print("protocol %d: similarityNet from file, embedding L_inf vs fp64 oracle %.3e" % (proto, float(np.abs(emb - ref_emb).max())))
    assert emb.shape == (5, 128) and np.abs(emb - ref_emb).max() < 1e-4


def _wide_net(spread):
    """A BN-calibrated net whose merge_conv_a BatchNorm under-estimates the spread of its pre-activations `spread`-fold (function unchanged: the fp64
    oracle of the plain net is the reference) - the stress case of tests/test_gpu_numerics.py, here fed through the DROP-IN callables."""
    import synth
    from surfacenet_amd import weights
    ix = {(layer, p): i for i, (layer, p, _) in enumerate(weights.PARAM_LAYOUT)}
    values = [np.array(v) for v in synth.calibrated_params(2)]
    values[ix[("merge_conv_a", "inv_std")]] *= np.float32(spread)
    return values


@pytest.mark.parametrize("spread,tol", [(3.2, 2e-4), (10.0, 1e-3)])
def test_dropin_callables_calibrate_saturating_nets_and_warn_once(dropin, spread, tol):
    """the numerics safety net must reach the caller who holds real weights (nets/SurfaceNet.py:385-402, main_reconstruct.py:145-146).
    `nViewPair_SurfaceNet_fn` reads the saturation warning after the batch, recalibrates the 6-bit premultipliers on it, redoes it and emits exactly
    ONE RuntimeWarning (layer name + exponents); later calls stay silent. x3.2: within the default tolerance 2e-4; x10: within the 1e-3 bar
    (uncalibrated: 1.35e-3). auto_calibrate=False keeps the static exponents (no warning, the graceful degradation of DESIGN 5.1)."""
    import warnings
    import synth
    from oracle import net_oracle
    _, SurfaceNet, runtime = dropin
    s, n, n_vp = 16, 2, 2
    values = _wide_net(spread)
    X = synth.random_cvc(n * n_vp, s, 31)
    w = (np.random.RandomState(2).rand(n, n_vp) + 0.1).astype(np.float32)
    f64, u64 = net_oracle.forward_torch(X, values, w=w, n_vp=n_vp)
    _, fn = SurfaceNet.SurfaceNet_inference(n_vp, None, param_values=values)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        fused, unfused = fn(X, w)
        fused2, unfused2 = fn(X, w)                    # the second call: calibrated already, silent
    msgs = [str(r.message) for r in rec if issubclass(r.category, RuntimeWarning)]
    assert len(msgs) == 1 and "merge_conv_a" in msgs[0] and "s_act 0 ->" in msgs[0], msgs
    e = float(np.abs(unfused - u64).max())
    print("   spread x%.1f through nViewPair_SurfaceNet_fn: L_inf %.3e (tolerance %.0e); warning: %s" % (spread, e, tol, msgs[0][:160]))
    assert e < tol and np.abs(fused - f64).max() < tol
    assert np.array_equal(unfused, unfused2) and np.array_equal(fused, fused2)
    # opt-out: static exponents, no warning
    runtime.reset()
    _, fn0 = SurfaceNet.SurfaceNet_inference(n_vp, None, param_values=values, auto_calibrate=False)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        _, unf0 = fn0(X, w)
    assert not [r for r in rec if issubclass(r.category, RuntimeWarning)]
    e0 = float(np.abs(unf0 - u64).max())
    print("   ... with auto_calibrate=False: L_inf %.3e" % e0)
    assert e0 > e


def test_hot_loop_and_sparse_loop_calibrate_on_their_first_batch(dropin):
    """The same through reconstruct.hot_loop (generator over batches) and reconstruct.SparseLoop.run / run_many (device-resident loop body): the first
    batch is checked, recalibrated and redone; one warning per loop; results equal a context calibrated by hand on that batch."""
    import warnings
    import surfacenet_amd
    from surfacenet_amd import reconstruct
    s, n_all, n_vp = 16, 7, 2
    values = _wide_net(10.0)
    sc = golden_util.synthetic_scene(n_all, n_vp, s=s, seed=4, hw=(600, 800))
    cubes_param_np = np.zeros(n_all, dtype=[("xyz", np.float32, (3,)), ("ijk", np.uint32, (3,)), ("resol", np.float32)])
    cubes_param_np["xyz"], cubes_param_np["resol"] = sc["xyz"], sc["resol"]
    valid = np.ones(n_all, dtype=bool)

    def fresh():
        ctx = surfacenet_amd.Context(cube_D=s, max_samples=8)
        ctx.load_param_values(values); ctx.set_cameras(sc["cams"]); ctx.set_images(sc["imgs"])
        return ctx
    # reference: calibrate by hand on the first batch (3 cubes), then run everything
    with fresh() as ctx:
        ctx.cvc_forward(sc["pairs"][:3], sc["xyz"][:3], sc["resol"][:3], sc["w"][:3])
        assert ctx.numeric_status() == ["merge_conv_a"]
        ctx.calibrate(0)
        want = [ctx.cvc_forward(sc["pairs"][i:i + 3], sc["xyz"][i:i + 3], sc["resol"][i:i + 3], sc["w"][i:i + 3])[0] for i in range(0, n_all, 3)]
        loop = reconstruct.SparseLoop(ctx, n_vp, max_cubes=3, min_prob=0.5, cube_Dcenter=12, auto_calibrate=False)
        want_sparse = loop.run_many(sc["pairs"], sc["xyz"], sc["resol"], sc["w"])
        loop.close()
    with fresh() as ctx, warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = [f for _, f, _, _ in reconstruct.hot_loop(ctx, valid, sc["pairs"], sc["w"], cubes_param_np, batch_size=3, return_cvc=False)]
        assert len([r for r in rec if issubclass(r.category, RuntimeWarning)]) == 1
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    with fresh() as ctx, warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        loop = reconstruct.SparseLoop(ctx, n_vp, max_cubes=3, min_prob=0.5, cube_Dcenter=12)
        got_sparse = loop.run_many(sc["pairs"], sc["xyz"], sc["resol"], sc["w"])
        first = loop.run(sc["pairs"][:3], sc["xyz"][:3], sc["resol"][:3], sc["w"][:3])     # calibrated already: silent
        loop.close()
        assert len([r for r in rec if issubclass(r.category, RuntimeWarning)]) == 1
    assert got_sparse[0] == want_sparse[0] and len(got_sparse[0]) > 0
    for k in (1, 2, 3, 4):
        assert all(np.array_equal(a, b) for a, b in zip(got_sparse[k], want_sparse[k]))
    assert first[0] == [i for i in want_sparse[0] if i < 3]
    with fresh() as ctx, warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        loop = reconstruct.SparseLoop(ctx, n_vp, max_cubes=3, min_prob=0.5, cube_Dcenter=12)
        one = loop.run(sc["pairs"][:3], sc["xyz"][:3], sc["resol"][:3], sc["w"][:3])       # run(): warns, recalibrates, redoes
        loop.close()
        assert len([r for r in rec if issubclass(r.category, RuntimeWarning)]) == 1
    assert one[0] == first[0] and all(np.array_equal(a, b) for a, b in zip(one[2], first[2]))


def test_cvc_forward_needs_weights_for_more_than_one_view_pair(gpu_required):
    """Context.cvc_forward with N_vp >= 2 and no w used to die inside numpy ("cannot reshape array of size 1"); the reference's loop always passes w
    (main_reconstruct.py:114-115, 145): a missing one is a TypeError that says so, before any device work."""
    import surfacenet_amd
    sc = golden_util.synthetic_scene(2, 2, s=16, seed=1, hw=(300, 400))
    with surfacenet_amd.Context(cube_D=16, max_samples=4) as ctx:
        ctx.set_cameras(sc["cams"]); ctx.set_images(sc["imgs"])
        with pytest.raises(TypeError, match="w .n, n_vp. float32 is required"):
            ctx.cvc_forward(sc["pairs"], sc["xyz"], sc["resol"])
        with pytest.raises(TypeError, match="must have shape"):
            ctx.cvc_forward(sc["pairs"], sc["xyz"], sc["resol"], np.ones((3,), np.float32))
