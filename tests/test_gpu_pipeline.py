"""GPU (-m gpu): reconstruct.reconstruct_scene (early rejection -> view-pair selection -> hot loop -> sparse lists, all on
the GPU) against the same pipeline written the way main_reconstruct.py:67-173 writes it, with the drop-in modules called
one by one through host arrays (each of them is parity-tested on its own against the reference goldens / the oracle)."""
import numpy as np
import pytest

import golden_util

pytestmark = pytest.mark.gpu
PARAM_DT = [("xyz", np.float32, (3,)), ("ijk", np.uint32, (3,)), ("resol", np.float32)]
MEAN_BGR = np.asarray([103.939, 116.779, 123.68]).astype(np.float32)
MEAN_CVC = np.asarray([123.68, 116.779, 103.939, 123.68, 116.779, 103.939]).astype(np.float32)


def test_reconstruct_scene_equals_reference_style_script(gpu_required):
    import synth
    from surfacenet_amd import CVC, SurfaceNet, camera, earlyRejection, reconstruct, runtime, similarityNet, sparseCubes, viewPairSelection, weights
    runtime.reset()
    cube_D, Dc, N_vp, resol = 16, 12, 2, np.float32(0.8)
    cube_D_mm = resol * cube_D
    P = golden_util.cameras()["P_dtu"][:4].copy()
    P[:, :2, :] *= 0.5                                                     # 600x800 synthetic views of the DTU rig
    imgs = [golden_util.synth_image(900 + v, 600, 800) for v in range(4)]
    rs = np.random.RandomState(3)
    N_cubes = 12
    cubes = np.empty((N_cubes,), dtype=PARAM_DT)
    cubes["xyz"] = (rs.rand(N_cubes, 3) * [80, 80, 40] + [-40, -40, 590]).astype(np.float32)
    cubes["xyz"][5] = [2000, 2000, 100]                                    # projects outside every view
    cubes["ijk"] = rs.randint(0, 40, (N_cubes, 3))
    cubes["resol"] = resol
    net_values = list(synth.calibrated_params(1))
    simil_values = weights.synthetic_simil_param_values(6)
    # make the logistic unit map the synthetic embedding distances into the accepted band 0.1 < p < 0.5
    simil_values[28][:] = 3.0; simil_values[29][:] = -2.5   # identical (all-black) patches: sigmoid(-2.5) < 0.1 -> rejected
    p2e, pair_fn = similarityNet.similarityNet_inference(None, (64, 64), param_values=simil_values)
    relw_fn, net_fn = SurfaceNet.SurfaceNet_inference(N_vp, None, None, cube_D=cube_D, param_values=net_values)
    min_prob, tau, gamma = 0.5, 0.6, 0.5

    res = reconstruct.reconstruct_scene(imgs, P, cubes, cube_D_mm, cube_D, N_vp, p2e, pair_fn, relw_fn, cube_Dcenter=Dc, patches_mean_bgr=MEAN_BGR,
                                        batchSize_similNet_patch2embedding=5, batchSize_similNet_embeddingPair2simil=7, batchSize_viewPair_w=13,
                                        batchSize_nViewPair_SurfaceNet=4, min_prob=min_prob, tau=tau, gamma=gamma)

    # ---- the reference's script, line by line, on the drop-in modules (host arrays everywhere) ----
    cameraTs = viewPairSelection.camera_centers(P)
    ih, iw = camera.perspectiveProj_cubesCorner(P, cubes["xyz"], cube_D_mm, return_int_hw=False)
    ch, cw = camera.perspectiveProj(P, cubes["xyz"] + cube_D_mm / 2., return_int_hw=False)
    viewPairs = viewPairSelection.k_combination_np(range(4), k=2)
    plain_p2e = lambda x: p2e(x)                                            # three-step crop / preprocess / embed protocol
    emb, inscope = earlyRejection.patch2embedding(imgs, ih, iw, plain_p2e, MEAN_BGR, N_cubes, 4, 128, patchSize=64, batchSize=5,
                                                  cubeCenter_hw=np.stack([ch, cw], axis=0))
    dis = earlyRejection.embeddingPairs2simil(embeddings=emb, embeddingPair2simil_fn=lambda e: pair_fn(e), inScope_cubes_vs_views=inscope, viewPairs=viewPairs,
                                              N_views=4, batchSize=7)
    valid = earlyRejection.selectFromSimilarity(dis, N_vp)
    assert 0 < valid.sum() < N_cubes and not valid[5], np.round(dis, 3)    # the early rejection really rejects something
    plain_relw = lambda f, n_samples_perGroup: relw_fn(f, n_samples_perGroup)      # explicit (N*P, 258) feature rows, batched
    vp, w = viewPairSelection.viewPairSelection(cameraTs, emb, dis, valid, cubes["xyz"] + cube_D_mm / 2., plain_relw, 13, N_vp, viewPairs)
    lists = ([], [], [], [], None, None, None)
    p_l, rgb_l, ijk_l, v_l, cube_ijk, param_np, vp_np = lists
    for _batch in reconstruct.gen_non0Batch_npBool(valid, 4):
        X1 = CVC.gen_coloredCubes(selected_viewPairs=vp[_batch[valid]], xyz=cubes["xyz"][_batch], resol=cubes["resol"][_batch], colorize_cube_D=cube_D,
                                  cameraPOs=P, models_img=imgs, visualization_ON=False)
        _, X2 = CVC.preprocess_augmentation(None, X1, mean_rgb=MEAN_CVC[None, :, None, None, None], augment_ON=False, crop_ON=False)
        fused, unfused = net_fn(X2, w[_batch[valid]])
        rgb = runtime.context_for(cube_D).color_fuse(X2, unfused, w[_batch[valid]])
        p_l, rgb_l, ijk_l, v_l, cube_ijk, param_np, vp_np = sparseCubes.append_dense_2sparseList(
            prediction_sub=fused, rgb_sub=rgb, param_sub=cubes[_batch], viewPair_sub=vp[_batch[valid]], min_prob=min_prob, rayPool_thresh=0,
            enable_centerCrop=True, cube_Dcenter=Dc, enable_rayPooling=True, cameraPOs=P, cameraTs=cameraTs, prediction_list=p_l, rgb_list=rgb_l,
            vxl_ijk_list=ijk_l, rayPooling_votes_list=v_l, cube_ijk_np=cube_ijk, param_np=param_np, viewPair_np=vp_np)
    masks = sparseCubes.filter_voxels(vxl_mask_list=[], prediction_list=p_l, prob_thresh=tau, rayPooling_votes_list=v_l, rayPool_thresh=gamma * N_vp * 2)

    assert np.array_equal(res["patches_embedding"], emb) and np.array_equal(res["dissimilarity"], dis) and np.array_equal(res["validCubes"], valid)
    assert np.array_equal(res["viewPairs4Reconstr"], vp) and np.array_equal(res["w_viewPairs4Reconstr"], w)
    assert len(res["prediction_list"]) == len(p_l) > 0
    for key, want in (("prediction_list", p_l), ("rgb_list", rgb_l), ("vxl_ijk_list", ijk_l), ("rayPooling_votes_list", v_l), ("vxl_mask_list", masks)):
        assert all(np.array_equal(a, b) for a, b in zip(res[key], want)), key
    assert np.array_equal(res["cube_ijk_np"], cube_ijk) and np.array_equal(res["viewPair_np"], vp_np)
    assert np.array_equal(res["param_np"]["xyz"], param_np["xyz"]) and np.array_equal(res["param_np"]["resol"], param_np["resol"])
    assert sum(int(m.sum()) for m in masks) > 0
    runtime.reset()


def _pipeline_inputs():
    """A small synthetic scene on the DTU rig: 4 views, 12 cubes (one projects outside every view), synthetic nets."""
    import synth
    from surfacenet_amd import weights
    cube_D, Dc, N_vp, resol = 16, 12, 2, np.float32(0.8)
    P = golden_util.cameras()["P_dtu"][:4].copy()
    P[:, :2, :] *= 0.5
    imgs = [golden_util.synth_image(900 + v, 600, 800) for v in range(4)]
    rs = np.random.RandomState(3)
    N_cubes = 12
    cubes = np.empty((N_cubes,), dtype=PARAM_DT)
    cubes["xyz"] = (rs.rand(N_cubes, 3) * [80, 80, 40] + [-40, -40, 590]).astype(np.float32)
    cubes["xyz"][5] = [2000, 2000, 100]
    cubes["ijk"] = rs.randint(0, 40, (N_cubes, 3))
    cubes["resol"] = resol
    simil_values = weights.synthetic_simil_param_values(6)
    simil_values[28][:] = 3.0; simil_values[29][:] = -2.5
    return dict(cube_D=cube_D, Dc=Dc, N_vp=N_vp, cube_D_mm=resol * cube_D, P=P, imgs=imgs, cubes=cubes, net_values=list(synth.calibrated_params(1)),
                simil_values=simil_values)


def _run_scene(inp, sharded):
    from surfacenet_amd import SurfaceNet, reconstruct, runtime, similarityNet
    runtime.reset()
    p2e, pair_fn = similarityNet.similarityNet_inference(None, (64, 64), param_values=inp["simil_values"])
    relw_fn, _ = SurfaceNet.SurfaceNet_inference(inp["N_vp"], None, None, cube_D=inp["cube_D"], param_values=inp["net_values"])
    fn = reconstruct.reconstruct_scene_sharded if sharded else reconstruct.reconstruct_scene
    extra = dict(gather_intermediates=True) if sharded else {}      # the test compares the per-cube embeddings / dissimilarities too
    res = fn(inp["imgs"], inp["P"], inp["cubes"], inp["cube_D_mm"], inp["cube_D"], inp["N_vp"], p2e, pair_fn, relw_fn, cube_Dcenter=inp["Dc"],
             patches_mean_bgr=MEAN_BGR, batchSize_nViewPair_SurfaceNet=4, min_prob=0.5, tau=0.6, gamma=0.5, **extra)
    runtime.reset()
    return res


def _sharded_worker(rank, world, port, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    import test_gpu_pipeline as T
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)       # both ranks share the box's one GPU; the exchange runs on gloo
    res = T._run_scene(T._pipeline_inputs(), sharded=True)
    q.put((rank, res))
    dist.destroy_process_group()


def test_sharded_scene_two_processes_equals_single_process(gpu_required):
    """reconstruct_scene_sharded with 2 ranks (two processes on this box's GPU, gloo exchange; early rejection + selection on the raw cube
    shards 0-5 / 6-11, the cube loop on halves of the VALID list) returns on every rank exactly what the single-process reconstruct_scene
    returns - the real GPU pipeline in both stages on both ranks."""
    import os
    import torch.multiprocessing as mp
    import test_dist_cpu
    want = _run_scene(_pipeline_inputs(), sharded=False)
    assert want["validCubes"].sum() > 0 and len(want["prediction_list"]) > 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n_valid = int(want["validCubes"].sum())
    for rank, full in res:
        test_dist_cpu._same_scene(full, want)
        assert full["cubes_per_rank"] == [(6, n_valid - n_valid // 2), (6, n_valid // 2)]        # the loop is cut over the valid list, to within one cube


def test_native_allgatherv_and_native_sharded_scene_one_rank(gpu_required):
    """SURVEY §8(e) "counts then all-gather-v" through the C ABI (sn_allgatherv_bytes_dev: RCCL on the context's stream) with a one-rank
    communicator - all this box offers: blobs of assorted lengths (0, 1, unaligned, 1 MB) come back unchanged, and reconstruct_scene_sharded with
    comm='native' (both exchanges through the library, torch.distributed never imported) equals the single-process reconstruct_scene; also the
    overlapped f32 all-gather (sn_allgather_f32_dev_overlap / sn_comm_wait) against the in-order one."""
    import test_dist_cpu
    from surfacenet_amd import Context, SurfaceNet, reconstruct, runtime, similarityNet
    inp = _pipeline_inputs()
    want = _run_scene(inp, sharded=False)
    runtime.reset()
    try:
        p2e, pair_fn = similarityNet.similarityNet_inference(None, (64, 64), param_values=inp["simil_values"])
        relw_fn, _ = SurfaceNet.SurfaceNet_inference(inp["N_vp"], None, None, cube_D=inp["cube_D"], param_values=inp["net_values"])
        ctx = runtime.context_for(inp["cube_D"])
        ctx.comm_init(1, 0, Context.comm_unique_id())
        rs = np.random.RandomState(5)
        for size in (0, 1, 17, 4099, 1 << 20):
            blob = rs.randint(0, 256, size).astype(np.uint8)
            got = ctx.allgatherv_bytes(blob)
            assert len(got) == 1 and got[0].dtype == np.uint8 and np.array_equal(got[0], blob)
        a = rs.rand(1 << 16).astype(np.float32)
        d_a, d_g1, d_g2 = ctx.upload(a), ctx.dev_alloc(a.nbytes), ctx.dev_alloc(a.nbytes)
        ctx.allgather_f32_dev(d_a, a.size, d_g1)
        ctx.allgather_f32_dev_overlap(d_a, a.size, d_g2, 3)
        ctx.comm_wait(3); ctx.comm_wait(4)                          # (slot 4 was never used: a no-op)
        g1, g2 = np.empty_like(a), np.empty_like(a)
        ctx.d2h(g1, d_g1); ctx.d2h(g2, d_g2)
        assert np.array_equal(g1, a) and np.array_equal(g2, a)
        for p in (d_a, d_g1, d_g2):
            ctx.dev_free(p)
        res = reconstruct.reconstruct_scene_sharded(inp["imgs"], inp["P"], inp["cubes"], inp["cube_D_mm"], inp["cube_D"], inp["N_vp"], p2e, pair_fn, relw_fn,
                                                    cube_Dcenter=inp["Dc"], patches_mean_bgr=MEAN_BGR, batchSize_nViewPair_SurfaceNet=4, min_prob=0.5, tau=0.6,
                                                    gamma=0.5, gather_intermediates=True, ctx=ctx, comm="native")
        test_dist_cpu._same_scene(res, want)
        assert res["cubes_per_rank"] == [(len(inp["cubes"]), int(want["validCubes"].sum()))]
    finally:
        runtime.reset()


def test_marked_readback_sees_its_own_batch(gpu_required):
    """sn_mark / sn_memcpy_d2h_after (reconstruct.SparseLoop.run_many's readback): the copy waits for the work enqueued BEFORE the mark
    only - it returns the data as of the mark even when later work on the context's stream has already been enqueued to overwrite
    the buffer's source - and an unmarked slot is an error, not a hang."""
    import surfacenet_amd as sn
    with sn.Context(cube_D=8, max_samples=2) as ctx:
        a = np.arange(1 << 20, dtype=np.float32)
        b = -a
        dev = ctx.dev_alloc(a.nbytes)
        keep = ctx.dev_alloc(a.nbytes)
        got = np.empty_like(a)
        with pytest.raises(sn.SurfaceNetHipError):
            ctx.d2h_after(5, got, dev)                          # slot 5 was never marked
        with pytest.raises(sn.SurfaceNetHipError):
            ctx.mark(8)
        ctx.h2d(dev, a)
        ctx.mark(0)
        ctx.h2d(keep, b)                                        # later work on the stream (does not touch dev)
        ctx.d2h_after(0, got, dev)
        assert np.array_equal(got, a)
        ctx.h2d(dev, b)
        ctx.mark(1)
        ctx.d2h_after(1, got, dev)
        assert np.array_equal(got, b)
        ctx.d2h_after(0, got, keep)                             # an old mark stays valid: everything before it has long completed
        assert np.array_equal(got, b)
        ctx.dev_free(dev); ctx.dev_free(keep)


def test_fused_early_rejection_with_cube_centres_sharing_a_pixel(gpu_required):
    """earlyRejection.patch2embedding, fused path (round 6): every row of the (cubes, views, 128) array is written once - in-scope rows with the view's
    embeddings, out-of-scope rows with the black-patch embedding - and cubes whose centre projections TRUNCATE to the same pixel of a view (utils/image.py:
    160-169) are embedded once. Forced here: three cubes share their centre up to a fraction of a pixel, one is an exact duplicate, one view sees
    nothing, one cube is out of every view; the result must equal the reference's three-step protocol (crop -> preprocess -> batched network calls on host
    arrays) bit for bit."""
    from surfacenet_amd import camera, earlyRejection, runtime, similarityNet, weights
    runtime.reset()
    cube_D_mm = np.float32(12.8)
    P = golden_util.cameras()["P_dtu"][:4].copy()
    P[:, :2, :] *= 0.5
    imgs = [golden_util.synth_image(700 + v, 600, 800) for v in range(4)]
    rs = np.random.RandomState(11)
    N = 40
    xyz = (rs.rand(N, 3) * [80, 80, 40] + [-40, -40, 590]).astype(np.float32)
    xyz[7] = [2000, 2000, 100]                                             # out of every view
    simil_values = weights.synthetic_simil_param_values(2)
    p2e, _ = similarityNet.similarityNet_inference(None, (64, 64), param_values=simil_values)
    ih, iw = camera.perspectiveProj_cubesCorner(P, xyz, cube_D_mm, return_int_hw=False)
    ch, cw = camera.perspectiveProj(P, xyz + cube_D_mm / 2., return_int_hw=False)
    centres = np.stack([ch, cw], axis=0)                                   # (2, views, cubes)
    for dup, src, dh, dw in ((11, 3, 0.0, 0.0), (12, 3, 0.3, -0.0), (13, 3, 0.6, 0.4), (20, 19, 0.0, 0.0)):
        base_h, base_w = np.floor(centres[0, :, src]) + 0.05, np.floor(centres[1, :, src]) + 0.05
        centres[0, :, src], centres[1, :, src] = base_h, base_w
        centres[0, :, dup], centres[1, :, dup] = base_h + dh, base_w + dw      # same truncated pixel in every view
        ih[:, dup], iw[:, dup] = ih[:, src], iw[:, src]                          # ... and the same in-scope verdicts
    ih[2] += 5000.0                                                        # view 2 sees no cube at all
    assert getattr(p2e, "sn_gpu", False)
    got, got_in = earlyRejection.patch2embedding(imgs, ih, iw, p2e, MEAN_BGR, N, 4, 128, patchSize=64, batchSize=7, cubeCenter_hw=centres)
    want, want_in = earlyRejection.patch2embedding(imgs, ih, iw, lambda x: p2e(x), MEAN_BGR, N, 4, 128, patchSize=64, batchSize=7, cubeCenter_hw=centres)
    assert np.array_equal(got_in, want_in) and not got_in[:, 2].any() and not got_in[7].any() and got_in[:, 0].sum() > 20
    assert got_in[3, 0] and got_in[11, 0] and got_in[12, 0] and got_in[13, 0]
    assert np.array_equal(got, want)
    assert np.array_equal(got[3, 0], got[12, 0]) and np.array_equal(got[3, 0], got[13, 0]) and np.array_equal(got[19, 1], got[20, 1])
    assert not np.array_equal(got[3, 0], got[4, 0])
