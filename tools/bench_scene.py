#!/usr/bin/env python3
"""tools/bench_scene.py — stage timing of reconstruct.reconstruct_scene on a synthetic DTU-like scene (not the headline bench).

    python tools/bench_scene.py --views 8 --cubes 2048 --cube-d 32

Synthetic 1200x1600 noise views on the first V cameras of the DTU rig (tests/golden/cameras.npz holds 4: they are reused
cyclically with a small rotation of the image plane so every view is distinct), cubes on a grid in front of the rig,
random-init weights of both networks. Prints one JSON line with the wall time of every stage of
main_reconstruct.py:67-173 as executed by the GPU drop-ins."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--cubes", type=int, default=2048)
    ap.add_argument("--cube-d", type=int, default=32)
    ap.add_argument("--n-vp", type=int, default=2)
    a = ap.parse_args()
    import golden_util
    from surfacenet_amd import SurfaceNet, camera, earlyRejection, reconstruct, runtime, similarityNet, viewPairSelection, weights

    P4 = golden_util.cameras()["P_dtu"]
    P = []
    for v in range(a.views):
        M = P4[v % 4].copy()
        th = 0.01 * (v // 4)
        R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
        P.append(R @ M)
    P = np.stack(P)
    imgs = [golden_util.synth_image(2000 + v, 1200, 1600) for v in range(a.views)]
    s, resol = a.cube_d, np.float32(0.4)
    cube_D_mm = resol * s
    g = int(np.ceil(a.cubes ** (1 / 3.0)))
    ijk = np.indices((g, g, g)).reshape(3, -1).T[: a.cubes]
    dt = [("xyz", np.float32, (3,)), ("ijk", np.uint32, (3,)), ("resol", np.float32)]
    cubes = np.empty((a.cubes,), dtype=dt)
    cubes["ijk"] = ijk
    cubes["xyz"] = (ijk * (cube_D_mm / 2) + np.array([-60.0, -60.0, 560.0])).astype(np.float32)
    cubes["resol"] = resol

    simil_values = weights.synthetic_simil_param_values(0)
    simil_values[28][:] = 3.0; simil_values[29][:] = -2.5          # puts the synthetic pair distances inside the accepted band
    runtime.DEFAULT_MAX_SAMPLES = 128
    p2e, pair_fn = similarityNet.similarityNet_inference(None, (64, 64), param_values=simil_values)
    relw_fn, _ = SurfaceNet.SurfaceNet_inference(a.n_vp, None, None, cube_D=s, param_values=weights.synthetic_param_values(0))
    mean_bgr = np.asarray([103.939, 116.779, 123.68], dtype=np.float32)
    Dc = {32: 26, 64: 52}.get(s, s - 4)

    t = {}
    clock = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        t[name] = round(now - clock[0], 4)
        clock[0] = now

    reconstruct.reconstruct_scene(imgs, P, cubes[:64], cube_D_mm, s, a.n_vp, p2e, pair_fn, relw_fn, cube_Dcenter=Dc, patches_mean_bgr=mean_bgr)   # warm-up
    clock[0] = time.perf_counter()
    # the stages of reconstruct_scene, timed one by one
    ih, iw = camera.perspectiveProj_cubesCorner(P, cubes["xyz"], cube_D_mm, return_int_hw=False)
    ch, cw = camera.perspectiveProj(P, cubes["xyz"] + cube_D_mm / 2., return_int_hw=False)
    lap("projections_host")
    emb, inscope = earlyRejection.patch2embedding(imgs, ih, iw, p2e, mean_bgr, a.cubes, a.views, 128, patchSize=64, batchSize=100,
                                                  cubeCenter_hw=np.stack([ch, cw], axis=0))
    lap("patch2embedding")
    viewPairs = viewPairSelection.k_combination_np(range(a.views), k=2)
    dis = earlyRejection.embeddingPairs2simil(embeddings=emb, embeddingPair2simil_fn=pair_fn, inScope_cubes_vs_views=inscope, viewPairs=viewPairs,
                                              N_views=a.views, batchSize=100000)
    valid = earlyRejection.selectFromSimilarity(dis, a.n_vp)
    lap("pair_similarity")
    vp, w = viewPairSelection.viewPairSelection(viewPairSelection.camera_centers(P), emb, dis, valid, cubes["xyz"] + cube_D_mm / 2., relw_fn, 100000, a.n_vp, viewPairs)
    lap("viewpair_selection")
    t0 = time.perf_counter()
    res = reconstruct.reconstruct_scene(imgs, P, cubes, cube_D_mm, s, a.n_vp, p2e, pair_fn, relw_fn, cube_Dcenter=Dc, patches_mean_bgr=mean_bgr)
    total = time.perf_counter() - t0
    front = sum(t.values())
    out = {"views": a.views, "cubes": a.cubes, "cube_D": s, "n_vp": a.n_vp, "in_scope_patches": int(inscope.sum()), "valid_cubes": int(valid.sum()),
           "stage_seconds": t, "loop_seconds": round(total - front, 4), "total_seconds": round(total, 4),
           "kept_voxels": int(sum(len(x) for x in res["prediction_list"])), "cubes_per_s_end_to_end": round(a.cubes / total, 1),
           "valid_cubes_per_s_in_loop": round(int(valid.sum()) / max(total - front, 1e-9), 1)}
    print(json.dumps(out))
    runtime.reset()


if __name__ == "__main__":
    main()
