#!/usr/bin/env python3
"""tools/bench_scene.py — end-to-end reconstruct.reconstruct_scene at the reference's operating points (not the headline bench).

    python tools/bench_scene.py --config dtu_scan9            # BASELINE configs[2]: DTU scan9 full bounding box, s=32, 49 views,
                                                              # N_viewPairs4inference = 5 (params.py:165), 195,360 cubes
    python tools/bench_scene.py --config dino                 # BASELINE configs[4] on one GPU: Middlebury dinoSparseRing, 16 views,
                                                              # s=32, 16 view pairs, early rejection + view-pair selection active
    python tools/bench_scene.py --config synthetic --views 8 --cubes 2048

What is real: the calibration (all P matrices, bounding boxes: surfacenet_amd/data/calibration.npz, read by the reference's own
readers in oracle/gen_golden_scene.py), the cube grid (synthetic.cube_grid == the reference's scene.initializeCubes, row for row),
image sizes, every stage of main_reconstruct.py:67-173 as executed by the GPU drop-ins. What is synthetic: the views (seeded noise
textures; no dataset images on the GPU box) and the weights of both networks (random init; the logistic unit of the similarityNet is
set so that a plausible share of the cubes survives early rejection). Prints one JSON line with the wall seconds of every stage.
--max-cubes N takes the first N cubes of the grid (smoke runs)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_scene(a):
    from surfacenet_amd import synthetic
    if a.config in ("dtu_scan9", "dino"):
        return synthetic.dataset_scene(a.config, a.cube_d, a.max_cubes, a.n_vp)
    cal = np.load(os.path.join(ROOT, "surfacenet_amd", "data", "calibration.npz"))
    s = a.cube_d
    Dc = {32: 26, 64: 52}.get(s, s - 4)                                      # params.py:107
    P4 = cal["P_dtu49"][:4]
    P = np.stack([np.array([[np.cos(0.01 * (v // 4)), -np.sin(0.01 * (v // 4)), 0], [np.sin(0.01 * (v // 4)), np.cos(0.01 * (v // 4)), 0],
                            [0, 0, 1]]) @ P4[v % 4] for v in range(a.views)])
    hw, resol, n_vp = (1200, 1600), np.float32(0.4), a.n_vp or 2
    g = int(np.ceil(a.cubes ** (1 / 3.0)))
    ijk = np.indices((g, g, g)).reshape(3, -1).T[: a.cubes]
    cubes = np.empty((a.cubes,), dtype=synthetic.CUBE_DTYPE)
    cube_D_mm = resol * s
    cubes["ijk"], cubes["resol"] = ijk, resol
    cubes["xyz"] = (ijk * (cube_D_mm / 2) + np.array([-60.0, -60.0, 560.0])).astype(np.float32)
    if a.max_cubes:
        cubes = cubes[np.linspace(0, len(cubes) - 1, min(a.max_cubes, len(cubes))).astype(np.int64)]   # an even sample of the grid
    imgs = [synthetic.synth_image(2000 + v, hw[0], hw[1]) for v in range(P.shape[0])]
    return P, imgs, cubes, cube_D_mm, Dc, n_vp


def run(argv):
    """-> the result dict (bench.py embeds bounded samples of the two dataset configurations in its JSON line)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="synthetic", choices=["synthetic", "dtu_scan9", "dino"])
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--cubes", type=int, default=2048)
    ap.add_argument("--cube-d", type=int, default=32)
    ap.add_argument("--n-vp", type=int, default=0)
    ap.add_argument("--max-cubes", type=int, default=0)
    ap.add_argument("--max-samples", type=int, default=0, help="cube-view-pair samples the scene's context is sized for (0: 128, or 8 x n_vp above 8 view pairs)")
    ap.add_argument("--batch", type=int, default=0, help="cubes per SurfaceNet batch (default: max_samples / n_vp; the reference's is 14 at s=32, params.py:117-118)")
    a = ap.parse_args(argv)
    from surfacenet_amd import SurfaceNet, reconstruct, runtime, similarityNet, weights

    t0 = time.perf_counter()
    P, imgs, cubes, cube_D_mm, Dc, n_vp = build_scene(a)
    t_build = time.perf_counter() - t0
    simil_values = weights.synthetic_simil_param_values(0)
    simil_values[28][:] = 3.0; simil_values[29][:] = -2.5          # puts the synthetic pair distances inside the accepted band
    runtime.DEFAULT_MAX_SAMPLES = a.max_samples or (128 if n_vp <= 8 else 8 * n_vp)
    p2e, pair_fn = similarityNet.similarityNet_inference(None, (64, 64), param_values=simil_values)
    relw_fn, _ = SurfaceNet.SurfaceNet_inference(n_vp, None, None, cube_D=a.cube_d, param_values=weights.synthetic_param_values(0))
    mean_bgr = np.asarray([103.939, 116.779, 123.68], dtype=np.float32)
    kw = dict(cube_Dcenter=Dc, patches_mean_bgr=mean_bgr, batchSize_nViewPair_SurfaceNet=(a.batch or None))
    reconstruct.reconstruct_scene(imgs, P, cubes[:64], cube_D_mm, a.cube_d, n_vp, p2e, pair_fn, relw_fn, **kw)      # warm-up
    stages = {}
    t0 = time.perf_counter()
    res = reconstruct.reconstruct_scene(imgs, P, cubes, cube_D_mm, a.cube_d, n_vp, p2e, pair_fn, relw_fn, timings=stages, **kw)
    total = time.perf_counter() - t0
    n_valid = int(res["validCubes"].sum())
    out = {"config": a.config, "views": int(P.shape[0]), "view_pairs_all": int(P.shape[0] * (P.shape[0] - 1) // 2), "cubes": int(len(cubes)), "cube_D": a.cube_d,
           "n_vp": n_vp, "image_hw": list(imgs[0].shape[:2]), "in_scope_patches": int(res["inScope_cubes_vs_views"].sum()), "valid_cubes": n_valid,
           "nonempty_cubes": len(res["prediction_list"]), "kept_voxels": int(sum(len(x) for x in res["prediction_list"])),
           "stage_seconds": {k: round(v, 4) for k, v in stages.items()}, "total_seconds": round(total, 3), "scene_build_seconds_host": round(t_build, 2),
           "cubes_per_s_end_to_end": round(len(cubes) / total, 1),
           "valid_cubes_per_s_in_loop": round(n_valid / max(stages.get("cube_loop", 0.0), 1e-9), 1),
           "patches_per_s": round(int(res["inScope_cubes_vs_views"].sum()) / max(stages.get("patch2embedding", 0.0), 1e-9), 1),
           "data": "calibration + cube grid of the dataset; synthetic noise views and random-init networks"}
    runtime.reset()
    return out


if __name__ == "__main__":
    print(json.dumps(run(sys.argv[1:])))
