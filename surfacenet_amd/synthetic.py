"""Synthetic inputs of the hot path for benches, smoke() and tests: seeded noise views, the DTU camera pair,
cube origins inside the view frusta, and the overlapping cube grid a scene's bounding box induces. Nothing here is read
from disk, so it travels to the GPU box as code.

    synthetic_scene   the 2-view workload of BASELINE configs[1] / SURVEY §8(d)
    dataset_scene     calibration + cube grid of BASELINE configs[2] / [4] (DTU scan9, Middlebury dino) with synthetic views
    cube_grid         the cube table of utils/scene.py:7-61 (`initializeCubes`) — an INPUT CONTRACT of the hot path
                      (structured dtype xyz f32x3 | ijk u32x3 | resol f32, k fastest); checked row for row against
                      tables produced by the reference itself (tests/golden/scene_cases.npz, tests/test_host_logic.py)
"""
import numpy as np

MEAN6 = np.asarray([123.68, 116.779, 103.939, 123.68, 116.779, 103.939]).astype(np.float32)      # params.py:129
CUBE_DTYPE = np.dtype([("xyz", np.float32, (3,)), ("ijk", np.uint32, (3,)), ("resol", np.float32)])   # utils/scene.py:55

# DTU calibration, views 1 and 2 (cal18/pos_001.txt, pos_002.txt: 12 numbers each)
P_DTU_12 = np.array([[[2607.429996, -3.844898, 1498.178098, -533936.661373],
                      [-192.07691, 2862.552532, 681.798177, 23434.686572],
                      [-0.241605, -0.030951, 0.969881, 22.540121]],
                     [[1977.784758, -1210.002933, 1915.072827, -784204.529312],
                      [976.333235, 2626.930526, 917.583083, -178022.203917],
                      [-0.416183, 0.073779, 0.906283, 71.276252]]], dtype=np.float64)


def synth_image(seed, H, W):
    """(H, W, 3) uint8 noise texture; the one image generator shared by fixtures, tests and benches."""
    return np.random.RandomState(int(seed)).randint(0, 256, (int(H), int(W), 3)).astype(np.uint8)


def synthetic_scene(n, n_vp=2, s=32, seed=0, hw=(1200, 1600), cams=None, n_views=2):
    """n cubes x n_vp view pairs over `n_views` noise views (default: the DTU pair): dict(cams, imgs, xyz, resol, pairs, w)."""
    cams = P_DTU_12 if cams is None else np.asarray(cams, dtype=np.float64)
    n_views = cams.shape[0]
    imgs = [synth_image(1234 + v, hw[0], hw[1]) for v in range(n_views)]
    rs = np.random.RandomState(seed)
    span = 0.4 * s
    xyz = (rs.rand(n, 3) * 40 + np.array([-20.0, -20.0, 580.0]) - np.array([0, 0, span / 2])).astype(np.float32)
    resol = np.full(n, 0.4, dtype=np.float32)
    if n_views == 2:
        pair_opts = np.array([[0, 1], [1, 0], [0, 0], [1, 1]], dtype=np.int64)
    else:
        pair_opts = np.array([[a, b] for a in range(n_views) for b in range(a + 1, n_views)], dtype=np.int64)
    pairs = np.stack([pair_opts[(np.arange(n_vp) + i) % len(pair_opts)] for i in range(n)]).astype(np.int64)
    w = (np.random.RandomState(seed + 1).rand(n, n_vp) + 0.1).astype(np.float32)
    return dict(cams=cams, imgs=imgs, xyz=xyz, resol=resol, pairs=pairs, w=w)


def cube_grid(resol, cube_D, cube_Dcenter, cube_overlapping_ratio, BB):
    """Overlapping cubes covering the bounding box BB (3,2) = [[x_min,x_max],...]: neighbours are cube_Dcenter*overlap voxels
    apart, the grid starts (cube_D - cube_Dcenter)/2 voxels before BB_min and ends that far past BB_max (utils/scene.py:43-58).
    Returns (cubes (N,) CUBE_DTYPE, cube_D_mm). Scalar dtypes are kept as the caller gives them (np.float32 resol in params.py)."""
    BB = np.asarray(BB)
    side, core = resol * cube_D, resol * cube_Dcenter
    step = core * cube_overlapping_ratio
    margin = (side - core) / 2
    counts = [int(np.ceil(((BB[ax][1] + margin) - (BB[ax][0] - margin)) / step)) for ax in range(3)]
    cubes = np.empty((counts[0] * counts[1] * counts[2],), dtype=CUBE_DTYPE)
    cubes["ijk"] = np.stack(np.unravel_index(np.arange(cubes.shape[0]), counts), axis=1)          # k fastest
    cubes["xyz"] = cubes["ijk"] * step + (BB[:, 0][None, :] - margin)
    cubes["resol"] = resol
    return cubes, side


def dataset_scene(config, cube_D=32, max_cubes=0, n_vp=0):
    """BASELINE configs[2] / configs[4] on their own calibration: all P matrices and the bounding box of DTU scan9 (49 views of
    1200x1600, resol 0.4 mm, N_viewPairs4inference 5; params.py:165-172) or Middlebury dinoSparseRing (16 views of 480x640, resol
    0.00025, 16 view pairs; params.py:176-182) from surfacenet_amd/data/calibration.npz (read by the reference's own readers in
    oracle/gen_golden_scene.py), the reference's overlapping cube grid (`cube_grid`, overlap 1/2: params.py:114) and seeded noise views
    (no dataset pixels on the GPU box). `max_cubes` > 0 takes an even sample of the grid.
    Returns (P (V,3,4) f64, images list, cubes CUBE_DTYPE, cube_D_mm, cube_Dcenter, N_viewPairs4inference)."""
    import os
    cal = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "calibration.npz"))
    Dc = {32: 26, 64: 52}.get(cube_D, cube_D - 4)                                                   # params.py:107
    if config == "dtu_scan9":
        P, hw, resol, BB, n_vp = cal["P_dtu49"], (1200, 1600), np.float32(0.4), cal["scan9_BB"], n_vp or 5
    elif config == "dino":
        P, hw, resol, BB, n_vp = cal["P_mid16"], (480, 640), np.float32(0.00025), cal["dino_BB"], n_vp or 16
    else:
        raise ValueError("config must be 'dtu_scan9' or 'dino'")
    cubes, cube_D_mm = cube_grid(resol, cube_D, Dc, 1 / 2., BB)
    if max_cubes:
        cubes = cubes[np.linspace(0, len(cubes) - 1, min(int(max_cubes), len(cubes))).astype(np.int64)]
    imgs = [synth_image(2000 + v, hw[0], hw[1]) for v in range(P.shape[0])]
    return np.asarray(P, dtype=np.float64), imgs, cubes, cube_D_mm, Dc, n_vp
