"""Host side of SURVEY §8(f) row N1: the producer of the hot path's `viewPairs` / `w` inputs.

Mirrors `utils/viewPairSelection.py` of the reference (same names, argument meaning, return conventions); the one
compiled function it calls, `viewPair_relativeImpt_fn` (nets/SurfaceNet.py:334-338), is the MI355X kernel pair behind
`sn_relative_weights` (returned by `surfacenet_amd.SurfaceNet.SurfaceNet_inference`). Everything else here is the
reference's small numpy bookkeeping — O(N_cubes * N_viewPairs) index work that stays on the host:

    k_combination_np        utils/utils.py:233-257
    yield_batch_npBool      utils/utils.py:149-177 (+ gen_batch_index :113-131)
    viewPairAngles_wrt_pts  utils/camera.py:275-309
    camera_centers          (what utils/camera.py:87-120 cameraPs2Ts provides to the caller)
    __argmaxN_viewPairs__   utils/viewPairSelection.py:8-41   (ascending order kept: the largest weight is LAST)
    viewPairSelection       utils/viewPairSelection.py:44-82
"""
import itertools
import math

import numpy as np


def k_combination_np(iterable, k=2):
    """All k-combinations along the rows, e.g. [2,5,8] -> [[2,5],[2,8],[5,8]] (utils/utils.py:233-257)."""
    return np.asarray(list(itertools.combinations(iterable, k)))


def yield_batch_npBool(N_all, batch_size):
    """Bool selectors (N_all,) of consecutive batches; the last may be short (utils/utils.py:113-131,149-177)."""
    N_all, batch_size = int(N_all), int(batch_size)
    for start in range(0, N_all, batch_size):
        sel = np.zeros((N_all,), dtype=bool)
        sel[start:min(N_all, start + batch_size)] = True
        yield sel


def camera_centers(cameraPOs):
    """Camera centres (V,3) of projection matrices (V,3,4): the right null vector of P = [M | p4], i.e. C = -M^-1 p4, one batched
    3x3 solve. (What the reference's camera.cameraPs2Ts, utils/camera.py:87-120, obtains from four 3x3 determinants; equal up to
    rounding. Only the view-pair angles below consume it.)"""
    P = np.asarray(cameraPOs, dtype=np.float64).reshape((-1, 3, 4))
    return np.linalg.solve(P[:, :, :3], -P[:, :, 3:])[..., 0]


def viewPairAngles_wrt_pts(cameraTs, pts_xyz):
    """Angle <c_i, p, c_j> for every 2-combination of views and every point: (N_pts, N_viewPairs)
    (utils/camera.py:275-309; dtype follows the inputs exactly as there). The dot products are formed component by component on contiguous arrays - the
    same three products summed in the same order as the reference's `np.sum(np.multiply(u[:, i], u[:, j]), axis=-1)`, bit for bit, without its three
    (N_pts, N_viewPairs, 3) temporaries (DTU scan9: 23,347 points x 1,176 pairs, 11x faster; round 6)."""
    cameraTs = np.asarray(cameraTs)
    pts_xyz = np.asarray(pts_xyz)
    v = pts_xyz[:, None, :] - cameraTs[None, ...]                           # (N_pts, N_views, 3)
    u = v / np.linalg.norm(v, axis=-1, ord=2, keepdims=True)
    pairs = k_combination_np(range(cameraTs.shape[0]), k=2)
    if pairs.size == 0 or u.shape[0] == 0:
        cos = np.sum(np.multiply(u[:, pairs[:, 0]], u[:, pairs[:, 1]]), axis=-1) if pairs.size else np.zeros((u.shape[0], 0), dtype=u.dtype)
        return np.arccos(np.clip(cos, -1.0, 1.0))
    p0, p1 = pairs[:, 0], pairs[:, 1]
    ux, uy, uz = (np.ascontiguousarray(u[..., k]) for k in range(3))
    cos = ux[:, p0] * ux[:, p1]
    cos += uy[:, p0] * uy[:, p1]
    cos += uz[:, p0] * uz[:, p1]
    np.clip(cos, -1.0, 1.0, out=cos)
    return np.arccos(cos, out=cos)


def __argmaxN_viewPairs__(viewPairs, w_viewPairs, N_argmax):
    """viewPairs (P,2), w (N,P) -> (N, N_argmax, 2) pairs and (N, N_argmax) weights of the N_argmax largest weights
    per cube, in ASCENDING weight order (utils/viewPairSelection.py:8-41: `w.argsort(axis=1)[:, -N_argmax:]`).
    Same result without sorting every row in full (DTU scan9: 23,347 valid cubes x 1,176 pairs - the full argsort was 1 s of the scene): the N largest by
    `argpartition`, sorted among themselves; a row whose choice or order could depend on how argsort breaks TIES (equal values inside the selection or at
    its boundary, NaNs) is redone by the reference's own expression."""
    viewPairs = np.asarray(viewPairs)
    w_viewPairs = np.asarray(w_viewPairs)
    N_validCubes, P = w_viewPairs.shape
    N_argmax = int(N_argmax)
    if N_validCubes == 0 or N_argmax <= 0 or P < 8 * N_argmax:
        indice_N_max = w_viewPairs.argsort(axis=1)[:, -1 * N_argmax:]
    else:
        part = np.argpartition(w_viewPairs, P - N_argmax, axis=1)[:, P - N_argmax:]
        vals = np.take_along_axis(w_viewPairs, part, axis=1)
        order = np.argsort(vals, axis=1, kind="stable")
        indice_N_max = np.take_along_axis(part, order, axis=1)
        vals = np.take_along_axis(vals, order, axis=1)
        ties = (np.diff(vals, axis=1) == 0).any(axis=1) | ((w_viewPairs >= vals[:, :1]).sum(axis=1) != N_argmax)
        if ties.any():
            indice_N_max[ties] = w_viewPairs[ties].argsort(axis=1)[:, -1 * N_argmax:]
    return viewPairs[indice_N_max], np.take_along_axis(w_viewPairs, indice_N_max, axis=1)


argmaxN_viewPairs = __argmaxN_viewPairs__


def viewPairSelection(cameraTs_np, e_viewPairs, d_viewPairs, validCubes, cubeCenters_xyz, viewPair_relativeImpt_fn, batchSize,
                      N_viewPairs4inference, viewPairs):
    """utils/viewPairSelection.py:44-82. e_viewPairs (N_cubes, N_views, D_emb) patch embeddings, d_viewPairs
    (N_cubes, P) pair dissimilarities, validCubes (N_cubes,) bool, cubeCenters_xyz (N_cubes,3), viewPairs (P,2).
    Returns selected pairs (N_valid, N_viewPairs4inference, 2) and their (un-renormalised) softmax weights."""
    validCubes = np.asarray(validCubes).astype(bool)
    viewPairs = np.asarray(viewPairs)
    N_viewPairs = d_viewPairs.shape[1]
    N_validCubes = int(validCubes.sum())
    D_embedding = e_viewPairs.shape[-1]
    theta = viewPairAngles_wrt_pts(cameraTs=cameraTs_np, pts_xyz=cubeCenters_xyz[validCubes])[..., None]
    d = d_viewPairs[validCubes][..., None]
    e_valid = e_viewPairs[validCubes]
    w_viewPairs = np.empty((N_validCubes, N_viewPairs), dtype=np.float32)
    N_views = e_viewPairs.shape[1]
    if getattr(viewPair_relativeImpt_fn, "sn_gpu", False) and N_validCubes and viewPairs.shape == (N_views * (N_views - 1) // 2, 2) \
            and np.array_equal(viewPairs, k_combination_np(range(N_views), k=2)):
        # all 2-combinations in combinations order: the GPU assembles the feature rows itself (bit-identical to the loop below)
        from . import runtime
        ctx = runtime.any_context() if viewPair_relativeImpt_fn.sn_cube_D is None else runtime.context_for(viewPair_relativeImpt_fn.sn_cube_D)
        w_viewPairs = ctx.viewpair_weights(e_valid, d[..., 0].astype(np.float32), theta[..., 0].astype(np.float32))
        return __argmaxN_viewPairs__(viewPairs=viewPairs, w_viewPairs=w_viewPairs, N_argmax=N_viewPairs4inference)
    for _batch in yield_batch_npBool(N_all=N_validCubes, batch_size=int(math.floor(float(batchSize) / N_viewPairs))):
        N_batch = int(_batch.sum())
        e = e_valid[_batch][:, viewPairs.flatten()].reshape((N_batch, N_viewPairs, 2 * D_embedding))
        N_features = 2 * D_embedding + 2
        features = np.concatenate([e, d[_batch], theta[_batch]], axis=-1).astype(np.float32).reshape((N_batch * N_viewPairs, N_features))
        w_viewPairs[_batch] = viewPair_relativeImpt_fn(features, n_samples_perGroup=N_viewPairs)
    return __argmaxN_viewPairs__(viewPairs=viewPairs, w_viewPairs=w_viewPairs, N_argmax=N_viewPairs4inference)
