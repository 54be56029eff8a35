"""oracle/post_oracle.py — TEST INFRASTRUCTURE (checker only; never imported by the product).

CPU restatement of the reference's per-cube post-pass, written independently of the reference's
dense-NDC-array formulation (this file sorts; the reference scatters into a zero array and argmaxes):

    ray_pool_1cube    utils/rayPooling.py:143-260  rayPooling_1cube_numpy
    dense2sparse      utils/sparseCubes.py:9-77    dense2sparse
    to_sparse_inputs  utils/sparseCubes.py:134-138 the casts append_dense_2sparseList applies first

Parity pin: tests/golden/post_cases.npz holds outputs of the reference's own functions executed in the build
container (oracle/gen_golden_post.py); tests/test_oracle_post.py checks this file against them.

Semantics restated (what the scatter/argmax of rayPooling.py:237-256 amounts to), per DISTINCT view of the cube:
  * voxels with fp16(pred) > fp16(thresh) are projected: X = double(i)*double(resol_f32) + double(x0_f32) (:219),
    q = P @ [X,Y,Z,1] with the K=4 FMA chain of the BLAS dgemm (camera.py:174), w,h = rint(q0/q2), rint(q1/q2)
    (camera.py:177-179), depth bin d = rint(q2 / double(resol_f32)) (:228-229);
  * a CELL is a (w,h,d) triple. Fancy assignment with repeated indices keeps the LAST write (:248): the cell holds
    the voxel with the LARGEST flat index;
  * per PIXEL (w,h), argmax over the depth axis of the cells' stored predictions (:250): the largest stored value,
    ties -> the smallest d. Empty cells hold 0, so when every stored value of a pixel is 0 the argmax is column 0
    (d = the view's minimum bin) whose stored voxel index is 0 unless a voxel was written there;
  * that cell's voxel gets this view's vote (:252-253); votes of a view repeated in the pair list count once per
    occurrence (:255).
Predictions must be >= 0 (they are sigmoid outputs); negative values would make empty cells win the argmax.
"""
import ctypes

import numpy as np

from . import cvc_oracle


def _numerators(P, X, Y, Z):
    """(3, n) float64 rows q0,q1,q2 = P @ [X,Y,Z,1] with the dgemm FMA chain (cvc_oracle.c dot4)."""
    n = X.size
    pts = np.ascontiguousarray(np.stack([X, Y, Z]), dtype=np.float64)
    out = np.empty((3, n), dtype=np.float64)
    Pc = np.ascontiguousarray(P, dtype=np.float64)
    cvc_oracle.lib().sn_oracle_dot34(ctypes.c_int(n), Pc.ctypes.data_as(ctypes.c_void_p), pts.ctypes.data_as(ctypes.c_void_p),
                                     out.ctypes.data_as(ctypes.c_void_p))
    return out


def ray_pool_1cube(cameraPOs, cube_prediction, viewPair_viewIndx, xyz, resol, prediction_thresh=None):
    """votes (D,D,D) int64, max = 2*N_viewPair. cube_prediction is used in ITS dtype (the caller passes float16)."""
    pred = np.squeeze(np.asarray(cube_prediction))
    if pred.ndim != 3:
        raise ValueError("cube_prediction must squeeze to 3 dims")
    shape = pred.shape
    flat = pred.reshape(-1)
    if flat.size and float(flat.min()) < 0:
        raise ValueError("predictions must be >= 0")
    views, inverse = np.unique(np.asarray(viewPair_viewIndx).reshape(-1), return_inverse=True)
    if prediction_thresh is None:
        sel = np.arange(flat.size)
    else:
        sel = np.nonzero(flat > flat.dtype.type(prediction_thresh))[0]
    votes_view = np.zeros((views.size, flat.size), dtype=bool)
    if sel.size:
        i, j, k = np.unravel_index(sel, shape)
        r = float(np.float32(resol))
        x0, y0, z0 = (float(np.float32(v)) for v in xyz)
        X, Y, Z = i.astype(np.float64) * r + x0, j.astype(np.float64) * r + y0, k.astype(np.float64) * r + z0
        p64 = flat[sel].astype(np.float64)
        for vi, view in enumerate(views):
            q = _numerators(np.asarray(cameraPOs)[view], X, Y, Z)
            w = np.rint(q[0] / q[2]).astype(np.int64)
            h = np.rint(q[1] / q[2]).astype(np.int64)
            d = np.rint(q[2] / r).astype(np.int32).astype(np.int64)
            d -= d.min()
            # cells: sort by (w,h,d,flat index); the last entry of each (w,h,d) run is the cell's voxel
            o = np.lexsort((sel, d, h, w))
            ws, hs, ds = w[o], h[o], d[o]
            last = np.ones(o.size, dtype=bool)
            last[:-1] = (ws[1:] != ws[:-1]) | (hs[1:] != hs[:-1]) | (ds[1:] != ds[:-1])
            c = o[last]                                             # one entry per cell
            cw, ch, cd, cp, cidx = w[c], h[c], d[c], p64[c], sel[c]
            # pixels: per (w,h) the cell with the largest stored value, ties -> smallest d
            o2 = np.lexsort((cd, -cp, ch, cw))
            first = np.ones(o2.size, dtype=bool)
            first[1:] = (cw[o2][1:] != cw[o2][:-1]) | (ch[o2][1:] != ch[o2][:-1])
            win = o2[first]
            voted = cidx[win].copy()
            zero = cp[win] == 0                                      # all stored values 0: argmax -> column 0
            if zero.any():
                # cells are sorted (.., -p, d): with all p == 0 the first cell of the pixel has its smallest d
                voted[zero] = np.where(cd[win][zero] == 0, cidx[win][zero], 0)
            votes_view[vi, voted] = True
    return votes_view[inverse].sum(axis=0).reshape(shape)


def to_sparse_inputs(prediction_sub, rgb_sub):
    """The casts of append_dense_2sparseList (sparseCubes.py:134-138): (N,1,D,D,D) f32 -> (N,D,D,D) f16 and
    (N,3,D,D,D) float -> (N,D,D,D,3) uint8."""
    p = np.asarray(prediction_sub)
    if p.ndim == 5:
        p = p.astype(np.float16)[:, 0]
    rgb = np.transpose(np.asarray(rgb_sub).astype(np.uint8), axes=(0, 2, 3, 4, 1))
    return p, rgb


def dense2sparse(prediction, rgb, xyz, resol, viewPair, min_prob=0.5, rayPool_thresh=0, enable_centerCrop=False,
                 cube_Dcenter=None, enable_rayPooling=False, cameraPOs=None):
    """prediction (N,D,D,D) float16, rgb (N,D,D,D,3) uint8, xyz (N,3) f32, resol (N,) f32, viewPair (N,N_vp,2).
    Returns (nonempty_cube_indx, vxl_ijk_list, prediction_list, rgb_list, rayPooling_votes_list, xyz_new)."""
    N, D = prediction.shape[:2]
    xyz_new = np.array(xyz, dtype=np.float32, copy=True)
    lo, hi = 0, D
    if enable_centerCrop:
        lo = (D - int(cube_Dcenter)) // 2
        hi = lo + int(cube_Dcenter)
        xyz_new += (np.asarray(resol, dtype=np.float32)[:, None] * lo).astype(np.float32)
    nonempty, ijk_l, p_l, rgb_l, v_l = [], [], [], [], []
    for n in range(N):
        pc = prediction[n][lo:hi, lo:hi, lo:hi]
        votes = None
        if enable_rayPooling:
            votes = ray_pool_1cube(cameraPOs, prediction[n], viewPair[n], xyz[n], resol[n], min_prob).astype(np.uint8)
            keep = votes[lo:hi, lo:hi, lo:hi] >= rayPool_thresh
        if (not enable_rayPooling) or rayPool_thresh == 0:
            keep = pc > pc.dtype.type(min_prob)
        idx = np.nonzero(keep)
        if idx[0].size == 0:
            continue
        nonempty.append(n)
        ijk_l.append(np.stack(idx, axis=1).astype(np.uint8))
        p_l.append(pc[idx].astype(np.float16))
        rgb_l.append(rgb[n][lo:hi, lo:hi, lo:hi][idx].astype(np.uint8))
        if enable_rayPooling:
            v_l.append(votes[lo:hi, lo:hi, lo:hi][idx].astype(np.uint8))
    return nonempty, ijk_l, p_l, rgb_l, v_l, xyz_new
