"""oracle/cvc_oracle.py — TEST INFRASTRUCTURE. ctypes wrapper of oracle/cvc_oracle.c (the C restatement of
utils/CVC.py:6-104,108-111 and camera.py:123-184) + a line-by-line numpy restatement for cross-checking.
Pinned against the reference's own outputs in tests/golden/ (see oracle/gen_golden.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcvc_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libcvc_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "cvc_oracle.c")):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.sn_oracle_cvc.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def gen_coloredCubes(selected_viewPairs, xyz, resol, cameraPOs, models_img, colorize_cube_D, mean6=None):
    """C restatement of CVC.gen_coloredCubes (+ preprocess mean subtraction when mean6 is given)."""
    pairs = np.ascontiguousarray(selected_viewPairs, dtype=np.int64)
    n, n_vp = pairs.shape[:2]
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(n, 3)
    resol = np.ascontiguousarray(resol, dtype=np.float32).reshape(n)
    P = np.ascontiguousarray(cameraPOs, dtype=np.float64)
    imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in models_img]
    V = len(imgs)
    ptrs = (ctypes.c_void_p * V)(*[im.ctypes.data for im in imgs])
    H = (ctypes.c_int * V)(*[im.shape[0] for im in imgs])
    W = (ctypes.c_int * V)(*[im.shape[1] for im in imgs])
    s = int(colorize_cube_D)
    out = np.empty((n * n_vp, 6, s, s, s), dtype=np.float32)
    m = None if mean6 is None else np.ascontiguousarray(mean6, dtype=np.float32).reshape(6)
    rc = lib().sn_oracle_cvc(n, n_vp, s, V, _p(pairs), _p(xyz), _p(resol), _p(P), ptrs, H, W, _p(m), _p(out))
    if rc != 0:
        raise IndexError("view index out of range")
    return out


def perspectiveProj(projection_M, xyz_3D, return_int_hw=True):
    """camera.py:123-184 for (N_Ms,3,4) x (N_pts,3): returns (h, w), each (N_Ms, N_pts)."""
    P = np.ascontiguousarray(projection_M, dtype=np.float64).reshape(-1, 3, 4)
    pts = np.ascontiguousarray(xyz_3D, dtype=np.float64).reshape(-1, 3)
    nm, npt = P.shape[0], pts.shape[0]
    hw_f = np.empty((2, nm, npt), dtype=np.float64)
    hw_i = np.empty((2, nm, npt), dtype=np.int64)
    lib().sn_oracle_perspective_proj(nm, npt, _p(P), _p(pts), _p(hw_f), _p(hw_i))
    return (hw_i[0], hw_i[1]) if return_int_hw else (hw_f[0], hw_f[1])


def gen_coloredCubes_numpy(selected_viewPairs, xyz, resol, cameraPOs, models_img, colorize_cube_D):
    """Independent numpy restatement of utils/CVC.py:6-104 (vectorised differently from the reference)."""
    pairs = np.asarray(selected_viewPairs)
    n, n_vp = pairs.shape[:2]
    s = int(colorize_cube_D)
    out = np.zeros((n, n_vp * 2, 3, s, s, s), dtype=np.float32)
    idx = np.arange(s)
    for c in range(n):
        r = np.float32(resol[c])
        gx = idx * r + np.float32(xyz[c][0]); gy = idx * r + np.float32(xyz[c][1]); gz = idx * r + np.float32(xyz[c][2])
        X, Y, Z = np.meshgrid(gx, gy, gz, indexing="ij")
        pts = np.stack([X.ravel(), Y.ravel(), Z.ravel(), np.ones(s ** 3)])
        for k, view in enumerate(pairs[c].flatten()):
            q = np.dot(cameraPOs[view], pts)
            u = np.rint(q[0] / q[2]); v = np.rint(q[1] / q[2])
            img = models_img[view]
            ok = (u >= 0) & (u < img.shape[1]) & (v >= 0) & (v < img.shape[0])
            rgb = np.zeros((s ** 3, 3))
            rgb[ok] = img[v[ok].astype(np.int64), u[ok].astype(np.int64)]
            out[c, k] = rgb.T.reshape(3, s, s, s)
    return out.reshape(n * n_vp, 6, s, s, s)


def gen_non0Batch_npBool(boolIndicators, batch_size):
    """utils/utils.py:77-110 restated: bool selectors of consecutive groups of `batch_size` True entries."""
    ind = np.asarray(boolIndicators, dtype=bool)
    cs = np.cumsum(ind)
    n_all = int(ind.sum())
    sel = []
    for start in range(0, n_all, batch_size):
        end = min(start + batch_size, n_all)
        sel.append((cs >= start + 1) & (cs <= end) & ind)
    return np.array(sel)


def color_fuse(viewPair_coloredCubes, viewPair_surf_predictions, weight4viewPair):
    """utils/utils.py:8-42 generate_voxelLevelWeighted_coloredCubes restated op by op in float32:
    vw = w*pred; vw /= sum_p vw; mc = mean over the pair's two views; rgb = uint8(sum_p vw*mc)."""
    pred = np.asarray(viewPair_surf_predictions, dtype=np.float32)
    n, P, D = pred.shape[:3]
    w = np.asarray(weight4viewPair, dtype=np.float32)
    X = np.asarray(viewPair_coloredCubes, dtype=np.float32).reshape(n, P, 2, 3, D, D, D)
    vw = np.empty_like(pred)
    for p in range(P):
        vw[:, p] = w[:, p, None, None, None] * pred[:, p]
    tot = np.zeros((n, D, D, D), dtype=np.float32)
    for p in range(P):
        tot = tot + vw[:, p]
    out = np.zeros((n, 3, D, D, D), dtype=np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        for p in range(P):
            mc = (X[:, p, 0] + X[:, p, 1]) / np.float32(2)
            out = out + (vw[:, p] / tot)[:, None] * mc
        return np.nan_to_num(out, nan=0.0).astype(np.uint8)
