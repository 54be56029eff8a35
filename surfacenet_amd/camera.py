"""Drop-in for the projection entry points of the reference's `utils/camera.py` that the reconstruction pipeline calls
(main_reconstruct.py:62-65), executed on the MI355X through `sn_project_points`:

    perspectiveProj              utils/camera.py:123-184   -> Context.project (project_points_kernel, csrc/postpass.h)
    perspectiveProj_cubesCorner  utils/camera.py:188-245   -> the 8 corners of every cube through the same kernel

Same signatures, result shapes / dtypes and ValueError contract as the reference; the arithmetic (fp64 FMA chain, IEEE
divide, half-to-even rounding) is the CVC warp's and is checked bit for bit against vectors produced by the reference
(`tests/test_gpu_dropin.py`). No CPU projection lives here. `cameraPs2Ts` (utils/camera.py:103-120, called next to the readers at
main_reconstruct.py:51) is kept with the reference's list-in / list-out contract on top of `viewPairSelection.camera_centers`. The
camera FILE readers (`readCameraPOs_as_np`, utils/camera.py:8-81) are host I/O and stay with the caller: swap the
functions, not the module - INTEGRATION.md §1.
"""
import numpy as np

from . import runtime

_CORNER_OFFSETS = np.array([[i, j, k] for i in (0, 1) for j in (0, 1) for k in (0, 1)])   # corner c = 4i + 2j + k, as np.indices((2,2,2))


def _project(projection_M, pts, return_int_hw, return_depth):
    M = np.asarray(projection_M)
    if M.shape[-2:] != (3, 4) or M.ndim not in (2, 3):
        raise ValueError("perspectiveProj needs projection_M with shape (3,4), however got {}".format(M.shape))
    res = runtime.any_context().project(pts, projection_M=M.reshape((-1, 3, 4)), return_int_hw=return_int_hw, return_depth=return_depth)
    return res if M.ndim == 3 else tuple(r[0] for r in res)       # one matrix: (N_pts,) results, a stack: (N_Ms, N_pts)


def perspectiveProj(projection_M, xyz_3D, return_int_hw=True, return_depth=False):
    """img_h, img_w [, depth] of the points xyz_3D (3,) / (N_pts,3) under projection_M (3,4) / (N_Ms,3,4)."""
    pts = np.asarray(xyz_3D)
    if pts.ndim == 1:
        pts = pts[None, :]
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError("perspectiveProj needs xyz_3D with shape (3,) or (N_pts, 3), however got {}".format(pts.shape))
    return _project(projection_M, pts, return_int_hw, return_depth)


def perspectiveProj_cubesCorner(projection_M, cube_xyz_min, cube_D_mm, return_int_hw=True, return_depth=False):
    """img_h, img_w of the 8 corners of every cube, shape (N_Ms, N_cubes, 8) (N_Ms = 1 for a single matrix)."""
    lo = np.asarray(cube_xyz_min)
    if lo.ndim == 1:
        lo = lo[None, :]
    if lo.ndim != 2 or lo.shape[1] != 3:
        raise ValueError("perspectiveProj needs cube_xyz_min with shape (3,) or (N_pts, 3), however got {}".format(lo.shape))
    corners = lo[:, None, :] + _CORNER_OFFSETS[None] * cube_D_mm          # numpy promotion as in the reference (int64 * scalar + array)
    h, w = _project(projection_M, corners.reshape((-1, 3)), return_int_hw, False)
    return h.reshape((-1, lo.shape[0], 8)), w.reshape((-1, lo.shape[0], 8))


def cameraPs2Ts(cameraPOs):
    """Camera centres of projection matrices (utils/camera.py:103-120): a list of (3,4) matrices gives a list of (3,) centres, an
    array (V,3,4) gives an array (V,3). Host arithmetic, O(V), runs once per scene (`viewPairSelection.camera_centers`: C = -M^-1 p4,
    equal to the reference's four 3x3 determinants up to rounding)."""
    from .viewPairSelection import camera_centers
    if type(cameraPOs) is list:
        return [camera_centers(P)[0] for P in cameraPOs]
    return camera_centers(cameraPOs)
