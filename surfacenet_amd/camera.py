"""Host-side projection helpers of the reference's `utils/camera.py` that feed early rejection:
O(N_views * N_cubes * 8) float64 arithmetic, kept in numpy with the reference's exact operation order.

    cameraPs2Ts                  utils/camera.py:87-120   (camera centres by cofactors, as the reference computes them)
    perspectiveProj              utils/camera.py:123-184
    perspectiveProj_cubesCorner  utils/camera.py:188-245
(`viewPairAngles_wrt_pts` lives in surfacenet_amd/viewPairSelection.py.)
"""
import numpy as np


def perspectiveProj(projection_M, xyz_3D, return_int_hw=True, return_depth=False):
    """projection_M (3,4)/(N_Ms,3,4), xyz_3D (3,)/(N_pts,3) -> img_h, img_w [, depth]: (N_pts,) / (N_Ms, N_pts)."""
    projection_M = np.asarray(projection_M)
    xyz_3D = np.asarray(xyz_3D)
    if projection_M.shape[-2:] != (3, 4):
        raise ValueError("perspectiveProj needs projection_M with shape (3,4), however got {}".format(projection_M.shape))
    if xyz_3D.ndim == 1:
        xyz_3D = xyz_3D[None, :]
    if xyz_3D.ndim != 2 or xyz_3D.shape[1] != 3:
        raise ValueError("perspectiveProj needs xyz_3D with shape (3,) or (N_pts, 3), however got {}".format(xyz_3D.shape))
    xyz1 = np.c_[xyz_3D, np.ones((xyz_3D.shape[0], 1))].astype(np.float64)
    pts_3D = np.matmul(projection_M, xyz1.T)
    pts_2D = pts_3D[..., :2, :]
    pts_2D /= pts_3D[..., 2:3, :]
    if return_int_hw:
        pts_2D = pts_2D.round().astype(np.int64)
    img_w, img_h = pts_2D[..., 0, :], pts_2D[..., 1, :]
    if return_depth:
        return img_h, img_w, pts_3D[..., 2, :]
    return img_h, img_w


def perspectiveProj_cubesCorner(projection_M, cube_xyz_min, cube_D_mm, return_int_hw=True, return_depth=False):
    """Projections of the 8 corners of every cube: img_h, img_w of shape (N_Ms, N_cubes, 8)."""
    cube_xyz_min = np.asarray(cube_xyz_min)
    if cube_xyz_min.ndim == 1:
        cube_xyz_min = cube_xyz_min[None, :]
    if cube_xyz_min.ndim != 2 or cube_xyz_min.shape[1] != 3:
        raise ValueError("perspectiveProj needs cube_xyz_min with shape (3,) or (N_pts, 3), however got {}".format(cube_xyz_min.shape))
    N_pts = cube_xyz_min.shape[0]
    shift = np.indices((2, 2, 2)).reshape((3, -1)).T[None, :, :] * cube_D_mm
    corners = cube_xyz_min[:, None, :] + shift
    img_h, img_w = perspectiveProj(projection_M, corners.reshape((N_pts * 8, 3)), return_int_hw=return_int_hw, return_depth=False)
    return img_h.reshape((-1, N_pts, 8)), img_w.reshape((-1, N_pts, 8))


def cameraPs2Ts(cameraPOs):
    """Camera centres (N,3) (or a list, if a list is given) of projection matrices (3,4): the null vector of P by
    cofactor expansion, C_i = (-1)^i det(P without column i), de-homogenised (utils/camera.py:87-120)."""
    def center(P):
        P = np.asarray(P)
        cof = np.array([(-1) ** i * np.linalg.det(P[:, [j for j in range(4) if j != i]]) for i in range(4)])
        return cof[:3] / cof[3]
    Ts = [center(P) for P in cameraPOs]
    return Ts if type(cameraPOs) is list else np.stack(Ts)
