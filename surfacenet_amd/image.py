"""Drop-in for the patch helpers of the reference's `utils/image.py` used by early rejection.

    preprocess_patches               utils/image.py:9-36     host numpy (a transpose, a flip, a subtraction)
    cropImgPatches                   utils/image.py:92-183   GPU gather (surfacenet_amd/csrc/simil.h patch_crop_kernel)
    img_hw_cubesCorner_inScopeCheck  utils/image.py:186-205  host numpy

`cropImgPatches` supports the one form the pipeline uses (utils/earlyRejection.py:50): pyramidRate = 1, i.e. ONE pyramid
level at zoom 1.0 — scipy's spline zoom at rate 1.0 returns the uint8 image unchanged (checked when the golden vectors
were generated) — so a patch is the 64x64 window of clamped pixels whose top-left corner is int(centre) - 32.
"""
import numpy as np

from . import runtime


def preprocess_patches(patches, mean_BGR):
    """(...,h,w,c) RGB -> (...,c,h,w) BGR minus mean_BGR (utils/image.py:9-36)."""
    patches = np.moveaxis(patches, -1, -3)
    patches = patches[..., ::-1, :, :]
    patches -= np.asarray(mean_BGR)[:, None, None]
    return patches


def cropImgPatches(img, range_h, range_w, patchSize=64, pyramidRate=1.2, interp_order=2, cubeCenter_hw=None):
    """patches (N_patches, 64, 64, 3) uint8 of `img` (h,w,3 uint8). range_h / range_w: (N_patches, 2) [min, max]."""
    if pyramidRate != 1:
        raise NotImplementedError("only pyramidRate = 1 (one pyramid level at zoom 1.0), the form used at utils/earlyRejection.py:50")
    if patchSize != 64:
        raise NotImplementedError("patchSize must be 64 (params.py:92)")
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        raise TypeError("img must be (h, w, 3) uint8")
    if cubeCenter_hw is None:
        center_h, center_w = np.mean(range_h, axis=1), np.mean(range_w, axis=1)
    else:
        center_h, center_w = cubeCenter_hw
    ctx = runtime.any_context()
    view = runtime.bind_single_image(ctx, img)
    return ctx.crop_patches(view, center_h, center_w)


def img_hw_cubesCorner_inScopeCheck(hw_shape, img_h_cubesCorner, img_w_cubesCorner):
    """(N_cubes,) bool: all 8 projected corners inside [0, img_h] x [0, img_w] (utils/image.py:186-205)."""
    img_h, img_w = hw_shape
    return (np.min(img_h_cubesCorner, axis=1) >= 0) & (np.max(img_h_cubesCorner, axis=1) <= img_h) & \
        (np.min(img_w_cubesCorner, axis=1) >= 0) & (np.max(img_w_cubesCorner, axis=1) <= img_w)
