"""Drop-in for the reference's `utils/CVC.py` hot-path functions, executed on the MI355X.

Same names, keyword arguments, shapes, dtypes and error behaviour as
  CVC.gen_coloredCubes          utils/CVC.py:56-104  (call site main_reconstruct.py:134-141)
  CVC.preprocess_augmentation   utils/CVC.py:108-122 (call site main_reconstruct.py:143)
so `import surfacenet_amd.CVC as CVC` keeps main_reconstruct.py's loop body unchanged.
The warp itself is surfacenet_amd/csrc/cvc_warp.h (HIP); there is no CPU implementation here.
"""
import numpy as np

from . import runtime


def gen_coloredCubes(selected_viewPairs, xyz, resol, cameraPOs, models_img, colorize_cube_D, visualization_ON=False,
                     occupiedCubes_01=None):
    """
    inputs:
    selected_viewPairs: (N_cubes, N_select_viewPairs, 2) view indices into cameraPOs / models_img
    xyz, resol: (N_cubes, 3) float32 min corners, (N_cubes,) float32 voxel sizes
    return:
    coloredCubes = (N_cubes*N_select_viewPairs, 3*2) + (colorize_cube_D,)*3 float32, raw 0..255
    """
    if visualization_ON:
        raise NotImplementedError("visualization_ON is a debugging aid of the reference (utils/CVC.py:49-50) and is not supported")
    selected_viewPairs = np.asarray(selected_viewPairs)
    if selected_viewPairs.shape[0] == 0:
        return np.zeros((0, 6) + (colorize_cube_D,) * 3, dtype=np.float32)
    ctx = runtime.context_for(colorize_cube_D, n_samples=selected_viewPairs.shape[0] * selected_viewPairs.shape[1])
    runtime.bind_scene(ctx, cameraPOs, models_img)
    V = ctx.n_views
    if selected_viewPairs.size and (selected_viewPairs.max() >= V or selected_viewPairs.min() < -V):
        raise IndexError("view index out of range for %d views" % V)  # numpy raises IndexError in the reference
    return ctx.cvc(selected_viewPairs, xyz, resol, mean=None)


def preprocess_augmentation(gt_sub, X_sub, mean_rgb, augment_ON=True, crop_ON=True):
    """Inference form only (augment_ON=False, crop_ON=False): returns (gt_sub, X_sub.astype(float32) - mean_rgb).
    The result is a fresh writable ndarray: the caller later does `X += mean` in place (main_reconstruct.py:150).
    (The fused entry point `runtime.Context.cvc_forward` performs this subtraction inside the warp kernel; this
    function exists so the reference's three-call protocol keeps working.)"""
    if augment_ON or crop_ON:
        raise NotImplementedError("training-time augmentation/cropping: the helpers the reference names "
                                  "(data_augment_rand_rotate, data_augment_crop) are not defined in the reference either")
    X_sub = np.asarray(X_sub).astype(np.float32)
    X_sub -= mean_rgb
    return gt_sub, X_sub
