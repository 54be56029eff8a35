"""Drop-in for the dense->sparse step of the reference's `utils/sparseCubes.py` on the MI355X.

    dense2sparse              utils/sparseCubes.py:9-77
    append_dense_2sparseList  utils/sparseCubes.py:82-160   (call site main_reconstruct.py:153-160)
    filter_voxels             utils/sparseCubes.py:205-243  (host-side list thresholding, unchanged semantics)

Same names, keyword arguments, output lists and dtypes. The ray pooling, thresholding, centre crop and compaction of a
whole batch run in one GPU call (surfacenet_amd/csrc/postpass.h); only the packed voxel lists cross PCIe. `param` is
the reference's structured array ('xyz' f32x3, 'ijk' u32x3, 'resol' f32; utils/scene.py:55).
"""
import numpy as np

from . import runtime


def dense2sparse(prediction, rgb, param, viewPair, min_prob=0.5, rayPool_thresh=0, enable_centerCrop=False, cube_Dcenter=None,
                 enable_rayPooling=False, cameraPOs=None, cameraTs=None):
    """prediction float16 (N,D,D,D), rgb uint8 (N,D,D,D,3), viewPair (N,N_vp,2) ->
    (nonempty_cube_indx, vxl_ijk_list, prediction_list, rgb_list, rayPooling_votes_list, param_new)."""
    prediction = np.asarray(prediction)
    N, D = prediction.shape[:2]
    param_new = np.copy(param)
    if enable_centerCrop:
        _Cmin = (D - cube_Dcenter) // 2
        param_new['xyz'] += param_new['resol'][:, None] * _Cmin
    if N == 0:
        return [], [], [], [], [], param_new
    ctx = runtime.context_for(D)
    if enable_rayPooling:
        runtime.bind_cameras(ctx, cameraPOs)
        vp = np.asarray(viewPair)
        if vp.size and (vp.max() >= ctx.n_cameras):
            raise IndexError("view index out of range for %d views" % ctx.n_cameras)
    else:
        viewPair = np.zeros((N, 1, 2), dtype=np.int64) if viewPair is None else viewPair
    rgb_planar = np.ascontiguousarray(np.transpose(np.asarray(rgb, dtype=np.uint8), (0, 4, 1, 2, 3)))      # (N,3,D,D,D)
    offsets, ijk, p16, rgb_out, votes = ctx.dense2sparse(prediction, rgb_planar, viewPair, param['xyz'], param['resol'], min_prob=min_prob,
                                                         rayPool_thresh=rayPool_thresh, enable_centerCrop=enable_centerCrop,
                                                         cube_Dcenter=cube_Dcenter, enable_rayPooling=enable_rayPooling)
    nonempty = [int(i) for i in np.nonzero(np.diff(offsets))[0]]
    cut = lambda a: [a[offsets[i]:offsets[i + 1]].copy() for i in nonempty]
    return nonempty, cut(ijk), cut(p16), cut(rgb_out), (cut(votes) if enable_rayPooling else []), param_new


def append_dense_2sparseList(prediction_sub, rgb_sub, param_sub, viewPair_sub, min_prob=0.5, rayPool_thresh=0, enable_centerCrop=False,
                             cube_Dcenter=None, enable_rayPooling=False, cameraPOs=None, cameraTs=None, prediction_list=[],
                             rgb_list=[], vxl_ijk_list=[], rayPooling_votes_list=[], cube_ijk_np=None, param_np=None, viewPair_np=None):
    """utils/sparseCubes.py:82-160: casts, dense2sparse, then appends to the running lists / arrays (lists are extended in
    place, as in the reference). prediction_sub (N,1,D,D,D)/(N,D,D,D), rgb_sub (N,3,D,D,D)."""
    prediction_sub = np.asarray(prediction_sub)
    if prediction_sub.ndim == 5:
        prediction_sub = prediction_sub.astype(np.float16)[:, 0]
    rgb_sub = np.transpose(np.asarray(rgb_sub).astype(np.uint8), axes=(0, 2, 3, 4, 1))
    cube_ijk_sub = param_sub['ijk']
    viewPair_sub = np.asarray(viewPair_sub).astype(np.uint16)
    nonempty, vxl_ijk_sub_list, prediction_sub_list, rgb_sub_list, votes_sub_list, param_new_sub = dense2sparse(
        prediction=prediction_sub, rgb=rgb_sub, param=param_sub, viewPair=viewPair_sub, min_prob=min_prob, rayPool_thresh=rayPool_thresh,
        enable_centerCrop=enable_centerCrop, cube_Dcenter=cube_Dcenter, enable_rayPooling=enable_rayPooling, cameraPOs=cameraPOs,
        cameraTs=cameraTs)
    param_sub = param_new_sub[nonempty]
    viewPair_sub = viewPair_sub[nonempty]
    cube_ijk_sub = cube_ijk_sub[nonempty]
    prediction_list.extend(prediction_sub_list)
    rgb_list.extend(rgb_sub_list)
    vxl_ijk_list.extend(vxl_ijk_sub_list)
    rayPooling_votes_list.extend(votes_sub_list)
    param_np = param_sub if param_np is None else np.concatenate([param_np, param_sub], axis=0)
    viewPair_np = viewPair_sub if viewPair_np is None else np.vstack([viewPair_np, viewPair_sub])
    cube_ijk_np = cube_ijk_sub if cube_ijk_np is None else np.vstack([cube_ijk_np, cube_ijk_sub])
    return prediction_list, rgb_list, vxl_ijk_list, rayPooling_votes_list, cube_ijk_np, param_np, viewPair_np


def filter_voxels(vxl_mask_list=[], prediction_list=None, prob_thresh=None, rayPooling_votes_list=None, rayPool_thresh=None):
    """Per-cube boolean masks: prediction >= prob_thresh (scalar or per-cube list) AND votes >= rayPool_thresh, ANDed into
    `vxl_mask_list` when it is non-empty (utils/sparseCubes.py:205-243; call site main_reconstruct.py:172-173)."""
    def merge(masks):
        if len(vxl_mask_list) == 0:
            vxl_mask_list.extend(masks)
        else:
            for i, m in enumerate(masks):
                vxl_mask_list[i] &= m

    if prediction_list is not None:
        if prob_thresh is None:
            raise Warning('prob_thresh should not be None.')
        per_cube = isinstance(prob_thresh, list)
        merge([p >= (prob_thresh[i] if per_cube else prob_thresh) for i, p in enumerate(prediction_list)])
    if rayPooling_votes_list is not None:
        if rayPool_thresh is None:
            raise Warning('rayPool_thresh should not be None.')
        merge([v >= rayPool_thresh for v in rayPooling_votes_list])
    return vxl_mask_list
