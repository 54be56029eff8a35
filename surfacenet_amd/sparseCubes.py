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


def _grow(table, rows):
    """A running per-cube table (None before the first batch) with the batch's rows appended along axis 0."""
    return rows if table is None else np.concatenate([table, rows], axis=0)


def append_dense_2sparseList(prediction_sub, rgb_sub, param_sub, viewPair_sub, min_prob=0.5, rayPool_thresh=0, enable_centerCrop=False,
                             cube_Dcenter=None, enable_rayPooling=False, cameraPOs=None, cameraTs=None, prediction_list=[],
                             rgb_list=[], vxl_ijk_list=[], rayPooling_votes_list=[], cube_ijk_np=None, param_np=None, viewPair_np=None):
    """One batch of the loop body's bookkeeping (call site main_reconstruct.py:154-162; contract of utils/sparseCubes.py:82-160): the dense
    batch - prediction_sub (N,1,D,D,D) or (N,D,D,D), rgb_sub (N,3,D,D,D) - goes through `dense2sparse` on the GPU and what survives is
    appended to the caller's running state: the four voxel lists are extended IN PLACE (the reference's callers rely on that), the three
    per-cube tables (cube ijk, cube parameters with the cropped origin, view pairs as uint16) grow by the batch's non-empty cubes and are
    returned. Returns (prediction_list, rgb_list, vxl_ijk_list, rayPooling_votes_list, cube_ijk_np, param_np, viewPair_np)."""
    pred = np.asarray(prediction_sub)
    if pred.ndim == 5:                                   # (N,1,D,D,D) float32 from the network -> the float16 (N,D,D,D) the sparse lists store
        pred = pred.astype(np.float16)[:, 0]
    colours = np.moveaxis(np.asarray(rgb_sub).astype(np.uint8), 1, -1)          # channels last, as dense2sparse indexes them
    pairs16 = np.asarray(viewPair_sub).astype(np.uint16)
    keep, ijk_b, pred_b, rgb_b, votes_b, param_cropped = dense2sparse(
        prediction=pred, rgb=colours, param=param_sub, viewPair=pairs16, min_prob=min_prob, rayPool_thresh=rayPool_thresh,
        enable_centerCrop=enable_centerCrop, cube_Dcenter=cube_Dcenter, enable_rayPooling=enable_rayPooling, cameraPOs=cameraPOs, cameraTs=cameraTs)
    for running, batch in ((prediction_list, pred_b), (rgb_list, rgb_b), (vxl_ijk_list, ijk_b), (rayPooling_votes_list, votes_b)):
        running.extend(batch)
    return (prediction_list, rgb_list, vxl_ijk_list, rayPooling_votes_list,
            _grow(cube_ijk_np, param_sub['ijk'][keep]), _grow(param_np, param_cropped[keep]), _grow(viewPair_np, pairs16[keep]))


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
