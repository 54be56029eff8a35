"""Drop-in for the reference's `utils/rayPooling.py`, executed on the MI355X.

`rayPooling_1cube_numpy(cameraPOs, cameraTs, cube_prediction, viewPair_viewIndx, xyz, resol, prediction_thresh=None)`
keeps the reference's signature (utils/rayPooling.py:143) and return value (votes, shape of the squeezed prediction).
`rayPooling_cubes` is the batched form the GPU is built for: all cubes of a batch in one call. The kernel is
surfacenet_amd/csrc/postpass.h; there is no CPU implementation here.

Predictions are taken at float16 precision: the reference's only call site (utils/sparseCubes.py:57-59) passes the
float16 array produced at sparseCubes.py:136, and the GPU entry point rounds to float16 itself. Passing a wider dtype
whose values are not float16-representable raises TypeError instead of silently changing the comparison results.
"""
import numpy as np

from . import runtime


def _as_f16(pred):
    pred = np.asarray(pred)
    p16 = pred.astype(np.float16)
    if pred.dtype != np.float16 and not np.array_equal(p16.astype(pred.dtype), pred):
        raise TypeError("cube_prediction must be float16 (or float16-representable): the GPU ray pooling compares float16 values, "
                        "as the reference's call site utils/sparseCubes.py:57-59 does")
    return p16


def rayPooling_cubes(cameraPOs, predictions, viewPairs, xyz, resol, prediction_thresh=None):
    """predictions (n,D,D,D) float16, viewPairs (n,N_vp,2), xyz (n,3), resol (n,) -> votes (n,D,D,D) uint8."""
    p16 = _as_f16(predictions)
    n, D = p16.shape[:2]
    if n == 0:
        return np.zeros(p16.shape, dtype=np.uint8)
    ctx = runtime.context_for(D)
    runtime.bind_cameras(ctx, cameraPOs)
    viewPairs = np.asarray(viewPairs)
    V = ctx.n_cameras
    if viewPairs.size and (viewPairs.max() >= V or viewPairs.astype(np.int64).min() < -V):
        raise IndexError("view index out of range for %d views" % V)
    return ctx.ray_pool(viewPairs, xyz, resol, p16, prediction_thresh)


def rayPooling_1cube_numpy(cameraPOs, cameraTs, cube_prediction, viewPair_viewIndx, xyz, resol, prediction_thresh=None):
    """cube_N_votes (D,D,D) with max = 2*N_viewPair (utils/rayPooling.py:143-260). cameraTs is unused, as there."""
    pred = np.asarray(cube_prediction).squeeze()
    if pred.ndim != 3:
        raise ValueError('rayPooling method argument cube_prediction has {} dims'.format(pred.ndim))
    votes = rayPooling_cubes(cameraPOs, pred[None], np.asarray(viewPair_viewIndx)[None], np.asarray(xyz, dtype=np.float32)[None],
                             np.asarray(resol, dtype=np.float32).reshape(1), prediction_thresh)
    return votes[0].astype(np.int64)
