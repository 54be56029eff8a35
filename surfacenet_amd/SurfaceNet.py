"""Drop-in for the inference entry point of the reference's `nets/SurfaceNet.py`, executed on the MI355X.

`SurfaceNet_inference(N_viewPairs4inference, model_file, layerNameList_2_load)` (nets/SurfaceNet.py:385-402)
returns `(viewPair_relativeImpt_fn, nViewPair_SurfaceNet_fn)` with the calling conventions of the two compiled
Theano functions (nets/SurfaceNet.py:337-338, 365-382):
    viewPair_relativeImpt_fn(features (n*P,258) f32 [, n_samples_perGroup=P]) -> (n, P) f32 softmax weights
    nViewPair_SurfaceNet_fn(X [, w][, n_samples_perGroup])  -> [fused (n,1,s,s,s) f32, unfused (n,N_vp,s,s,s) f32]
X is float32 (n*N_vp, 6, s,s,s), mean-subtracted; w is float32 (n, N_vp). TypeError on dtype/ndim mismatch, as Theano.
"""
import numpy as np

from . import runtime, weights
from .context import NumericsGuard


def SurfaceNet_inference(N_viewPairs4inference, model_file, layerNameList_2_load=None, cube_D=None, param_values=None, auto_calibrate=True):
    """model_file: the reference's `*.model` pickle. `param_values` (list of arrays in weight-file order) may be given
    instead, e.g. weights.synthetic_param_values(seed). cube_D = None (default): inferred from X.shape at every call of
    nViewPair_SurfaceNet_fn (the reference fixes it at compile time from params.__cube_D, params.py:65: 64, or 32), so the
    drop-in accepts whichever of the two the caller's params selects; an int pins it (any other X then raises TypeError).
    auto_calibrate (default on): after every call `nViewPair_SurfaceNet_fn` reads the library's saturation warning (context.NumericsGuard); the
    first time the loaded weights push stored activations past the range of the default mode's 6-bit code planes it derives the premultipliers
    from that batch, recomputes the batch and emits one RuntimeWarning with the layer names and the exponents chosen."""
    values = param_values if param_values is not None else weights.load_lasagne_pickle(model_file)
    runtime.set_param_values(values)
    if cube_D is not None:
        runtime.prefer_cube_D(cube_D)
    N_vp = int(N_viewPairs4inference)
    guards = {}                                  # one NumericsGuard per context (cube size) this callable has run on

    def viewPair_relativeImpt_fn(similFeature, n_samples_perGroup=N_vp):
        f = np.asarray(similFeature)
        if f.dtype != np.float32 or f.ndim != 2:
            raise TypeError("similFeature must be a float32 matrix")
        ctx = runtime.any_context() if cube_D is None else runtime.context_for(cube_D, n_samples=1)
        return ctx.relative_weights(f, int(n_samples_perGroup))

    def nViewPair_SurfaceNet_fn(X, *args, **kwargs):
        n_per = int(kwargs.pop("n_samples_perGroup", N_vp))
        if kwargs:
            raise TypeError("unexpected keyword arguments %s" % sorted(kwargs))
        if N_vp == 1:
            if len(args) > 0:
                raise TypeError("the N_viewPairs4inference == 1 function takes X only (nets/SurfaceNet.py:354-357)")
            w = None
            n_per = 1
        else:
            if len(args) < 1 or len(args) > 2:
                raise TypeError("expected (X, similWeight[, n_samples_perGroup])")
            w = args[0]
            if len(args) == 2:
                n_per = int(args[1])
        if not isinstance(X, np.ndarray) or X.dtype != np.float32 or X.ndim != 5:
            raise TypeError("X must be a float32 5-D ndarray")
        if X.shape[1] != 6 or X.shape[2] != X.shape[3] or X.shape[3] != X.shape[4]:
            raise TypeError("X must have shape (N*n_vp, 6, s, s, s), got %s" % (X.shape,))
        ctx = runtime.context_for(X.shape[2] if cube_D is None else cube_D, n_samples=X.shape[0])
        fused, unfused = ctx.forward(X, w, n_vp=n_per, return_unfused=True)
        guard = guards.get(id(ctx))
        if guard is None:
            guard = guards[id(ctx)] = NumericsGuard(ctx, enabled=auto_calibrate)
        if guard.check("nViewPair_SurfaceNet_fn") is not None:
            fused, unfused = ctx.forward(X, w, n_vp=n_per, return_unfused=True)       # premultipliers recalibrated on this batch: redo it
            ctx.numeric_status()                                                         # (clears what the calibration's own tolerance leaves)
        if N_vp == 1:
            return [fused, fused]      # both outputs are the same tensor in the reference (SurfaceNet.py:355-357)
        return [fused, unfused]

    viewPair_relativeImpt_fn.sn_gpu = True     # lets viewPairSelection.viewPairSelection skip the (N*P, 258) feature matrix
    viewPair_relativeImpt_fn.sn_cube_D = cube_D
    return viewPair_relativeImpt_fn, nViewPair_SurfaceNet_fn
