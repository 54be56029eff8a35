"""The cube-batch loop of the reference's `main_reconstruct.py:126-166`, on the MI355X, plus its multi-GPU form.

* `gen_non0Batch_npBool`   the reference's batch partition (utils/utils.py:77-110), same selectors
* `hot_loop`               generator with the loop body's contract: per batch selector it yields
                           (surfacePrediction, unfused_predictions, CVC + mean) exactly as lines 134-150 produce them
* `infer_cubes`            the same work for a plain list of cubes, fused CVC->CNN->fusion per batch
* `shard_bounds`, `infer_cubes_sharded`  cubes are independent: contiguous ranges per rank, no data-path
                           collective; ONE all-gather of the fused probabilities at the end (RCCL over xGMI when the
                           process group is "nccl"; gloo on CPU in the tests)
"""
import numpy as np

from .context import MEAN_CVC_RGBRGB


def gen_non0Batch_npBool(boolIndicators, batch_size):
    """Bool selectors (N_all,) picking consecutive groups of `batch_size` True entries (utils/utils.py:77-110)."""
    ind = np.asarray(boolIndicators).astype(bool)
    csum = np.cumsum(ind)
    n_true = int(ind.sum())
    out = []
    for start in range(0, n_true, int(batch_size)):
        end = min(start + int(batch_size), n_true)
        out.append((csum >= start + 1) & (csum <= end) & ind)
    return np.array(out)


def hot_loop(ctx, validCubes, viewPairs4Reconstr, w_viewPairs4Reconstr, cubes_param_np, batch_size, return_cvc=True):
    """main_reconstruct.py:132-150 for every batch: yields (_batch, surfacePrediction (n,1,s,s,s), unfused (n,N_vp,s,s,s),
    _CVCs2_sub + mean (n*N_vp,6,s,s,s) raw colours or None). `cubes_param_np` is the reference's structured array
    ('xyz' f32x3, 'resol' f32, ...; utils/scene.py:7-61)."""
    validCubes = np.asarray(validCubes).astype(bool)
    n_vp = viewPairs4Reconstr.shape[1]
    mean = MEAN_CVC_RGBRGB[None, :, None, None, None]
    for _batch in gen_non0Batch_npBool(validCubes, batch_size):
        sel = _batch[validCubes]
        w = None if n_vp == 1 else np.ascontiguousarray(w_viewPairs4Reconstr[sel], dtype=np.float32)
        fused, unfused, cvc = ctx.cvc_forward(viewPairs4Reconstr[sel], cubes_param_np["xyz"][_batch], cubes_param_np["resol"][_batch],
                                              w, return_unfused=True, return_cvc=return_cvc)
        if cvc is not None:
            cvc += mean          # main_reconstruct.py:150
        yield _batch, fused, (fused if n_vp == 1 else unfused), cvc


def infer_cubes(ctx, viewPairs, xyz, resol, w=None, batch_size=None):
    """Fused surface probabilities (n,1,s,s,s) for n cubes, processed in batches of `batch_size` cubes."""
    viewPairs = np.asarray(viewPairs)
    n, n_vp = viewPairs.shape[:2]
    s = ctx.cube_D
    batch_size = int(batch_size or max(1, ctx.max_samples // n_vp))
    out = np.empty((n, 1, s, s, s), dtype=np.float32)
    for i0 in range(0, n, batch_size):
        i1 = min(n, i0 + batch_size)
        f, _, _ = ctx.cvc_forward(viewPairs[i0:i1], xyz[i0:i1], resol[i0:i1], None if w is None else w[i0:i1], return_unfused=False)
        out[i0:i1] = f
    return out


def shard_bounds(n, world, rank):
    """Contiguous range [lo, hi) of rank's cubes: ceil(n/world) per rank, the tail ranks may be short or empty."""
    per = -(-int(n) // int(world))
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def infer_cubes_sharded(compute_fn, n, s, group=None, device=None):
    """Every rank computes its shard with compute_fn(lo, hi) -> (hi-lo, 1, s,s,s) float32 ndarray, then ONE all-gather
    (padded to equal shard length) returns the full (n,1,s,s,s) array on every rank."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = -(-int(n) // world)
    lo, hi = shard_bounds(n, world, rank)
    local = torch.zeros((per, 1, s, s, s), dtype=torch.float32, device=device)
    if hi > lo:
        res = compute_fn(lo, hi)
        local[: hi - lo] = torch.from_numpy(np.ascontiguousarray(res, dtype=np.float32)).to(local.device) if isinstance(res, np.ndarray) else res
    full = torch.empty((world * per, 1, s, s, s), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(full, local, group=group)
    return full[:n].cpu().numpy()
