"""The cube-batch loop of the reference's `main_reconstruct.py:126-166`, on the MI355X, plus its multi-GPU form.

* `gen_non0Batch_npBool`   the reference's batch partition (utils/utils.py:77-110), same selectors
* `hot_loop`               generator with the loop body's contract: per batch selector it yields
                           (surfacePrediction, unfused_predictions, CVC + mean) exactly as lines 134-150 produce them
* `infer_cubes`            the same work for a plain list of cubes, fused CVC->CNN->fusion per batch
* `SparseLoop`             the WHOLE loop body (main_reconstruct.py:134-160: CVC -> CNN -> fusion -> voxel colours ->
                           ray pooling -> dense2sparse) device-resident: only cube parameters go up and only the packed
                           sparse voxel lists come down
* `reconstruct_scene`       main_reconstruct.reconstruction() from the early rejection to the sparse voxel lists
                           (main_reconstruct.py:67-173) for in-memory images / cameras / cubes: every stage on the GPU
* `gather_sparse_sharded`   the multi-GPU exchange when the post-pass runs on the GPU: every rank holds only the packed
                           sparse voxel lists of its cube shard (9 B per kept voxel instead of 4 B x s^3 per cube); two
                           all-gathers (lengths, then one padded byte buffer) rebuild the global lists on every rank
* `reconstruct_scene_sharded`  the same for a cube list sharded over the ranks of a process group: every stage (early rejection,
                           view-pair selection, the cube loop) is per-cube, so each rank runs `reconstruct_scene` on its contiguous
                           cube range and ONE exchange of the packed per-rank results rebuilds the scene's lists on every rank
* `shard_bounds`, `infer_cubes_sharded`  cubes are independent: contiguous ranges per rank, no data-path
                           collective; ONE all-gather of the fused probabilities at the end (RCCL over xGMI when the
                           process group is "nccl"; gloo on CPU in the tests)
"""
import numpy as np

from .context import MEAN_CVC_RGBRGB, NumericsGuard


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


def hot_loop(ctx, validCubes, viewPairs4Reconstr, w_viewPairs4Reconstr, cubes_param_np, batch_size, return_cvc=True, auto_calibrate=True):
    """main_reconstruct.py:132-150 for every batch: yields (_batch, surfacePrediction (n,1,s,s,s), unfused (n,N_vp,s,s,s),
    _CVCs2_sub + mean (n*N_vp,6,s,s,s) raw colours or None). `cubes_param_np` is the reference's structured array
    ('xyz' f32x3, 'resol' f32, ...; utils/scene.py:7-61).
    auto_calibrate (default on): the saturation warning of the default mode's 6-bit code planes is read after every batch; the first report
    recalibrates the premultipliers on that batch, which is recomputed, with one RuntimeWarning (context.NumericsGuard)."""
    validCubes = np.asarray(validCubes).astype(bool)
    n_vp = viewPairs4Reconstr.shape[1]
    mean = MEAN_CVC_RGBRGB[None, :, None, None, None]
    guard = NumericsGuard(ctx, enabled=auto_calibrate)
    for _batch in gen_non0Batch_npBool(validCubes, batch_size):
        sel = _batch[validCubes]
        w = None if n_vp == 1 else np.ascontiguousarray(w_viewPairs4Reconstr[sel], dtype=np.float32)
        args = (viewPairs4Reconstr[sel], cubes_param_np["xyz"][_batch], cubes_param_np["resol"][_batch], w)
        fused, unfused, cvc = ctx.cvc_forward(*args, return_unfused=True, return_cvc=return_cvc)
        if guard.check("hot_loop") is not None:
            fused, unfused, cvc = ctx.cvc_forward(*args, return_unfused=True, return_cvc=return_cvc)      # recalibrated on this batch: redo it
            ctx.numeric_status()
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


class SparseLoop(object):
    """Device-resident form of one loop iteration of main_reconstruct.py:132-160 for up to `max_cubes` cubes per call.

    run(viewPairs (n,N_vp,2), xyz (n,3), resol (n,), w (n,N_vp)) returns what `sparseCubes.dense2sparse` returns for
    the batch: (nonempty_cube_indx, vxl_ijk_list, prediction_list, rgb_list, rayPooling_votes_list, xyz_new) with the
    keyword settings given at construction (defaults = the reference's call: min_prob params.__min_prob, rayPool_thresh 0,
    centre crop on, ray pooling on). auto_calibrate (default on): the first batch whose stored activations exceed the range of the default mode's 6-bit
    code planes recalibrates their premultipliers and is recomputed, with one RuntimeWarning (context.NumericsGuard)."""

    def __init__(self, ctx, n_vp, max_cubes=None, min_prob=0.5, rayPool_thresh=0, enable_centerCrop=True, cube_Dcenter=None,
                 enable_rayPooling=True, mean=MEAN_CVC_RGBRGB, auto_calibrate=True):
        self.ctx, self.n_vp = ctx, int(n_vp)
        self.guard = NumericsGuard(ctx, enabled=auto_calibrate)     # saturation of the 6-bit code planes: checked on the first batch, then once per call
        s = ctx.cube_D
        self.max_cubes = int(max_cubes or max(1, ctx.max_samples // self.n_vp))
        if self.max_cubes * self.n_vp > ctx.max_samples:
            raise ValueError("max_cubes * n_vp exceeds the context's max_samples")
        self.cfg = dict(min_prob=min_prob, rayPool_thresh=rayPool_thresh, enable_centerCrop=enable_centerCrop,
                        cube_Dcenter=(cube_Dcenter if enable_centerCrop else None), enable_rayPooling=enable_rayPooling)
        self.dc = int(cube_Dcenter) if enable_centerCrop else s
        self.lo = (s - self.dc) // 2
        self.mean = np.ascontiguousarray(mean, dtype=np.float32)
        N, S, v = self.max_cubes, self.max_cubes * self.n_vp, s ** 3
        cap = N * self.dc ** 3
        A = ctx.dev_alloc
        self.d = dict(pairs=A(S * 16), xyz=A(N * 12), resol=A(N * 4), w=A(S * 4), fused=A(N * v * 4), unfused=A(S * v * 4),
                      cvc=A(S * 6 * v * 4), rgb=A(N * 3 * v), votes=A(N * v), offsets=A((N + 1) * 8), ijk=A(cap * 3), p16=A(cap * 2),
                      rgb_out=A(cap * 3), votes_out=A(cap))
        self._off = np.zeros((N + 1,), dtype=np.int64)

    def close(self):
        for p in self.d.values():
            self.ctx.dev_free(p)
        self.d = {}

    def run(self, viewPairs, xyz, resol, w=None):
        ctx, d, n_vp = self.ctx, self.d, self.n_vp
        pairs = np.ascontiguousarray(viewPairs, dtype=np.int64)
        n = pairs.shape[0]
        if pairs.shape[1:] != (n_vp, 2) or n > self.max_cubes:
            raise ValueError("viewPairs must have shape (n <= %d, %d, 2)" % (self.max_cubes, n_vp))
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(n, 3)
        resol = np.ascontiguousarray(resol, dtype=np.float32).reshape(n)
        xyz_new = xyz + (resol[:, None] * self.lo).astype(np.float32) if self.cfg["enable_centerCrop"] else xyz.copy()
        if n == 0:
            return [], [], [], [], [], xyz_new
        V = ctx.n_views
        if pairs.max() >= V or pairs.min() < -V:
            raise IndexError("view index out of range for %d views" % V)
        pairs = np.where(pairs < 0, pairs + V, pairs)
        w = np.full((n, n_vp), 1.0 / n_vp, np.float32) if w is None else np.ascontiguousarray(w, dtype=np.float32).reshape(n, n_vp)
        ctx.h2d(d["pairs"], pairs); ctx.h2d(d["xyz"], xyz); ctx.h2d(d["resol"], resol); ctx.h2d(d["w"], w)
        ctx.cvc_forward_dev(n, n_vp, d["pairs"], d["xyz"], d["resol"], d["w"], d["fused"], d["unfused"], d["cvc"], mean=self.mean)
        ctx.color_fuse_dev(n, n_vp, d["cvc"], d["unfused"], d["w"], d["rgb"], mean=self.mean)      # main_reconstruct.py:150-152
        ctx.dense2sparse_dev(n, n_vp, d["pairs"], d["xyz"], d["resol"], d["fused"], d["rgb"], d["votes"], d["offsets"], d["ijk"], d["p16"],
                             d["rgb_out"], d["votes_out"], **self.cfg)                                 # :153-160
        if self.guard.check("SparseLoop.run") is not None:
            ctx.numeric_status()
            return self.run(viewPairs, xyz, resol, w)              # premultipliers recalibrated on this batch: redo it (the guard acts once)
        off = self._off[:n + 1]
        ctx.d2h(off, d["offsets"])
        ctx.synchronize()
        T = int(off[-1])
        ijk = np.empty((T, 3), np.uint8); p16 = np.empty((T,), np.float16); rgb = np.empty((T, 3), np.uint8)
        votes = np.empty((T,), np.uint8) if self.cfg["enable_rayPooling"] else None
        if T:
            ctx.d2h(ijk, d["ijk"]); ctx.d2h(p16, d["p16"]); ctx.d2h(rgb, d["rgb_out"])
            if votes is not None:
                ctx.d2h(votes, d["votes_out"])
        nonempty = [int(i) for i in np.nonzero(np.diff(off))[0]]
        cut = lambda a: [a[off[i]:off[i + 1]] for i in nonempty]
        return nonempty, cut(ijk), cut(p16), cut(rgb), (cut(votes) if votes is not None else []), xyz_new


    def run_many(self, viewPairs, xyz, resol, w=None):
        """The same for ANY number of cubes, software-pipelined over batches of `max_cubes`: all cube parameters go up in
        one piece, the kernels of batches i+1 and i+2 are enqueued before the sparse lists of batch i are fetched (three sets of
        output buffers), so the GPU never waits for the host's small D2H copies. Same return value as `run`."""
        ctx, d, n_vp, B = self.ctx, self.d, self.n_vp, self.max_cubes
        pairs = np.ascontiguousarray(viewPairs, dtype=np.int64)
        n = pairs.shape[0]
        if pairs.shape[1:] != (n_vp, 2):
            raise ValueError("viewPairs must have shape (n, %d, 2)" % n_vp)
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(n, 3)
        resol = np.ascontiguousarray(resol, dtype=np.float32).reshape(n)
        xyz_new = xyz + (resol[:, None] * self.lo).astype(np.float32) if self.cfg["enable_centerCrop"] else xyz.copy()
        out = ([], [], [], [], [], xyz_new)
        if n == 0:
            return out
        V = ctx.n_views
        if pairs.max() >= V or pairs.min() < -V:
            raise IndexError("view index out of range for %d views" % V)
        pairs = np.where(pairs < 0, pairs + V, pairs)
        w = np.full((n, n_vp), 1.0 / n_vp, np.float32) if w is None else np.ascontiguousarray(w, dtype=np.float32).reshape(n, n_vp)
        votes_on = bool(self.cfg["enable_rayPooling"])
        if "ijk3" not in d:          # second and third set of output buffers
            cap = B * self.dc ** 3
            for t in ("2", "3"):
                d.update({"offsets" + t: ctx.dev_alloc((B + 1) * 8), "ijk" + t: ctx.dev_alloc(cap * 3), "p16" + t: ctx.dev_alloc(cap * 2),
                          "rgb_out" + t: ctx.dev_alloc(cap * 3), "votes_out" + t: ctx.dev_alloc(cap)})
        outs = [dict(offsets=d["offsets" + t], ijk=d["ijk" + t], p16=d["p16" + t], rgb_out=d["rgb_out" + t], votes_out=d["votes_out" + t])
                for t in ("", "2", "3")]
        gp, gx, gr, gw = ctx.upload(pairs), ctx.upload(xyz), ctx.upload(resol), ctx.upload(w)
        try:
            def enqueue(i0, o):
                m = min(B, n - i0)
                pp, px, pr, pw = gp + i0 * n_vp * 16, gx + i0 * 12, gr + i0 * 4, gw + i0 * n_vp * 4
                ctx.cvc_forward_dev(m, n_vp, pp, px, pr, pw, d["fused"], d["unfused"], d["cvc"], mean=self.mean)
                ctx.color_fuse_dev(m, n_vp, d["cvc"], d["unfused"], pw, d["rgb"], mean=self.mean)
                ctx.dense2sparse_dev(m, n_vp, pp, px, pr, d["fused"], d["rgb"], d["votes"], o["offsets"], o["ijk"], o["p16"], o["rgb_out"],
                                     o["votes_out"], **self.cfg)
                return m

            def fetch(i0, m, o, slot):
                off = np.zeros((m + 1,), dtype=np.int64)
                ctx.d2h_after(slot, off, o["offsets"])
                T = int(off[-1])
                ijk = np.empty((T, 3), np.uint8); p16 = np.empty((T,), np.float16); rgb = np.empty((T, 3), np.uint8)
                votes = np.empty((T,), np.uint8) if votes_on else None
                if T:
                    ctx.d2h_after(slot, ijk, o["ijk"]); ctx.d2h_after(slot, p16, o["p16"]); ctx.d2h_after(slot, rgb, o["rgb_out"])
                    if votes_on:
                        ctx.d2h_after(slot, votes, o["votes_out"])
                for i in np.nonzero(np.diff(off))[0]:
                    a, b = off[i], off[i + 1]
                    out[0].append(int(i0 + i)); out[1].append(ijk[a:b]); out[2].append(p16[a:b]); out[3].append(rgb[a:b])
                    if votes_on:
                        out[4].append(votes[a:b])

            # Two batches are always enqueued ahead of the one being fetched (three output sets), and the fetch copies on a second stream
            # that waits only for ITS batch (sn_mark / sn_memcpy_d2h_after): the GPU never waits for the host's five small copies and
            # the unpacking. (A plain d2h on the one stream is ordered behind everything enqueued so far: ~0.5 ms of idle GPU per batch.)
            starts = list(range(0, n, B))
            pending = []
            for k, i0 in enumerate(starts):
                m = enqueue(i0, outs[k % 3])
                if k == 0 and self.guard.enabled and not self.guard.checks:
                    # first batch of this loop's life: one stream synchronisation (~ a batch) to read the saturation word before anything is fetched
                    if self.guard.check("SparseLoop.run_many") is not None:
                        m = enqueue(i0, outs[0])               # premultipliers recalibrated on this batch: redo it
                        ctx.numeric_status()
                ctx.mark(k % 3)
                pending.append((i0, m, outs[k % 3], k % 3))
                if len(pending) == 3:
                    fetch(*pending.pop(0))
            for item in pending:
                fetch(*item)
            ctx.synchronize()                    # surfaces the ray-pooling range error, if any
            if self.guard.checks and self.guard.check("SparseLoop.run_many, later batches") is not None:
                # a batch after the first saturated codes for the first time: the premultipliers are recalibrated (on the call's last batch) for what
                # follows; this call's results stand - a saturated value loses its own correction term only
                ctx.numeric_status()
        finally:
            for p in (gp, gx, gr, gw):
                ctx.dev_free(p)
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


def gather_sparse_sharded(local, lo, group=None, device=None):
    """`local` = dense2sparse's tuple for this rank's cube shard [lo, hi) (indices in `local[0]` are shard-relative).
    Returns the same tuple for ALL cubes on every rank (global cube indices, rank order = cube order). Wire format per
    rank: int64 header (n_cubes, then per cube: global index, voxel count) + packed ijk | pred16 | rgb | votes bytes."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    nonempty, ijk_l, p_l, rgb_l, v_l, xyz_new = local
    counts = np.asarray([len(x) for x in p_l], dtype=np.int64)
    head = np.concatenate([[len(nonempty), int(bool(len(v_l)) or len(p_l) == 0)], np.asarray(nonempty, dtype=np.int64) + lo, counts]).astype(np.int64)
    cat = lambda lst, dt, tail: (np.concatenate(lst) if len(lst) else np.zeros((0,) + tail, dt)).astype(dt, copy=False)
    body = [cat(ijk_l, np.uint8, (3,)).reshape(-1), cat(p_l, np.float16, ()).view(np.uint8).reshape(-1), cat(rgb_l, np.uint8, (3,)).reshape(-1)]
    if len(v_l):
        body.append(cat(v_l, np.uint8, ()).reshape(-1))
    xyz_b = np.ascontiguousarray(xyz_new, dtype=np.float32).view(np.uint8).reshape(-1)
    blob = np.concatenate([np.asarray([head.size, xyz_b.size], np.int64).view(np.uint8), head.view(np.uint8), xyz_b] + body)
    sizes = torch.zeros((world,), dtype=torch.int64, device=device)
    mine = torch.tensor([blob.size], dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    cap = int(sizes.max().item())
    buf = torch.zeros((cap,), dtype=torch.uint8, device=device)
    buf[: blob.size] = torch.from_numpy(blob).to(buf.device)
    allb = torch.empty((world * cap,), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(allb, buf, group=group)
    allb = allb.cpu().numpy()
    sizes_h = sizes.cpu().numpy()           # one device-to-host copy, not one per rank
    out = ([], [], [], [], [], [])
    for r in range(world):
        b = allb[r * cap: r * cap + int(sizes_h[r])]
        nh, nx = (int(v) for v in b[:16].view(np.int64))
        h = b[16:16 + 8 * nh].view(np.int64)
        o = 16 + 8 * nh
        out[5].append(b[o:o + nx].view(np.float32).reshape(-1, 3))
        o += nx
        k, votes_on = int(h[0]), bool(h[1])
        idx, cnt = h[2:2 + k], h[2 + k:2 + 2 * k]
        T = int(cnt.sum())
        ijk = b[o:o + 3 * T].reshape(T, 3); o += 3 * T
        p16 = b[o:o + 2 * T].view(np.float16); o += 2 * T
        rgb = b[o:o + 3 * T].reshape(T, 3); o += 3 * T
        votes = b[o:o + T] if votes_on else None
        ends = np.cumsum(cnt)
        for j in range(k):
            a, e = int(ends[j] - cnt[j]), int(ends[j])
            out[0].append(int(idx[j])); out[1].append(ijk[a:e].copy()); out[2].append(p16[a:e].copy()); out[3].append(rgb[a:e].copy())
            if votes is not None:
                out[4].append(votes[a:e].copy())
    return out[0], out[1], out[2], out[3], out[4], np.concatenate(out[5], axis=0)


def _lap_fn(timings):
    import time
    clock = [time.perf_counter()]

    def lap(name):
        if timings is not None:
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (now - clock[0])
            clock[0] = now
    return lap


def scene_select(images_list, cameraPOs_np, cubes_param_np, cube_D_mm, cube_D, N_viewPairs4inference, patch2embedding_fn, embeddingPair2simil_fn,
                 viewPair_relativeImpt_fn, patches_mean_bgr, batchSize_similNet_patch2embedding=100, batchSize_similNet_embeddingPair2simil=100000,
                 batchSize_viewPair_w=100000, weighted_fusion=True, D_embedding=128, patchSize=64, ctx=None, timings=None):
    """main_reconstruct.py:67-116 for the cubes given: corner / centre projections -> early rejection (similarityNet) -> view-pair selection
    (relative-weight MLP). Every step is per cube, so any subset of the scene's cubes gives that subset's rows. Returns a dict with the
    reference's variable names: patches_embedding (N,V,128), inScope_cubes_vs_views (N,V), dissimilarity (N,P), validCubes (N,) and - when any
    cube is valid - viewPairs4Reconstr (Nv,N_vp,2), w_viewPairs4Reconstr (Nv,N_vp)."""
    from . import camera, earlyRejection, runtime, viewPairSelection
    lap = _lap_fn(timings)
    cameraPOs_np = np.asarray(cameraPOs_np, dtype=np.float64)
    N_vp, N_views, N_cubes = int(N_viewPairs4inference), cameraPOs_np.shape[0], len(cubes_param_np)
    viewPairs = viewPairSelection.k_combination_np(range(N_views), k=2)
    if N_cubes == 0:           # an empty shard (more ranks than cubes): the projections below cannot reshape zero points
        return dict(patches_embedding=np.zeros((0, N_views, D_embedding), np.float32), inScope_cubes_vs_views=np.zeros((0, N_views), bool),
                    dissimilarity=np.zeros((0, len(viewPairs)), np.float32), validCubes=np.zeros((0,), bool))
    with runtime.scene_context(ctx=ctx, cube_D=cube_D):           # the projections / early rejection below run in the scene's own context
        return _scene_select_body(images_list, cameraPOs_np, cubes_param_np, cube_D_mm, N_vp, N_views, N_cubes, viewPairs, patch2embedding_fn,
                                  embeddingPair2simil_fn, viewPair_relativeImpt_fn, patches_mean_bgr, batchSize_similNet_patch2embedding,
                                  batchSize_similNet_embeddingPair2simil, batchSize_viewPair_w, weighted_fusion, D_embedding, patchSize, lap)


def _scene_select_body(images_list, cameraPOs_np, cubes_param_np, cube_D_mm, N_vp, N_views, N_cubes, viewPairs, patch2embedding_fn, embeddingPair2simil_fn,
                       viewPair_relativeImpt_fn, patches_mean_bgr, batchSize_similNet_patch2embedding, batchSize_similNet_embeddingPair2simil,
                       batchSize_viewPair_w, weighted_fusion, D_embedding, patchSize, lap):
    from . import camera, earlyRejection, viewPairSelection
    # main_reconstruct.py:67-71
    img_h_cubesCorner, img_w_cubesCorner = camera.perspectiveProj_cubesCorner(projection_M=cameraPOs_np, cube_xyz_min=cubes_param_np['xyz'],
                                                                              cube_D_mm=cube_D_mm, return_int_hw=False, return_depth=False)
    img_h_cubesCenter, img_w_cubesCenter = camera.perspectiveProj(projection_M=cameraPOs_np, xyz_3D=cubes_param_np['xyz'] + cube_D_mm / 2.,
                                                                  return_int_hw=False, return_depth=False)
    cameraTs_np = viewPairSelection.camera_centers(cameraPOs_np)                                    # main_reconstruct.py:50
    lap("projections")
    # :84-97 early rejection
    patches_embedding, inScope_cubes_vs_views = earlyRejection.patch2embedding(
        images_list, img_h_cubesCorner, img_w_cubesCorner, patch2embedding_fn, patches_mean_bgr, N_cubes, N_views, D_embedding, patchSize=patchSize,
        batchSize=batchSize_similNet_patch2embedding, cubeCenter_hw=np.stack([img_h_cubesCenter, img_w_cubesCenter], axis=0))
    lap("patch2embedding")
    dissimilarity = earlyRejection.embeddingPairs2simil(embeddings=patches_embedding, embeddingPair2simil_fn=embeddingPair2simil_fn,
                                                        inScope_cubes_vs_views=inScope_cubes_vs_views, viewPairs=viewPairs, N_views=N_views,
                                                        batchSize=batchSize_similNet_embeddingPair2simil)
    validCubes = earlyRejection.selectFromSimilarity(dissimilarityProb=dissimilarity, N_viewPairs4inference=N_vp)
    lap("pair_similarity")
    out = dict(patches_embedding=patches_embedding, inScope_cubes_vs_views=inScope_cubes_vs_views, dissimilarity=dissimilarity, validCubes=validCubes)
    if not validCubes.any():
        return out
    # :103-116 view-pair selection
    viewPairs4Reconstr, w_viewPairs4Reconstr = viewPairSelection.viewPairSelection(
        cameraTs_np=cameraTs_np, e_viewPairs=patches_embedding, d_viewPairs=dissimilarity, validCubes=validCubes,
        cubeCenters_xyz=cubes_param_np['xyz'] + cube_D_mm / 2., viewPair_relativeImpt_fn=viewPair_relativeImpt_fn, batchSize=batchSize_viewPair_w,
        N_viewPairs4inference=N_vp, viewPairs=viewPairs)
    if weighted_fusion is False:
        w_viewPairs4Reconstr[:] = 1.0 / N_vp
    out.update(viewPairs4Reconstr=viewPairs4Reconstr, w_viewPairs4Reconstr=w_viewPairs4Reconstr)
    lap("viewpair_selection")
    return out


_LOOP_EMPTY = dict(prediction_list=[], rgb_list=[], vxl_ijk_list=[], rayPooling_votes_list=[], cube_ijk_np=None, param_np=None, viewPair_np=None,
                   vxl_mask_list=[])


def scene_cube_loop(images_list, cameraPOs_np, valid_cubes_param_np, viewPairs4Reconstr, w_viewPairs4Reconstr, cube_D, N_viewPairs4inference,
                    cube_Dcenter, batchSize_nViewPair_SurfaceNet=None, min_prob=0.46, tau=0.7, gamma=0.8, ctx=None, timings=None, auto_calibrate=True):
    """main_reconstruct.py:126-173 for a run of VALID cubes (their rows of the cube table, their selected view pairs and weights): per batch
    CVC, SurfaceNet, fusion, voxel colours, ray pooling, dense2sparse - device-resident (`SparseLoop.run_many`) - then the thinning masks.
    Returns prediction_list, rgb_list, vxl_ijk_list, rayPooling_votes_list, vxl_mask_list, param_np, viewPair_np, cube_ijk_np (rows of the
    non-empty cubes), as the reference accumulates them."""
    from . import runtime, sparseCubes
    lap = _lap_fn(timings)
    out = {k: (list(v) if isinstance(v, list) else v) for k, v in _LOOP_EMPTY.items()}
    if len(valid_cubes_param_np) == 0:
        return out
    N_vp = int(N_viewPairs4inference)
    ctx = ctx or runtime.context_for(cube_D)
    runtime.bind_scene(ctx, np.asarray(cameraPOs_np, dtype=np.float64), images_list)
    bs = int(batchSize_nViewPair_SurfaceNet or max(1, ctx.max_samples // N_vp))
    bs = min(bs, max(1, ctx.max_samples // N_vp))
    loop = SparseLoop(ctx, N_vp, max_cubes=bs, min_prob=min_prob, rayPool_thresh=0, enable_centerCrop=True, cube_Dcenter=cube_Dcenter,
                      enable_rayPooling=True, auto_calibrate=auto_calibrate)
    try:
        # the batches of gen_non0Batch_npBool(validCubes, bs) are consecutive runs of valid cubes: run_many walks exactly those
        nonempty, ijk_l, p_l, rgb_l, v_l, xyz_new = loop.run_many(viewPairs4Reconstr, valid_cubes_param_np['xyz'], valid_cubes_param_np['resol'],
                                                                  w_viewPairs4Reconstr)
    finally:
        loop.close()
    lap("cube_loop")
    param_sub = np.copy(valid_cubes_param_np)
    param_sub['xyz'] = xyz_new
    out.update(prediction_list=p_l, rgb_list=rgb_l, vxl_ijk_list=ijk_l, rayPooling_votes_list=v_l, param_np=param_sub[nonempty],
               viewPair_np=np.asarray(viewPairs4Reconstr).astype(np.uint16)[nonempty], cube_ijk_np=valid_cubes_param_np['ijk'][nonempty])
    # :172-173 thinning
    out["vxl_mask_list"] = sparseCubes.filter_voxels(vxl_mask_list=[], prediction_list=out["prediction_list"], prob_thresh=tau,
                                                     rayPooling_votes_list=out["rayPooling_votes_list"], rayPool_thresh=gamma * N_vp * 2)
    lap("thinning")
    return out


_SELECT_KEYS = ("patch2embedding_fn", "embeddingPair2simil_fn", "viewPair_relativeImpt_fn", "patches_mean_bgr", "batchSize_similNet_patch2embedding",
                "batchSize_similNet_embeddingPair2simil", "batchSize_viewPair_w", "weighted_fusion", "D_embedding", "patchSize")
_LOOP_KEYS = ("cube_Dcenter", "batchSize_nViewPair_SurfaceNet", "min_prob", "tau", "gamma", "auto_calibrate")


def _split_scene_kw(kw):
    """One dict of reconstruct_scene's named arguments -> (keywords of scene_select, keywords of scene_cube_loop)."""
    sel = {k: kw[k] for k in _SELECT_KEYS}
    loop = {k: kw[k] for k in _LOOP_KEYS}
    for d in (sel, loop):
        d["ctx"], d["timings"] = kw.get("ctx"), kw.get("timings")
    return sel, loop


def reconstruct_scene(images_list, cameraPOs_np, cubes_param_np, cube_D_mm, cube_D, N_viewPairs4inference, patch2embedding_fn, embeddingPair2simil_fn,
                      viewPair_relativeImpt_fn, cube_Dcenter, patches_mean_bgr, batchSize_similNet_patch2embedding=100,
                      batchSize_similNet_embeddingPair2simil=100000, batchSize_viewPair_w=100000, batchSize_nViewPair_SurfaceNet=None,
                      weighted_fusion=True, min_prob=0.46, tau=0.7, gamma=0.8, D_embedding=128, patchSize=64, ctx=None, timings=None, auto_calibrate=True):
    """The body of main_reconstruct.reconstruction() between file input and PLY output (main_reconstruct.py:67-173):
    corner / centre projections -> early rejection (similarityNet) -> view-pair selection (relative-weight MLP) ->
    per batch: CVC, SurfaceNet, fusion, voxel colours, ray pooling, dense2sparse -> thinning masks  (= `scene_select` + `scene_cube_loop`).
    patch2embedding_fn, embeddingPair2simil_fn, viewPair_relativeImpt_fn: the callables of similarityNet.similarityNet_inference /
    SurfaceNet.SurfaceNet_inference (their weights are bound in `runtime`); `timings`: a dict that receives the wall seconds of every stage.
    A missing argument fails here, before any stage has run. Returns a dict with the reference's variable names. Not included (SURVEY §2.1,
    out of scope): image / camera readers, cube tiling, cross-cube denoising, PLY / npz writers."""
    sel_kw, loop_kw = _split_scene_kw(locals())
    out = scene_select(images_list, cameraPOs_np, cubes_param_np, cube_D_mm, cube_D, N_viewPairs4inference, **sel_kw)
    valid = out["validCubes"]
    if not valid.any():
        out.update({k: (list(v) if isinstance(v, list) else v) for k, v in _LOOP_EMPTY.items()})
        return out
    out.update(scene_cube_loop(images_list, cameraPOs_np, cubes_param_np[valid], out["viewPairs4Reconstr"], out["w_viewPairs4Reconstr"], cube_D,
                               N_viewPairs4inference, **loop_kw))
    return out


def _allgather_bytes(blob, group=None, device=None, ctx=None):
    """All-gather of one variable-length byte string per rank (np.uint8 1-D): lengths first, then one padded buffer. `ctx`: a Context whose
    own RCCL communicator is up (`comm_init`) - the exchange then runs through the C ABI (`sn_allgatherv_bytes_dev`: device to device over
    xGMI on the context's stream) and torch.distributed is not touched. Otherwise `device` says where the torch collective runs (None =
    CPU tensors for gloo; a CUDA device for RCCL)."""
    if ctx is not None and getattr(ctx, "comm_world", 0):
        return ctx.allgatherv_bytes(blob)
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = torch.zeros((world,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, torch.tensor([blob.size], dtype=torch.int64, device=device), group=group)
    sizes = sizes.cpu().numpy()                      # one device-to-host copy
    cap = max(1, int(sizes.max()))
    buf = torch.zeros((cap,), dtype=torch.uint8, device=device)
    buf[: blob.size] = torch.from_numpy(np.ascontiguousarray(blob)).to(buf.device)
    allb = torch.empty((world * cap,), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(allb, buf, group=group)
    allb = allb.cpu().numpy()
    return [allb[r * cap: r * cap + int(sizes[r])] for r in range(world)]


def _exchange(stage, make_blob, group=None, device=None, ctx=None):
    """Runs `make_blob()` (this rank's stage work -> np.uint8 payload) and all-gathers the payloads. A rank whose work raised does NOT
    leave the others blocked in the collective: every payload is prefixed with a status byte, the exception text travels in its place and
    every rank raises the same RuntimeError after the exchange."""
    try:
        payload, err = make_blob(), None
    except Exception as e:              # noqa: BLE001 - whatever the stage raised must reach the other ranks as a message, not a hang
        payload, err = np.frombuffer(("%s: %s" % (type(e).__name__, e)).encode("utf-8", "replace"), dtype=np.uint8), e
    blob = np.concatenate([np.asarray([0 if err is None else 1], np.uint8), np.asarray(payload, np.uint8).reshape(-1)])
    parts = _allgather_bytes(blob, group=group, device=device, ctx=ctx)
    failed = [(r, bytes(b[1:]).decode("utf-8", "replace")) for r, b in enumerate(parts) if b[0] != 0]
    if failed:
        msg = "; ".join("rank %d: %s" % f for f in failed)
        if err is not None:
            raise RuntimeError("%s failed on %s" % (stage, msg)) from err
        raise RuntimeError("%s failed on %s" % (stage, msg))
    return [b[1:] for b in parts]


def _npz_bytes(arrs):
    import io
    f = io.BytesIO()
    np.savez(f, **arrs)
    return np.frombuffer(f.getvalue(), dtype=np.uint8)


def _npz_load(b):
    import io
    return np.load(io.BytesIO(np.asarray(b, np.uint8).tobytes()), allow_pickle=False)


_SCENE_LISTS = ("prediction_list", "rgb_list", "vxl_ijk_list", "rayPooling_votes_list", "vxl_mask_list")
_SELECT_ROWS = ("patches_embedding", "inScope_cubes_vs_views", "dissimilarity")                          # one row per cube (intermediates)
_SCENE_NONEMPTY_ROWS = ("cube_ijk_np", "param_np", "viewPair_np")                                        # one row per non-empty cube


def _pack_lists(res):
    """scene_cube_loop's dict -> one npz byte string (variable-length lists as a concatenation + lengths)."""
    arrs = {}
    for k in _SCENE_NONEMPTY_ROWS:
        if res.get(k) is not None:
            arrs[k] = np.asarray(res[k])
    for k in _SCENE_LISTS:
        lst = res.get(k) or []
        arrs[k + "/len"] = np.asarray([len(x) for x in lst], dtype=np.int64)
        if lst:
            arrs[k + "/cat"] = np.concatenate([np.asarray(x) for x in lst])
    return _npz_bytes(arrs)


def _merge_lists(blobs):
    """Per-rank npz byte strings (rank order = valid-cube order) -> scene_cube_loop's dict for all valid cubes."""
    parts = [_npz_load(b) for b in blobs]
    out = {}
    for k in _SCENE_NONEMPTY_ROWS:
        have = [p[k] for p in parts if k in p.files]
        out[k] = np.concatenate(have, axis=0) if have else None
    for k in _SCENE_LISTS:
        out[k] = []
        for p in parts:
            lens = p[k + "/len"]
            if lens.size:
                cat, ends = p[k + "/cat"], np.cumsum(lens)
                out[k].extend(cat[e - n: e] for n, e in zip(lens, ends))
    return out


def reconstruct_scene_sharded(images_list, cameraPOs_np, cubes_param_np, cube_D_mm=None, cube_D=None, N_viewPairs4inference=None, patch2embedding_fn=None,
                              embeddingPair2simil_fn=None, viewPair_relativeImpt_fn=None, cube_Dcenter=None, patches_mean_bgr=None,
                              batchSize_similNet_patch2embedding=100, batchSize_similNet_embeddingPair2simil=100000, batchSize_viewPair_w=100000,
                              batchSize_nViewPair_SurfaceNet=None, weighted_fusion=True, min_prob=0.46, tau=0.7, gamma=0.8, D_embedding=128, patchSize=64,
                              ctx=None, timings=None, group=None, comm_device=None, comm="auto", world=None, rank=None, select_fn=None, loop_fn=None,
                              gather_intermediates=False, auto_calibrate=True):
    """`reconstruct_scene` over the ranks of a job (one process per GPU; weights / images / cameras replicated).

    Stage 1 - early rejection and view-pair selection (`scene_select`) - is sharded by RAW cube range: its cost is per cube (V patches each),
    so contiguous ranges of the cube table balance it. Then ONE exchange of what stage 2 needs - the validCubes bits (1 bit per cube), the
    selected pairs and weights of the valid cubes (N_vp x 20 B each) - gives every rank the scene's valid-cube list, and stage 2 - the cube loop
    (`scene_cube_loop`: CVC, CNN, fusion, ray pooling, dense2sparse) - is sharded over THAT list (`main_reconstruct.py:126` batches over
    `validCubes`): early rejection keeps 12 % of DTU scan9's cubes and they are spatially clustered, so a cut of the raw table
    would leave most of the loop on a few ranks. A second exchange of the packed sparse lists (9 B per kept voxel) rebuilds on every rank
    the dict `reconstruct_scene` returns (rank order = cube order). No data-path collective inside either stage.

    The exchange (`comm`): "native" - the C ABI's own variable-length all-gather (`sn_allgatherv_bytes_dev`: RCCL on the context's stream, device
    to device over xGMI) on `ctx`, whose communicator the caller has set up (`ctx.comm_init`; `world` / `rank` then default to the context's and
    torch.distributed is not imported); "torch" - torch.distributed on `group` (`comm_device` None: CPU tensors, e.g. gloo; a CUDA device for
    RCCL); "auto": native when `ctx` has a communicator, else torch. `select_fn` / `loop_fn`: stand-ins for the two stages (tests).
    `gather_intermediates` False: the per-cube embeddings / dissimilarities (25 KB per cube at 49 views; no later stage needs them) are NOT
    exchanged - the returned dict holds None under those keys (unlike `reconstruct_scene`'s, which holds the arrays) and this rank's own rows
    under `local_select` (a dict with the three arrays) and `local_range` = (lo, hi) of the raw cube table.
    A rank that raises inside a stage makes every rank raise after the next exchange (never a hang). The returned dict also carries
    `cubes_per_rank` = [(raw cubes, valid cubes in the loop), ...]. Missing stage arguments fail before any work or exchange."""
    native = comm == "native" or (comm == "auto" and ctx is not None and getattr(ctx, "comm_world", 0))
    if native:
        if ctx is None or not getattr(ctx, "comm_world", 0):
            raise ValueError("comm='native' needs a ctx whose communicator is initialised (ctx.comm_init)")
        world, rank = (ctx.comm_world if world is None else int(world)), (ctx.comm_rank if rank is None else int(rank))
    else:
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    xctx = ctx if native else None
    gather_intermediates = bool(gather_intermediates)
    if select_fn is None or loop_fn is None:
        need = dict(cube_D_mm=cube_D_mm, cube_D=cube_D, N_viewPairs4inference=N_viewPairs4inference)
        if select_fn is None:
            need.update(patch2embedding_fn=patch2embedding_fn, embeddingPair2simil_fn=embeddingPair2simil_fn, viewPair_relativeImpt_fn=viewPair_relativeImpt_fn,
                        patches_mean_bgr=patches_mean_bgr)
        if loop_fn is None:
            need.update(cube_Dcenter=cube_Dcenter)
        missing = sorted(k for k, v in need.items() if v is None)
        if missing:
            raise TypeError("reconstruct_scene_sharded: missing argument(s) %s" % ", ".join(missing))
        sel_kw, loop_kw = _split_scene_kw(locals())
    N_cubes = len(cubes_param_np)
    lo, hi = shard_bounds(N_cubes, world, rank)
    local = {}

    def stage1():
        sel = (select_fn(images_list, cameraPOs_np, cubes_param_np[lo:hi]) if select_fn is not None else
               scene_select(images_list, cameraPOs_np, cubes_param_np[lo:hi], cube_D_mm, cube_D, N_viewPairs4inference, **sel_kw))
        local.update(sel)
        valid = np.asarray(sel["validCubes"], dtype=bool)
        arrs = {"bits": np.packbits(valid), "n": np.asarray([valid.size], np.int64)}
        if valid.any():
            arrs["vp"], arrs["w"] = np.asarray(sel["viewPairs4Reconstr"]), np.asarray(sel["w_viewPairs4Reconstr"])
        if gather_intermediates:
            arrs.update({k: np.asarray(sel[k]) for k in _SELECT_ROWS})
        return _npz_bytes(arrs)

    parts = [_npz_load(b) for b in _exchange("early rejection / view-pair selection", stage1, group=group, device=comm_device, ctx=xctx)]
    validCubes = np.concatenate([np.unpackbits(p["bits"])[: int(p["n"][0])].astype(bool) for p in parts]) if N_cubes else np.zeros((0,), bool)
    have = [p for p in parts if "vp" in p.files]
    out = dict(validCubes=validCubes, viewPairs4Reconstr=None, w_viewPairs4Reconstr=None)
    for k in _SELECT_ROWS:
        out[k] = np.concatenate([p[k] for p in parts], axis=0) if gather_intermediates else None
    out["local_select"], out["local_range"] = {k: local.get(k) for k in _SELECT_ROWS}, (lo, hi)
    n_valid = int(validCubes.sum())
    per_rank_raw = [int(p["n"][0]) for p in parts]
    if n_valid == 0:
        out.update({k: (list(v) if isinstance(v, list) else v) for k, v in _LOOP_EMPTY.items()})
        out["cubes_per_rank"] = [(r, 0) for r in per_rank_raw]
        return out
    vp, w = np.concatenate([p["vp"] for p in have], axis=0), np.concatenate([p["w"] for p in have], axis=0)
    out.update(viewPairs4Reconstr=vp, w_viewPairs4Reconstr=w)
    valid_rows = cubes_param_np[validCubes]
    vlo, vhi = shard_bounds(n_valid, world, rank)                     # contiguous ranges of the VALID list: equal to within one cube

    def stage2():
        res = (loop_fn(images_list, cameraPOs_np, valid_rows[vlo:vhi], vp[vlo:vhi], w[vlo:vhi]) if loop_fn is not None else
               scene_cube_loop(images_list, cameraPOs_np, valid_rows[vlo:vhi], vp[vlo:vhi], w[vlo:vhi], cube_D, N_viewPairs4inference, **loop_kw))
        return _pack_lists(res)

    out.update(_merge_lists(_exchange("cube loop", stage2, group=group, device=comm_device, ctx=xctx)))
    out["cubes_per_rank"] = [(per_rank_raw[r], shard_bounds(n_valid, world, r)[1] - shard_bounds(n_valid, world, r)[0]) for r in range(world)]
    return out
