"""CPU, world_size 2, gloo: the N>1 path of the product (contiguous cube shards, one all-gather of fused probabilities)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, s, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from surfacenet_amd import reconstruct
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def compute(lo, hi):                       # stand-in for the GPU work: cube i -> constant i + 0.25
        calls.append((lo, hi))
        return (np.arange(lo, hi, dtype=np.float32)[:, None, None, None, None] + 0.25) * np.ones((1, 1, s, s, s), np.float32)

    full = reconstruct.infer_cubes_sharded(compute, n, s)
    q.put((rank, calls, full[:, 0, 0, 0, 0].tolist(), full.shape))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 8, 1])
def test_sharded_inference_allgather_gloo(n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per = -(-n // 2)
    for rank, calls, vals, shape in res:
        assert shape == (n, 1, 4, 4, 4)
        assert vals == [i + 0.25 for i in range(n)]                       # every rank holds every cube, in order
        lo, hi = min(n, rank * per), min(n, rank * per + per)
        assert calls == ([(lo, hi)] if hi > lo else [])                   # each rank computed only its own shard


def _sparse_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from surfacenet_amd import reconstruct
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = reconstruct.shard_bounds(n, world, rank)
    local = _fake_sparse(lo, hi)                      # stand-in for SparseLoop.run on this rank's shard
    full = reconstruct.gather_sparse_sharded(local, lo)
    q.put((rank, full))
    dist.destroy_process_group()


def _fake_sparse(lo, hi):
    """Deterministic sparse lists for cubes [lo, hi): cube g keeps (g*7) % 5 voxels (so some cubes are empty)."""
    ne, ijk, p, rgb, v = [], [], [], [], []
    for g in range(lo, hi):
        k = (g * 7) % 5
        if k == 0:
            continue
        rs = np.random.RandomState(g)
        ne.append(g - lo)
        ijk.append(rs.randint(0, 26, (k, 3)).astype(np.uint8)); p.append(rs.rand(k).astype(np.float16))
        rgb.append(rs.randint(0, 256, (k, 3)).astype(np.uint8)); v.append(rs.randint(0, 5, k).astype(np.uint8))
    xyz = np.arange(lo, hi, dtype=np.float32)[:, None] * np.ones((1, 3), np.float32)
    return ne, ijk, p, rgb, v, xyz


@pytest.mark.parametrize("n", [7, 2, 1])
def test_sparse_lists_exchange_gloo(n):
    """N>1 path of the GPU post-pass: ranks exchange packed sparse voxel lists, not dense probability cubes."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_sparse_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _fake_sparse(0, n)
    for rank, full in res:
        assert full[0] == want[0]
        for k in (1, 2, 3, 4):
            assert len(full[k]) == len(want[k]) and all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(full[k], want[k])), k
        assert np.array_equal(full[5], want[5])


PARAM_DT = [("xyz", np.float32, (3,)), ("ijk", np.uint32, (3,)), ("resol", np.float32)]


def _fake_select(images_list, cameraPOs_np, cubes, valid_rule=None):
    """Stand-in for stage 1 (scene_select: early rejection + view-pair selection) with its return contract: everything is a deterministic
    function of the cube's 'ijk' tag, so a sharded run must reproduce the one-piece run bit for bit. Cube g is rejected when g % 4 == 3
    (or by `valid_rule`)."""
    g = cubes["ijk"][:, 0].astype(np.int64)
    n, V, N_vp = len(cubes), 3, 2
    rs = lambda i: np.random.RandomState(int(i))
    emb = np.stack([rs(i).rand(V, 8).astype(np.float32) for i in g]) if n else np.zeros((0, V, 8), np.float32)
    valid = ((g % 4) != 3) if valid_rule is None else valid_rule(g)
    out = dict(patches_embedding=emb, inScope_cubes_vs_views=(emb[:, :, 0] > 0.2), dissimilarity=emb[:, :, 1].copy(), validCubes=valid)
    if valid.any():
        gv = g[valid]
        out["viewPairs4Reconstr"] = np.stack([rs(100 + i).randint(0, V, (N_vp, 2)) for i in gv])
        out["w_viewPairs4Reconstr"] = np.stack([rs(200 + i).rand(N_vp).astype(np.float32) for i in gv])
    return out


def _fake_loop(images_list, cameraPOs_np, valid_cubes, vp, w):
    """Stand-in for stage 2 (scene_cube_loop) on a run of valid cubes: cube g yields no voxels when g % 5 == 0."""
    out = dict(prediction_list=[], rgb_list=[], vxl_ijk_list=[], rayPooling_votes_list=[], cube_ijk_np=None, param_np=None, viewPair_np=None,
               vxl_mask_list=[])
    if len(valid_cubes) == 0:
        return out
    rs = lambda i: np.random.RandomState(int(i))
    gv = valid_cubes["ijk"][:, 0].astype(np.int64)
    assert np.array_equal(vp, np.stack([rs(100 + i).randint(0, 3, (2, 2)) for i in gv]))      # the selections travelled with their cubes
    keep = [j for j, i in enumerate(gv) if i % 5]
    for j in keep:
        i, k = gv[j], int(gv[j] % 7) + 1
        out["prediction_list"].append(rs(300 + i).rand(k).astype(np.float16)); out["rgb_list"].append(rs(400 + i).randint(0, 256, (k, 3)).astype(np.uint8))
        out["vxl_ijk_list"].append(rs(500 + i).randint(0, 26, (k, 3)).astype(np.uint8)); out["rayPooling_votes_list"].append(rs(600 + i).randint(0, 5, k).astype(np.uint8))
        out["vxl_mask_list"].append(rs(700 + i).rand(k) > 0.5)
    sub = valid_cubes[keep]
    out.update(param_np=sub, cube_ijk_np=sub["ijk"], viewPair_np=np.asarray(vp).astype(np.uint16)[keep])
    return out


def _fake_scene(images_list, cameraPOs_np, cubes, valid_rule=None):
    """The one-piece run: stage 1 on all cubes, stage 2 on all valid cubes."""
    out = _fake_select(images_list, cameraPOs_np, cubes, valid_rule)
    v = out["validCubes"]
    if v.any():
        out.update(_fake_loop(images_list, cameraPOs_np, cubes[v], out["viewPairs4Reconstr"], out["w_viewPairs4Reconstr"]))
    else:
        out.update(_fake_loop(images_list, cameraPOs_np, cubes[:0], None, None))
    return out


def _scene_cubes(n):
    cubes = np.zeros((n,), dtype=PARAM_DT)
    cubes["ijk"][:, 0] = np.arange(n)
    cubes["xyz"] = np.arange(n, dtype=np.float32)[:, None] * 0.5
    cubes["resol"] = 0.4
    return cubes


def _clustered(g):
    return g >= 13                                   # every valid cube lies in the LAST raw shard of a 2-rank cut of 20 cubes


def _scene_worker(rank, world, port, n, mode, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from surfacenet_amd import reconstruct
    import test_dist_cpu as T
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counts = []

    def loop_fn(imgs, P, valid_cubes, vp, w):
        counts.append(len(valid_cubes))
        if mode == "raise" and rank == 1:
            raise ValueError("SN_ERR_RANGE stand-in on rank 1")
        return T._fake_loop(imgs, P, valid_cubes, vp, w)

    rule = T._clustered if mode == "clustered" else (T._clustered8 if mode == "clustered8" else None)
    try:
        res = reconstruct.reconstruct_scene_sharded([], np.zeros((3, 3, 4)), T._scene_cubes(n), select_fn=lambda i, P, c: T._fake_select(i, P, c, rule),
                                                    loop_fn=loop_fn, gather_intermediates=(mode != "lean"))
        q.put((rank, res, counts))
    except RuntimeError as e:
        q.put((rank, "RuntimeError: %s" % e, counts))
    dist.destroy_process_group()


def _same_scene(a, b, intermediates=True):
    keys = ("validCubes", "viewPairs4Reconstr", "w_viewPairs4Reconstr", "cube_ijk_np", "param_np", "viewPair_np")
    if intermediates:
        keys += ("patches_embedding", "inScope_cubes_vs_views", "dissimilarity")
    for k in keys:
        x, y = a.get(k), b.get(k)
        assert (x is None) == (y is None), k
        assert x is None or (np.array_equal(x, y) and x.dtype == y.dtype), k
    for k in ("prediction_list", "rgb_list", "vxl_ijk_list", "rayPooling_votes_list", "vxl_mask_list"):
        assert len(a[k]) == len(b[k]) and all(np.array_equal(x, y) and x.dtype == y.dtype for x, y in zip(a[k], b[k])), k


def _run_scene_ranks(n, mode, port_base, world=2):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = port_base + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_scene_worker, args=(r, world, port, n, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("n", [11, 3, 1])
def test_sharded_scene_equals_one_piece_gloo(n):
    """reconstruct_scene_sharded over 2 ranks (stage 1 on contiguous raw cube shards, exchange of the valid bits + selections, stage 2 on
    contiguous shards of the VALID list, exchange of the packed sparse lists) == the one-piece run; n = 1 leaves rank 1 with an empty
    raw shard, n = 3 gives rank 1 a raw shard whose only cube is rejected."""
    want = _fake_scene([], None, _scene_cubes(n))
    for rank, full, counts in _run_scene_ranks(n, "full", 33500):
        _same_scene(full, want)


def test_sharded_scene_balances_the_valid_list_gloo():
    """SURVEY §8(e) / main_reconstruct.py:126: the cube loop is partitioned over the VALID cubes. 20 cubes whose 7 valid ones all lie in rank
    1's raw shard: a cut of the raw table would give rank 0 nothing; the per-rank loop counts must differ by at most one. Also the lean
    exchange (no per-cube intermediates) returns None for them and everything else unchanged."""
    want = _fake_scene([], None, _scene_cubes(20), _clustered)
    assert int(want["validCubes"].sum()) == 7 and not want["validCubes"][:13].any()
    for mode in ("clustered",):
        res = _run_scene_ranks(20, mode, 35500)
        per_rank = [c for _, _, c in res]
        assert sorted(sum(c) for c in per_rank) == [3, 4], per_rank
        for rank, full, counts in res:
            _same_scene(full, want)
            assert full["cubes_per_rank"] == [(10, 4), (10, 3)]
    want = _fake_scene([], None, _scene_cubes(9))
    for rank, full, counts in _run_scene_ranks(9, "lean", 37500):
        _same_scene(full, want, intermediates=False)
        assert full["patches_embedding"] is None and full["dissimilarity"] is None
        # ... and the rank's own rows stay available under local_select / local_range
        lo, hi = full["local_range"]
        assert (lo, hi) == ((0, 5) if rank == 0 else (5, 9))
        for k in ("patches_embedding", "inScope_cubes_vs_views", "dissimilarity"):
            assert np.array_equal(full["local_select"][k], want[k][lo:hi]), k


def _clustered8(g):
    return g >= 37                                   # every valid cube lies in the LAST raw shard of an 8-rank cut of 40 cubes


@pytest.mark.parametrize("n,mode", [(5, "full"), (40, "clustered8"), (21, "lean")])
def test_sharded_scene_eight_ranks_gloo(n, mode):
    """The node's real shape (Next #5): 8 ranks. n = 5: three ranks hold EMPTY raw shards and most valid-list shards are empty too;
    n = 40, clustered: the 3 valid cubes all lie in rank 7's raw shard - the second cut hands them to three different ranks, five ranks run an empty
    loop; n = 21: the lean exchange. Every rank must return the one-piece scene."""
    rule = _clustered8 if mode == "clustered8" else None
    want = _fake_scene([], None, _scene_cubes(n), rule)
    res = _run_scene_ranks(n, mode, 41500, world=8)
    assert [r for r, _, _ in res] == list(range(8))
    for rank, full, counts in res:
        assert not isinstance(full, str), full
        _same_scene(full, want, intermediates=(mode != "lean"))
        assert len(full["cubes_per_rank"]) == 8 and sum(a for a, _ in full["cubes_per_rank"]) == n
        assert sum(b for _, b in full["cubes_per_rank"]) == int(want["validCubes"].sum())
    if mode == "clustered8":
        assert int(want["validCubes"].sum()) == 3 and not want["validCubes"][:37].any()
        assert sorted(sum(c) for _, _, c in res) == [0, 0, 0, 0, 0, 1, 1, 1]
        assert res[0][1]["cubes_per_rank"][7][0] == 5 and all(a == 5 for a, _ in res[0][1]["cubes_per_rank"])
    if mode == "full":
        assert [a for a, _ in res[0][1]["cubes_per_rank"]].count(0) == 3


def test_sharded_scene_checks_its_arguments_before_any_work():
    """The stage arguments are named parameters: one that is missing fails at the call, not after the early-rejection stage and its collective
    (ADVICE r3); comm='native' without a context that holds a communicator fails likewise, without importing torch.distributed."""
    from surfacenet_amd import reconstruct
    import inspect
    sig = inspect.signature(reconstruct.reconstruct_scene_sharded)
    assert "cube_Dcenter" in sig.parameters and "patch2embedding_fn" in sig.parameters and "comm" in sig.parameters
    assert "cube_Dcenter" in inspect.signature(reconstruct.reconstruct_scene).parameters
    with pytest.raises(ValueError, match="comm='native' needs a ctx"):
        reconstruct.reconstruct_scene_sharded([], np.zeros((3, 3, 4)), _scene_cubes(3), comm="native")

    class _Comm(object):                  # a context whose communicator is "up": the argument check comes before any exchange
        comm_world, comm_rank = 1, 0

        def allgatherv_bytes(self, blob):
            raise AssertionError("no exchange may start before the arguments are checked")
    with pytest.raises(TypeError, match="missing argument.*cube_Dcenter"):
        reconstruct.reconstruct_scene_sharded([], np.zeros((3, 3, 4)), _scene_cubes(3), 12.8, 32, 2, lambda *a: None, lambda *a: None, lambda *a: None,
                                              patches_mean_bgr=np.zeros(3, np.float32), ctx=_Comm(), comm="native")


def test_sharded_scene_native_exchange_interface():
    """The scene's two exchanges through a context's own all-gather-v (`ctx.allgatherv_bytes` = sn_allgatherv_bytes_dev on a GPU box) instead of
    torch.distributed: same packing, same result. Here a one-rank stand-in that returns the blob it is given (world 1); the GPU suite runs
    the real call (tests/test_gpu_pipeline.py)."""
    from surfacenet_amd import reconstruct

    class _Loopback(object):
        comm_world, comm_rank = 1, 0
        calls = 0

        def allgatherv_bytes(self, blob):
            self.calls += 1
            return [np.array(blob, dtype=np.uint8, copy=True)]
    ctx = _Loopback()
    want = _fake_scene([], None, _scene_cubes(11))
    got = reconstruct.reconstruct_scene_sharded([], np.zeros((3, 3, 4)), _scene_cubes(11), select_fn=lambda i, P, c: _fake_select(i, P, c, None),
                                                loop_fn=_fake_loop, gather_intermediates=True, ctx=ctx, comm="native")
    _same_scene(got, want)
    assert ctx.calls == 2 and got["cubes_per_rank"] == [(11, int(want["validCubes"].sum()))]


def test_sharded_scene_failure_on_one_rank_raises_everywhere_gloo():
    """A rank whose stage raises (SN_ERR_RANGE, a ray-pooling error ...) must not leave the others blocked in the all-gather: the exception
    text travels in the exchange and EVERY rank raises."""
    res = _run_scene_ranks(11, "raise", 39500)
    for rank, full, counts in res:
        assert isinstance(full, str) and "cube loop failed on rank 1" in full and "SN_ERR_RANGE stand-in" in full, full


class _GlooNativeStandIn(object):
    """CPU stand-in of a Context whose communicator is up, for the protocol of `Context.allgatherv_bytes` (`context.allgatherv_two_step`):
    the two C entry points are restated over gloo EXACTLY as sn_api.hip runs them - sn_allgatherv_counts = one 8-byte all-gather;
    sn_allgatherv_bytes_dev = the counts all-gather, then the payload all-gather padded to the largest contribution, the destination size checked
    only afterwards - and every collective first all-gathers its own (kind, size) tag, so that ranks issuing DIFFERENT collectives at the same
    step (the hang / corruption of ADVICE r4) fail the test instead of blocking it."""

    def __init__(self, dist, world, rank):
        self.dist, self.comm_world, self.comm_rank, self.log = dist, world, rank, []

    def _collective(self, kind, nbytes, payload):
        import torch
        tags = [None] * self.comm_world
        self.dist.all_gather_object(tags, (kind, int(nbytes)))
        assert len(set(tags)) == 1, "ranks disagree on the collective at this step: %r" % (tags,)
        self.log.append((kind, int(nbytes)))
        out = torch.empty((self.comm_world * nbytes,), dtype=torch.uint8)
        buf = torch.zeros((nbytes,), dtype=torch.uint8)
        buf[: payload.size] = torch.from_numpy(np.ascontiguousarray(payload))
        self.dist.all_gather_into_tensor(out, buf)
        return out.numpy().reshape(self.comm_world, nbytes)

    def _counts(self, n_local):
        rows = self._collective("counts", 8, np.frombuffer(np.asarray([n_local], np.uint64).tobytes(), np.uint8))
        return [int(np.frombuffer(r.tobytes(), np.uint64)[0]) for r in rows]

    def allgatherv_bytes(self, blob):
        from surfacenet_amd import context
        blob = np.ascontiguousarray(np.asarray(blob, np.uint8).reshape(-1))

        def payload_fn(n_local, total):
            counts = self._counts(n_local)
            if sum(counts) == 0:
                return counts, np.empty((0,), np.uint8)
            cap = (max(counts) + 15) & ~15
            rows = self._collective("payload", cap, blob)
            assert total >= sum(counts), "destination too small: reported only AFTER the payload collective"
            return counts, np.concatenate([rows[r, : counts[r]] for r in range(self.comm_world)])
        return context.allgatherv_two_step(int(blob.size), self._counts, payload_fn)


def _allgatherv_worker(rank, world, port, sizes, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from surfacenet_amd import reconstruct
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = _GlooNativeStandIn(dist, world, rank)
    out = []
    for step, per_rank in enumerate(sizes):
        blob = (np.arange(per_rank[rank], dtype=np.int64) * 7 + 13 * rank + step).astype(np.uint8)
        parts = ctx.allgatherv_bytes(blob)
        out.append([bytes(p) for p in parts])
    # the error path of the scene exchange: rank 1's stage raises (a short error string against rank 0's 3 MB payload)
    def stage():
        if rank == 1:
            raise RuntimeError("SN_ERR_RANGE stand-in")
        return np.full((3 << 20,), 5, np.uint8)
    try:
        reconstruct._exchange("cube loop", stage, ctx=ctx)
        err = None
    except RuntimeError as e:
        err = str(e)
    q.put((rank, out, err, ctx.log))
    dist.destroy_process_group()


def test_native_allgatherv_protocol_very_unequal_blobs_gloo():
    """the first allgatherv_bytes sized its destination from the rank's OWN blob and retried alone when the C call said "too
    small" - a rank with a small blob then issued a counts all-gather against its peers' payload all-gather. Now the sequence is the same on
    every rank (counts; counts + payload) whatever the sizes: 3 bytes against 2 MB, an empty blob, all empty, and the error path of `_exchange`."""
    import torch.multiprocessing as mp
    sizes = [(3, 2 << 20), (1 << 20, 0), (0, 0), (17, 17)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 41000 + os.getpid() % 2000
    procs = [ctx.Process(target=_allgatherv_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][3] == res[1][3]              # same bytes, same collective sequence on both ranks
    for step, per_rank in enumerate(sizes):
        for r in range(2):
            want = (np.arange(per_rank[r], dtype=np.int64) * 7 + 13 * r + step).astype(np.uint8).tobytes()
            assert res[0][1][step][r] == want
    for rank, _, err, log in res:
        assert err is not None and "cube loop failed on rank 1" in err and "SN_ERR_RANGE stand-in" in err, err
        assert [k for k, _ in log] == ["counts", "counts", "payload"] * 2 + ["counts", "counts"] + ["counts", "counts", "payload"] * 2
