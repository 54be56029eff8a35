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
