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
