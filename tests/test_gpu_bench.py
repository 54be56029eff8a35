"""GPU (-m gpu): bench.py's multi-GPU branch on the box's one device. The driver launches `bench.py --gpus N` under torch.distributed.run on an
8-GPU node at round end; the builder's boxes have ONE GPU, so the `use_dist` branch - process group, the library's own RCCL communicator
(sn_comm_init_deadline), the verified probe, the double-buffered overlapped all-gather, the max-over-ranks timing - is exercised here with a forced
one-rank group (BENCH_FORCE_DIST=1), through the same launcher and the same command line. No reference counterpart."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra, port):
    env = dict(os.environ, BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cubes", "8", "--no-fast-mode", "--no-cpu-baseline", "--no-s64",
           "--no-simil", "--no-post-pass", "--no-scenes"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, "bench.py under torch.distributed.run failed:\n%s\n%s" % (p.stdout[-2000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert lines, p.stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_dist_branch_native_collective(gpu_required):
    """Default N>1 path: the library's own collective. rc 0, the parallelism string names it, no fall-back note, and the line says which RCCL file
    the entry points were bound to (a process with two RCCL copies must be visible in SCALE_r*.json, not a hang)."""
    j = _run_bench([], 29871)
    cfg = j["config"]
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["value"] > 0 and j["scaling"] == "weak"
    assert "comm_note" not in cfg, cfg
    assert "rccl" in cfg and "librccl" in cfg["rccl"]["file"] and 20000 <= cfg["rccl"]["version_code"] < 30000, cfg
    # torch.distributed's nccl backend loaded RCCL first: the library must have bound THAT copy, not a second one
    assert "already mapped" in cfg["rccl"]["file"], cfg["rccl"]
    assert j["roofline"]["frac"] > 0 and "kernels_ms_per_step" in j


def test_bench_dist_branch_torch_collective(gpu_required):
    """--torch-comm: the torch.distributed all-gather on torch's stream, ordered against the context's stream by events."""
    j = _run_bench(["--torch-comm"], 29873)
    assert j["value"] > 0 and "comm_note" not in j["config"]


def test_comm_init_deadline_and_info_one_rank(gpu_required):
    """sn_comm_init_deadline with a one-rank group joins at once; with a two-rank group whose peer never comes it returns SN_ERR_COMM after the
    deadline and leaves the context usable (VERDICT r4 #4 iii: library users other than bench.py cannot hang in the set-up)."""
    import time
    import numpy as np
    import surfacenet_amd
    f, ver = surfacenet_amd.Context.comm_info()
    assert "librccl" in f and 20000 <= ver < 30000
    with surfacenet_amd.Context(cube_D=16, max_samples=2) as ctx:
        uid = surfacenet_amd.Context.comm_unique_id()
        ctx.comm_init(1, 0, uid, timeout_s=120)
        assert ctx.comm_world == 1
        parts = ctx.allgatherv_bytes(np.arange(1000, dtype=np.uint8))
        assert len(parts) == 1 and np.array_equal(parts[0], np.arange(1000, dtype=np.uint8))
    with surfacenet_amd.Context(cube_D=16, max_samples=2) as ctx:
        uid = surfacenet_amd.Context.comm_unique_id()
        t0 = time.time()
        with pytest.raises(surfacenet_amd.SurfaceNetHipError, match="did not return within"):
            ctx.comm_init(2, 0, uid, timeout_s=5)              # rank 1 never joins
        assert 4 < time.time() - t0 < 60 and ctx.comm_world == 0
        with pytest.raises(surfacenet_amd.SurfaceNetHipError, match="comm_init has not been called"):
            ctx.allgatherv_bytes(np.zeros(4, np.uint8))
        from surfacenet_amd import weights
        ctx.load_param_values(weights.synthetic_param_values(0))      # the context still works
        ctx.forward(np.zeros((1, 6, 16, 16, 16), np.float32), None, n_vp=1)


def _run_bench_ranks(world, extra, port, env_extra=None, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_TEST_SHARE_GPU0="1", **(env_extra or {}))      # (all ranks on device 0 even on a multi-GPU node)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--cubes", "8", "--no-fast-mode", "--no-cpu-baseline", "--no-s64",
           "--no-simil", "--no-post-pass", "--no-scenes"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, "bench.py --gpus %d under torch.distributed.run failed:\n%s\n%s" % (world, p.stdout[-2000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0):\n%s" % p.stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_two_ranks_share_the_gpu_and_fall_back_together(gpu_required):
    """bench.py's N>1 DECISION LOGIC with N = 2 (Next #5): two ranks share the box's one GPU under a gloo process group. RCCL refuses a
    communicator with two ranks on one device (or the set-up runs into the shortened deadline), so on BOTH ranks the native path is off: the MIN
    all-reduce of the per-rank verdicts, the all-ranks fall-back with its `comm_note`, the host-staged all-gather with a real peer, the rank-seeded
    shards and the MAX-over-ranks timing (two different per-rank times in the line) all execute. No reference counterpart."""
    j = _run_bench_ranks(2, ["--dist-backend", "gloo", "--native-deadline", "20"], 29875)
    cfg = j["config"]
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["value"] > 0 and j["scaling"] == "weak"
    assert "comm_note" in cfg and "torch.distributed all-gather instead" in cfg["comm_note"], cfg
    assert cfg["dist_backend"] == "gloo" and len(cfg["elapsed_s_per_rank"]) == 2 and cfg["samples_per_step"] == 2 * 8 * 2
    # value is the whole job over the slowest rank's clock
    slowest = max(cfg["elapsed_s_per_rank"])
    assert abs(j["value"] - 2 * 8 * 2 / slowest) <= 0.02 * j["value"] and abs(j["ms_per_step"] - slowest / 2 * 1e3) <= 0.02 * j["ms_per_step"]
    assert "native" not in cfg["parallelism"] and "gloo" in cfg["parallelism"]


def test_bench_late_rank_does_not_hang_the_early_one(gpu_required):
    """Rank 1 reaches the communicator set-up 15 s late, past a 5 s deadline: rank 0's sn_comm_init_deadline returns, rank 0 waits for its peer in the
    control plane (not inside RCCL), and both ranks finish on the fall-back path."""
    import time
    t0 = time.time()
    j = _run_bench_ranks(2, ["--dist-backend", "gloo", "--native-deadline", "5"], 29877, env_extra={"BENCH_TEST_LATE_RANK1_S": "15"}, timeout=600)
    assert j["n_gpus"] == 2 and j["value"] > 0 and "comm_note" in j["config"], j["config"]
    assert time.time() - t0 < 500


def test_mfma_probe_reports_a_plausible_box_speed(gpu_required):
    """sn_mfma_probe (bench.py -> `box`): a pure fp16 MFMA stream on an MI355X sustains 1.6 .. 2.5 PF (nominal dense peak 2.5 PF at 2.4 GHz; the boxes of
    the pool measure 1.85 .. 2.0 PF at 1.8 .. 1.9 GHz) - and two probes of one box agree to a few per cent."""
    import surfacenet_amd
    with surfacenet_amd.Context(cube_D=16, max_samples=2) as ctx:
        tf, ghz = ctx.mfma_probe(10.0)
        tf2, ghz2 = ctx.mfma_probe(5.0)
    print("sustained fp16 MFMA: %.1f TF at %.3f GHz (second probe %.1f TF, %.3f GHz)" % (tf, ghz, tf2, ghz2))
    assert 1600.0 <= tf <= 2500.0 and 1.2 <= ghz <= 2.5
    assert abs(tf2 - tf) <= 0.08 * tf
    # clocks per MFMA and wave implied by (rate, clock): 4 waves per CU x 256 CUs x 16x16x32 x 2 FLOP per instruction
    per_mfma = 256 * 4 * 2 * 16 * 16 * 32 * ghz * 1e9 / (tf * 1e12)
    assert 15.5 <= per_mfma <= 18.0, per_mfma
