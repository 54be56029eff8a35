#!/usr/bin/env python3
"""oracle/gen_golden_configs.py — TEST INFRASTRUCTURE. CVC vectors AT THE SIZES OF BASELINE.json's CONFIGS, produced by EXECUTING THE REFERENCE.

tests/golden/cvc_cases.npz pins the CVC restatement (oracle/cvc_oracle.c) and the HIP kernel at s in {8, 16, 32} on hand-placed cubes; the
bench's own workloads - BASELINE configs[1] (synthetic 2-view 1200x1600 frame, s = 32: surfacenet_amd/synthetic.synthetic_scene) and
configs[3] (s = 64) - were checked against that oracle only. This script runs the reference's own utils/CVC.py (gen_coloredCubes,
CVC.py:56-104, through the in-memory lib2to3 print fix of oracle/gen_golden.py) on exactly those workloads and on a hand-placed s = 64 cube that
straddles the image border, and stores, per case, a CHECKSUM-STYLE fixture: sha256 of the uint8 output, 4,096 sampled voxels (seeded flat
indices), per-sample per-channel sums, the in-scope fraction - a few KB instead of 3 MB per s = 64 sample. Inputs are regenerated from seeds.
Runs only in the build container (needs /root/reference). -> tests/golden/cvc_config_cases.npz

Usage:  python oracle/gen_golden_configs.py   (from the repo root)
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden                                     # noqa: E402  (load_reference_modules)
from surfacenet_amd import synthetic                  # noqa: E402

N_SAMPLED = 4096


def digest(out_u8):
    """What the fixture keeps of a (N, 6, s, s, s) uint8 CVC tensor."""
    flat = out_u8.reshape(-1)
    idx = np.random.RandomState(20260929).randint(0, flat.size, N_SAMPLED).astype(np.int64)
    return {"sha256": np.frombuffer(hashlib.sha256(np.ascontiguousarray(out_u8).tobytes()).digest(), dtype=np.uint8),
            "idx": idx, "val": flat[idx], "chan_sum": out_u8.reshape(out_u8.shape[0], 6, -1).sum(axis=2, dtype=np.int64),
            "shape": np.asarray(out_u8.shape, dtype=np.int64),
            "inscope": np.asarray((out_u8.reshape(out_u8.shape[0], 2, 3, -1).max(axis=2) > 0).mean())}


def main():
    _, cvc, _ = gen_golden.load_reference_modules()
    cases = {}

    def run(name, sc, s):
        out = cvc.gen_coloredCubes(selected_viewPairs=sc["pairs"], xyz=sc["xyz"], resol=sc["resol"], cameraPOs=sc["cams"], models_img=sc["imgs"],
                                   colorize_cube_D=s, visualization_ON=False)
        assert out.dtype == np.float32 and np.array_equal(out, np.round(out)) and out.min() >= 0 and out.max() <= 255
        d = digest(out.astype(np.uint8))
        for k, v in d.items():
            cases[name + "/" + k] = v
        print("%-28s out %s  in-scope %.4f  sha256 %s" % (name, tuple(d["shape"]), float(d["inscope"]), bytes(d["sha256"]).hex()[:16]))

    # BASELINE configs[1]: the bench's synthetic scene, s = 32, full 1200x1600 frames; 4 cubes x 2 view pairs (the generator of bench.py, seed 0)
    n, n_vp = 4, 2
    sc = synthetic.synthetic_scene(n, n_vp, s=32, seed=0)
    cases["cfg1_s32/n"], cases["cfg1_s32/n_vp"], cases["cfg1_s32/s"], cases["cfg1_s32/seed"] = (np.asarray(v, np.int64) for v in (n, n_vp, 32, 0))
    run("cfg1_s32", sc, 32)
    # BASELINE configs[3]: s = 64; 2 cubes x 2 view pairs of the s64 leg's scene
    n = 2
    sc = synthetic.synthetic_scene(n, n_vp, s=64, seed=0)
    cases["cfg3_s64/n"], cases["cfg3_s64/n_vp"], cases["cfg3_s64/s"], cases["cfg3_s64/seed"] = (np.asarray(v, np.int64) for v in (n, n_vp, 64, 0))
    run("cfg3_s64", sc, 64)
    # s = 64 with out-of-scope voxels: the same views, one cube across the image border, one beside a camera (w close to 0 for some voxels), N_vp = 3
    sc = synthetic.synthetic_scene(2, 3, s=64, seed=5)
    sc["xyz"] = np.asarray([[-175.0, -112.0, 628.0], [148.0, 97.0, 600.0]], dtype=np.float32)
    sc["resol"] = np.asarray([0.8, 0.4], dtype=np.float32)
    cases["edge_s64/xyz"], cases["edge_s64/resol"], cases["edge_s64/pairs"] = sc["xyz"], sc["resol"], sc["pairs"]
    cases["edge_s64/s"] = np.asarray(64, np.int64)
    run("edge_s64", sc, 64)
    path = os.path.join(ROOT, "tests", "golden", "cvc_config_cases.npz")
    np.savez_compressed(path, **cases)
    print("%s: %d bytes" % (path, os.path.getsize(path)))


if __name__ == "__main__":
    main()
