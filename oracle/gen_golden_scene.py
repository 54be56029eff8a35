#!/usr/bin/env python3
"""oracle/gen_golden_scene.py — TEST INFRASTRUCTURE. Generates tests/golden/scene_cases.npz by EXECUTING THE REFERENCE.

Runs only in the build container (needs /root/reference). What BASELINE configs 3 and 5 need on the GPU box, as numbers:
  P_dtu49     the 49 DTU projection matrices of params.py:172 (viewList = range(1,50)), read by the reference's own
              camera.readCameraPOs_as_np (utils/camera.py:62-81) from cal18/pos_001..049.txt
  P_mid16     the 16 Middlebury dinoSparseRing matrices K[R|t] computed by camera.py:26-58 from dinoSR_par.txt
  scan9_*     params.load_modelSpecific_params('DTU', 9) inputs (params.py:162-172: resol 0.4, BB from ObsMask9_10.mat) and the
              cube grid scene.initializeCubes (utils/scene.py:7-61) returns for s = 32 (Dcenter 26) and s = 64 (Dcenter 52),
              overlap 1/2 (params.py:107,114): grid extents, cube_D_mm, a strided sample of (index, xyz, ijk, resol) rows
              and float64 checksums over ALL rows — the full 195,360 / 24,420-row tables are regenerated on the GPU box by
              surfacenet_amd/synthetic.py::cube_grid and must reproduce these.
  dino_*      the same for Middlebury dinoSparseRing (params.py:176-182: resol 0.00025, hard-coded BB).
  doc_*       the doctest INPUTS of scene.py:26-40 (its printed rows are stale w.r.t. the code; the executed code is recorded).
Accommodations for Python 3 (numbers are unaffected): scene.py's module-level imports of plyfile / mesh_util (absent here,
unused by initializeCubes) are satisfied by empty stand-in modules; `cubes_ijk.size / 3` (scene.py:53, py2 integer division)
is executed as `//`. No reference source text is written anywhere; only numbers.

Also writes surfacenet_amd/data/calibration.npz: the same P matrices and bounding boxes (dataset calibration data only), which
tools/bench_scene.py and bench.py's scene legs read on the GPU box.

Usage:  python oracle/gen_golden_scene.py   (from the repo root)
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import scipy.io

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_reference_modules():
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "utils"))
    sys.path.insert(0, os.path.join(REF, "utils"))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import camera                                        # reference module, unmodified
            for name in ("plyfile", "mesh_util"):
                if name not in sys.modules:
                    m = types.ModuleType(name)
                    m.PlyData = m.PlyElement = None
                    sys.modules[name] = m
            src = open(os.path.join(REF, "utils", "scene.py")).read()
            assert src.count("cubes_ijk.size / 3") == 1
            src = src.replace("cubes_ijk.size / 3", "cubes_ijk.size // 3")
            src = src.split("import doctest")[0]                 # the doctests print py2 reprs
            scene = types.ModuleType("ref_scene")
            exec(compile(src, "ref_scene", "exec"), scene.__dict__)
    finally:
        os.chdir(cwd)
    return camera, scene


def grid_record(prefix, cubes, cube_D_mm, out):
    n = cubes.shape[0]
    idx = np.unique(np.r_[0:min(n, 64), np.arange(0, n, max(1, n // 997)), max(0, n - 64):n]).astype(np.int64)
    out[prefix + "_n"] = np.int64(n)
    out[prefix + "_cube_D_mm"] = np.float64(cube_D_mm)
    out[prefix + "_grid"] = (cubes["ijk"].max(axis=0).astype(np.int64) + 1)
    out[prefix + "_idx"] = idx
    out[prefix + "_xyz"] = cubes["xyz"][idx]
    out[prefix + "_ijk"] = cubes["ijk"][idx]
    out[prefix + "_resol"] = cubes["resol"][idx]
    w = np.arange(1, n + 1, dtype=np.float64)
    out[prefix + "_xyz_sum"] = cubes["xyz"].astype(np.float64).sum(axis=0)
    out[prefix + "_xyz_wsum"] = (cubes["xyz"].astype(np.float64) * w[:, None]).sum(axis=0)


def main():
    camera, scene = load_reference_modules()
    out = {}
    dtu_dir = os.path.join(REF, "inputs/DTU_MVS/SampleSet/MVS Data/Calibration/cal18")
    out["P_dtu49"] = camera.readCameraPOs_as_np(dtu_dir, "DTU", "pos_#.txt", 9, list(range(1, 50)))
    out["P_mid16"] = camera.readCameraPOs_as_np(os.path.join(REF, "inputs/Middlebury/dinoSparseRing"), "Middlebury", "dinoSR_par.txt",
                                                "dinoSparseRing", list(range(1, 17)))
    with contextlib.redirect_stdout(io.StringIO()):
        # params.py:166-171
        BB9 = scipy.io.loadmat(os.path.join(REF, "inputs/DTU_MVS/SampleSet/MVS Data/ObsMask/ObsMask9_10.mat"))["BB"].T
        out["scan9_BB"] = BB9
        for s, dc in ((32, 26), (64, 52)):
            cubes, dmm = scene.initializeCubes(resol=np.float32(0.4), cube_D=s, cube_Dcenter=dc, cube_overlapping_ratio=1 / 2., BB=BB9)
            grid_record("scan9_s%d" % s, cubes, dmm, out)
        BBd = np.array([(-0.061897, 0.010897), (-0.018874, 0.068227), (-0.057845, 0.015495)], dtype=np.float32)   # params.py:181
        out["dino_BB"] = BBd
        cubes, dmm = scene.initializeCubes(resol=np.float32(0.00025), cube_D=32, cube_Dcenter=26, cube_overlapping_ratio=1 / 2., BB=BBd)
        grid_record("dino_s32", cubes, dmm, out)
        BBdoc = np.array([[3, 88], [-11, 99], [-110, -11]])                                       # scene.py:26 doctest
        cubes, dmm = scene.initializeCubes(resol=1, cube_D=22, cube_Dcenter=10, cube_overlapping_ratio=0.5, BB=BBdoc)
        out["doc_BB"] = BBdoc
        grid_record("doc", cubes, dmm, out)
        # (the doctest's printed xyz rows are stale in the reference: they predate the safeMargin shift of scene.py:46,56 - the code,
        # executed here, starts at BB_min - 6 = [-3, -17, -116]; only `cube_D_mm == 22` of that doctest still holds)
        assert dmm == 22 and np.array_equal(cubes["xyz"][0], [-3, -17, -116])
    np.savez_compressed(os.path.join(OUT, "scene_cases.npz"), **out)
    # the calibration numbers alone (dataset data, no expected outputs) also ship with the package, for the scene benches
    data_dir = os.path.join(os.path.dirname(OUT), "..", "surfacenet_amd", "data")
    os.makedirs(data_dir, exist_ok=True)
    np.savez(os.path.join(data_dir, "calibration.npz"), P_dtu49=out["P_dtu49"], P_mid16=out["P_mid16"], scan9_BB=out["scan9_BB"], dino_BB=out["dino_BB"])
    print("scene_cases.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.endswith("_n") or k.endswith("_grid") or k.startswith("P_")})
    print({k: out[k] for k in out if k.endswith("_n") or k.endswith("_grid") or k.endswith("_cube_D_mm")})


if __name__ == "__main__":
    main()
