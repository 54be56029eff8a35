#!/usr/bin/env python3
"""oracle/gen_golden.py — TEST INFRASTRUCTURE. Generates tests/golden/*.npz by EXECUTING THE REFERENCE.

Runs only in the build container (needs /root/reference). It imports the reference's own numpy code
(utils/camera.py as-is; utils/CVC.py through an in-memory lib2to3 `fix_print`, the file's single
py2-only statement is CVC.py:74; utils/utils.py as-is) and records input parameters + expected
outputs as small arrays. No reference source text is written anywhere; only numbers.

Fixtures written (all consumed by tests/, never by the product):
  cameras.npz      P matrices read by the reference's readers (DTU cal18 pos_001..004,
                   Middlebury dinoSR views 7..9 = K[R|t] as computed by camera.py:26-58)
  cvc_cases.npz    gen_coloredCubes (CVC.py:56-104) outputs for seeded synthetic images
                   (images are regenerated from the stored seed, never stored) + one
                   preprocess_augmentation (CVC.py:108-111) output
  proj_cases.npz   camera.perspectiveProj (camera.py:123-184) float + rounded outputs, incl. the
                   doctest inputs of camera.py:144-160
  batch_cases.npz  utils.gen_non0Batch_npBool (utils/utils.py:77-110) selectors
  color_cases.npz  utils.generate_voxelLevelWeighted_coloredCubes (utils/utils.py:8-42) output

Usage:  python oracle/gen_golden.py   (from the repo root)
"""
import io
import os
import sys
import types
import contextlib
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
MEAN6 = np.asarray([123.68, 116.779, 103.939, 123.68, 116.779, 103.939]).astype(np.float32)  # params.py:129


def load_reference_modules():
    from lib2to3 import refactor
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "utils"))  # the import-time doctests assume cwd = utils/
    sys.path.insert(0, os.path.join(REF, "utils"))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import camera  # noqa: E402  (reference module, unmodified)
            import utils as ref_utils  # noqa: E402
            src = open(os.path.join(REF, "utils", "CVC.py")).read()
            rt = refactor.RefactoringTool(["lib2to3.fixes.fix_print"])
            src3 = str(rt.refactor_string(src, "CVC.py"))
            cvc = types.ModuleType("ref_CVC")
            exec(compile(src3, "ref_CVC", "exec"), cvc.__dict__)
    finally:
        os.chdir(cwd)
    return camera, cvc, ref_utils


def synth_image(seed, H, W):
    """The one image generator shared with the tests (tests/golden_util.py restates it)."""
    return np.random.RandomState(seed).randint(0, 256, (H, W, 3)).astype(np.uint8)


def main():
    camera, cvc, ref_utils = load_reference_modules()
    os.makedirs(OUT, exist_ok=True)

    # ---------------- cameras ----------------
    dtu_dir = os.path.join(REF, "inputs/DTU_MVS/SampleSet/MVS Data/Calibration/cal18")
    P_dtu = camera.readCameraPOs_as_np(dtu_dir, "DTU", "pos_#.txt", 9, [1, 2, 3, 4])
    P_mid = camera.readCameraPOs_as_np(os.path.join(REF, "inputs/Middlebury/dinoSparseRing"), "Middlebury",
                                       "dinoSR_par.txt", "dinoSparseRing", [7, 8, 9])
    np.savez(os.path.join(OUT, "cameras.npz"), P_dtu=P_dtu, P_mid=P_mid)

    # ---------------- CVC cases ----------------
    cases = {}
    meta = []

    def add_case(name, P, HW, seeds, pairs, xyz, resol, s):
        imgs = [synth_image(sd, HW[0], HW[1]) for sd in seeds]
        pairs = np.asarray(pairs, dtype=np.int64)
        xyz = np.asarray(xyz, dtype=np.float32)
        resol = np.asarray(resol, dtype=np.float32)
        out = cvc.gen_coloredCubes(selected_viewPairs=pairs, xyz=xyz, resol=resol, cameraPOs=P,
                                   models_img=imgs, colorize_cube_D=s, visualization_ON=False)
        assert out.dtype == np.float32 and out.shape == (pairs.shape[0] * pairs.shape[1], 6, s, s, s)
        assert np.array_equal(out, np.round(out)) and out.min() >= 0 and out.max() <= 255
        inscope = float((out.reshape(out.shape[0], 2, 3, -1).max(axis=2) > 0).mean())
        cases[name + "/P"] = P
        cases[name + "/HW"] = np.asarray(HW, dtype=np.int64)
        cases[name + "/seeds"] = np.asarray(seeds, dtype=np.int64)
        cases[name + "/pairs"] = pairs
        cases[name + "/xyz"] = xyz
        cases[name + "/resol"] = resol
        cases[name + "/s"] = np.asarray(s, dtype=np.int64)
        cases[name + "/out_u8"] = out.astype(np.uint8)  # values are exact integers 0..255
        meta.append((name, out.shape, round(inscope, 4)))
        return out

    # DTU: images 1200x1600, resol 0.4 (params.py:166). Cube corners chosen interior / on the image
    # border / fully outside so the out-of-scope branch (CVC.py:45) is exercised.
    dtu_hw = (1200, 1600)
    add_case("dtu_s8_vp1", P_dtu, dtu_hw, [11, 12, 13, 14], [[[0, 1]], [[2, 3]], [[1, 1]]],
             [[-10.0, -20.0, 600.0], [35.5, 10.25, 640.0], [0.0, 0.0, 620.0]], [0.4, 0.4, 0.8], 8)
    add_case("dtu_s16_vp2", P_dtu, dtu_hw, [21, 22, 23, 24], [[[0, 1], [1, 0]], [[2, 3], [0, 2]], [[3, 3], [1, 2]]],
             [[-13.7, 20.3, 601.2], [-153.0, -103.0, 637.0], [250.0, 120.0, 560.0]], [0.4, 0.4, 0.4], 16)
    add_case("dtu_s16_vp3_edge", P_dtu, dtu_hw, [31, 32, 33, 34],
             [[[0, 1], [0, 2], [0, 3]], [[1, 2], [1, 3], [2, 3]]],
             [[-170.0, -110.0, 630.0], [2000.0, 2000.0, 100.0]], [1.6, 0.4], 16)
    add_case("dtu_s32_vp2", P_dtu, dtu_hw, [41, 42, 43, 44], [[[0, 1], [1, 0]], [[2, 3], [3, 0]]],
             [[5.1, -31.9, 590.7], [148.0, 97.0, 635.0]], [0.4, 0.4], 32)
    # Middlebury: images 480x640, resol 0.00025 (params.py:177)
    mid_hw = (480, 640)
    add_case("mid_s16_vp2", P_mid, mid_hw, [51, 52, 53], [[[0, 1], [1, 2]], [[2, 0], [1, 1]]],
             [[-0.02, 0.02, -0.02], [0.005, 0.05, -0.01]], [0.00025, 0.0005], 16)
    add_case("mid_s32_vp1", P_mid, mid_hw, [61, 62, 63], [[[0, 2]]], [[-0.03, 0.01, -0.03]], [0.002], 32)

    # preprocess_augmentation (hot-path call: main_reconstruct.py:143)
    raw = cases["dtu_s8_vp1/out_u8"].astype(np.float32)
    _, pre = cvc.preprocess_augmentation(None, raw.copy(), mean_rgb=MEAN6[None, :, None, None, None],
                                         augment_ON=False, crop_ON=False)
    cases["dtu_s8_vp1/pre_f32"] = pre
    np.savez_compressed(os.path.join(OUT, "cvc_cases.npz"), **cases)

    # ---------------- perspectiveProj ----------------
    np.random.seed(201611)  # camera.py:144-147 doctest inputs
    Ms = np.random.rand(2, 3, 4)
    pts = np.random.rand(2, 3)
    h_f, w_f = camera.perspectiveProj(Ms, pts, return_int_hw=False)
    h_i, w_i = camera.perspectiveProj(Ms, pts, return_int_hw=True)
    rs = np.random.RandomState(7)
    pts_dtu = rs.rand(257, 3) * [400, 300, 120] + [-200, -150, 560]
    hd_f, wd_f = camera.perspectiveProj(P_dtu, pts_dtu, return_int_hw=False)
    hd_i, wd_i = camera.perspectiveProj(P_dtu, pts_dtu, return_int_hw=True)
    np.savez(os.path.join(OUT, "proj_cases.npz"), doc_Ms=Ms, doc_pts=pts, doc_h_f=h_f, doc_w_f=w_f, doc_h_i=h_i,
             doc_w_i=w_i, dtu_P=P_dtu, dtu_pts=pts_dtu, dtu_h_f=hd_f, dtu_w_f=wd_f, dtu_h_i=hd_i, dtu_w_i=wd_i)

    # ---------------- batch selectors ----------------
    b = {}
    for i, (seed, n, bs) in enumerate([(0, 11, 3), (1, 40, 14), (2, 5, 8), (3, 64, 64), (4, 33, 1)]):
        ind = np.random.RandomState(seed).rand(n) > 0.4
        if i == 0:
            ind = np.array([0, 1, 1, 1, 0, 0, 1, 0, 1, 1, 1], dtype=bool)  # utils.py:92 doctest
        sel = ref_utils.gen_non0Batch_npBool(ind, bs)
        b["c%d/ind" % i] = ind
        b["c%d/bs" % i] = np.asarray(bs)
        b["c%d/sel" % i] = np.asarray(sel, dtype=bool)
    np.savez(os.path.join(OUT, "batch_cases.npz"), **b)

    # ---------------- voxel-level colour fusion (utils/utils.py:8-42; call site main_reconstruct.py:150-152) -------
    rs = np.random.RandomState(9)
    Nc, Np, D = 3, 2, 8
    cvc_raw = cases["dtu_s8_vp1/out_u8"].astype(np.float32)[:0]          # shape only
    col = rs.randint(0, 256, (Nc * Np, 6, D, D, D)).astype(np.float32)
    X = col - MEAN6[None, :, None, None, None]
    X += MEAN6[None, :, None, None, None]                                  # exactly what the caller holds at :150
    pred = rs.rand(Nc, Np, D, D, D).astype(np.float32)
    wgt = (rs.rand(Nc, Np) + 0.05).astype(np.float32)
    rgb = ref_utils.generate_voxelLevelWeighted_coloredCubes(viewPair_coloredCubes=X, viewPair_surf_predictions=pred,
                                                             weight4viewPair=wgt)
    np.savez_compressed(os.path.join(OUT, "color_cases.npz"), col_u8=col.astype(np.uint8), pred=pred, w=wgt, rgb=rgb)

    for m in meta:
        print("case %-18s out %s in-scope fraction %.4f" % m)
    for f in sorted(os.listdir(OUT)):
        print("%-20s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
