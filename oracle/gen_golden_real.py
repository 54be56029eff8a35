#!/usr/bin/env python3
"""oracle/gen_golden_real.py — TEST INFRASTRUCTURE. CVC golden cases on REAL dataset pixels, produced by EXECUTING THE REFERENCE.

Runs only in the build container (needs /root/reference and PIL). Every other CVC fixture uses seeded noise images; this one decodes two
DTU scan9 views (inputs/DTU_MVS/Rectified/scan9/rect_001/002_3_r5000.jpg) and two Middlebury dino views (dinoSR0007/0008.png), cuts a
128 x 160 window around the projection of a cube on the object out of each, shifts the principal point of the view's P matrix by the
window origin (P' = T P, T = translation by (-x0, -y0): the window is then a self-contained small image of the same scene), and runs the
reference's own `CVC.gen_coloredCubes` / `preprocess_augmentation` (utils/CVC.py:56-111, through the in-memory lib2to3 `fix_print` of
oracle/gen_golden.py) on (windows, P', cubes). Stored: the windows as uint8 ARRAYS (decoded pixels, no image file), P', cube parameters,
expected outputs -> tests/golden/real_cases.npz. What it adds over the noise fixtures: piecewise-smooth, heavy-tailed inputs (dark
background, specular highlights, object edges) for the CNN parity tests, and out-of-window voxels on a real silhouette.

Usage:  python oracle/gen_golden_real.py   (from the repo root)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden                                              # noqa: E402  (load_reference_modules, REF, OUT, MEAN6)

WIN_H, WIN_W = 128, 160


def window_case(cvc, camera, name, P, imgs, centre_xyz, resol, s, cases):
    """Windows around the projection of the cube centred at `centre_xyz` + a second cube shifted so that it leaves the windows."""
    P = np.asarray(P, dtype=np.float64)
    half = 0.5 * s * resol
    xyz0 = (np.asarray(centre_xyz, np.float64) - half).astype(np.float32)
    h, w = camera.perspectiveProj(P, np.asarray([centre_xyz], np.float64), return_int_hw=True)
    wins, Pw = [], []
    for v in range(P.shape[0]):
        H, W = imgs[v].shape[:2]
        y0 = int(np.clip(h[v, 0] - WIN_H // 2, 0, H - WIN_H))
        x0 = int(np.clip(w[v, 0] - WIN_W // 2, 0, W - WIN_W))
        wins.append(np.ascontiguousarray(imgs[v][y0:y0 + WIN_H, x0:x0 + WIN_W]))
        T = np.array([[1.0, 0.0, -x0], [0.0, 1.0, -y0], [0.0, 0.0, 1.0]])
        Pw.append(T @ P[v])
    Pw = np.stack(Pw)
    xyz = np.stack([xyz0, xyz0 + np.float32(1.5 * s * resol) * np.asarray([1, 0.3, 0], np.float32)]).astype(np.float32)   # the second cube crosses the window border
    res = np.asarray([resol, resol], np.float32)
    pairs = np.asarray([[[0, 1]], [[1, 0]]], dtype=np.int64)
    out = cvc.gen_coloredCubes(selected_viewPairs=pairs, xyz=xyz, resol=res, cameraPOs=Pw, models_img=wins, colorize_cube_D=s, visualization_ON=False)
    assert out.dtype == np.float32 and np.array_equal(out, np.round(out))
    _, pre = cvc.preprocess_augmentation(None, out.copy(), mean_rgb=gen_golden.MEAN6[None, :, None, None, None], augment_ON=False, crop_ON=False)
    inscope = [float((out[i].reshape(2, 3, -1).max(axis=1) > 0).mean()) for i in range(out.shape[0])]
    cases.update({name + "/P": Pw, name + "/imgs": np.stack(wins), name + "/pairs": pairs, name + "/xyz": xyz, name + "/resol": res,
                  name + "/s": np.asarray(s, np.int64), name + "/out_u8": out.astype(np.uint8), name + "/pre_f32_cube0": pre[:1]})
    print("%-10s windows %s, in-scope fraction per sample %s, mean colour %s" % (name, wins[0].shape, np.round(inscope, 3), np.round(out[0].reshape(6, -1).mean(axis=1), 1)))


def main():
    from PIL import Image
    camera, cvc, _ = gen_golden.load_reference_modules()
    REF = gen_golden.REF
    cases = {}
    P_dtu = camera.readCameraPOs_as_np(os.path.join(REF, "inputs/DTU_MVS/SampleSet/MVS Data/Calibration/cal18"), "DTU", "pos_#.txt", 9, [1, 2])
    dtu = [np.asarray(Image.open(os.path.join(REF, "inputs/DTU_MVS/Rectified/scan9/rect_%03d_3_r5000.jpg" % v)).convert("RGB")) for v in (1, 2)]
    assert dtu[0].shape == (1200, 1600, 3) and dtu[0].dtype == np.uint8
    # a cube on the scan9 object (bounding box centre region, params.py:168-172), resol 0.4, s = 32
    window_case(cvc, camera, "dtu_real", P_dtu, dtu, [10.0, -30.0, 650.0], np.float32(0.4), 32, cases)
    P_mid = camera.readCameraPOs_as_np(os.path.join(REF, "inputs/Middlebury/dinoSparseRing"), "Middlebury", "dinoSR_par.txt", "dinoSparseRing", [7, 8])
    mid = [np.asarray(Image.open(os.path.join(REF, "inputs/Middlebury/dinoSparseRing/dinoSR%04d.png" % v)).convert("RGB")) for v in (7, 8)]
    assert mid[0].shape == (480, 640, 3)
    window_case(cvc, camera, "mid_real", P_mid, mid, [-0.02, 0.02, -0.02], np.float32(0.00025), 32, cases)
    out = os.path.join(gen_golden.OUT, "real_cases.npz")
    np.savez_compressed(out, **cases)
    print("%s: %d bytes" % (out, os.path.getsize(out)))


if __name__ == "__main__":
    main()
