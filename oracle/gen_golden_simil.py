#!/usr/bin/env python3
"""oracle/gen_golden_simil.py — TEST INFRASTRUCTURE. Generates tests/golden/simil_cases.npz by EXECUTING THE REFERENCE's
early-rejection host code in the build container. Numbers only are recorded.

Reference functions run:
  utils/image.py:9-36      preprocess_patches              (incl. its doctest input)
  utils/image.py:92-183    cropImgPatches                  (pyramidRate = 1, interp_order = 2: the call form of earlyRejection.py:50)
  utils/image.py:186-205   img_hw_cubesCorner_inScopeCheck
  utils/camera.py:188-245  perspectiveProj_cubesCorner     (incl. its doctest inputs)
  utils/earlyRejection.py  patch2embedding, embeddingPairs2simil, selectFromSimilarity
The two network callables those functions take are replaced here by EXACTLY-ROUNDED stand-ins (`toy_embedding`,
`toy_pair_simil`: exact float64 sums, one IEEE division) so that the recorded arrays are reproducible bit for bit on any
host; tests/golden_util.py restates them. (The network itself is checked against oracle/simil_oracle.py on the GPU.)

Python-2 / old-numpy accommodations, applied in memory at import time: `np.int` / `np.bool` aliases (removed in numpy
1.24) are set to int / bool; image.py:141 `patchSize / 2` is Python-2 integer division, executed as `//`;
earlyRejection.py / image.py are otherwise run unmodified. Also checked here: scipy's spline zoom at rate 1.0 returns the
uint8 image unchanged (assert below), which is what the product's cropImgPatches relies on.

Usage:  python oracle/gen_golden_simil.py   (from the repo root)
"""
import io
import os
import sys
import math
import types
import warnings
import contextlib
import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_reference_modules():
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "utils"))
    sys.path.insert(0, os.path.join(REF, "utils"))
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "bool"):
        np.bool = bool
    try:
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import camera
            import utils as ref_utils
            src = open(os.path.join(REF, "utils", "image.py")).read()
            assert src.count("patchSize_r = patchSize / 2") == 1
            src = src.replace("patchSize_r = patchSize / 2", "patchSize_r = patchSize // 2")
            image = types.ModuleType("image")
            image.__file__ = os.path.join(REF, "utils", "image.py")
            exec(compile(src, "image", "exec"), image.__dict__)
            sys.modules["image"] = image               # earlyRejection imports it by name
            import earlyRejection
    finally:
        os.chdir(cwd)
    return camera, ref_utils, image, earlyRejection


def main():
    import golden_util
    camera, ref_utils, image, er = load_reference_modules()
    import scipy.ndimage
    P = np.load(os.path.join(OUT, "cameras.npz"))["P_dtu"]
    out = {}

    # ---- preprocess_patches: doctest input (image.py:24-33) + a random one ----
    out["pre_doc"] = image.preprocess_patches(np.zeros((2, 2, 5, 3)), mean_BGR=np.array([1, 2, 3]))
    rs = np.random.RandomState(3)
    pat = rs.randint(0, 256, (3, 64, 64, 3)).astype(np.uint8)
    mean = np.asarray([103.939, 116.779, 123.68]).astype(np.float32)            # params.py:130
    out["pre_in"], out["pre_mean"] = pat, mean
    out["pre_out"] = image.preprocess_patches(pat.astype(np.float32), mean_BGR=mean)

    # ---- perspectiveProj_cubesCorner: doctest inputs (camera.py:211-219) + DTU cameras ----
    np.random.seed(201611)
    Ms, pts = np.random.rand(2, 3, 4), np.random.rand(2, 3)
    h, w = camera.perspectiveProj_cubesCorner(Ms, pts, cube_D_mm=1, return_int_hw=False)
    out["cc_doc_Ms"], out["cc_doc_pts"], out["cc_doc_h"], out["cc_doc_w"] = Ms, pts, h, w

    # ---- a small scene: 3 views 600x800, 7 cubes, some outside a view ----
    hw = (600, 800)
    imgs = [golden_util.synth_image(700 + v, hw[0], hw[1]) for v in range(3)]
    for im in imgs:   # the premise of the product's cropImgPatches: zoom 1.0 with a quadratic spline is the identity on uint8
        z = scipy.ndimage.zoom(input=im, zoom=(1.0, 1.0, 1.0), output=im.dtype, order=2)
        assert z.shape == im.shape and np.array_equal(z, im)
    cube_D_mm = np.float32(0.4 * 32)
    xyz = np.asarray([[-20.0, -30.0, 600.0], [35.5, 10.25, 640.0], [-150.0, -100.0, 630.0], [5.1, -31.9, 590.7], [148.0, 97.0, 635.0],
                      [2000.0, 2000.0, 100.0], [-13.7, 20.3, 601.2]], dtype=np.float32)
    # DTU cameras are calibrated for 1200x1600: scale the image plane by 1/2 for the 600x800 synthetic views
    Ps = P[:3].copy()
    Ps[:, :2, :] *= 0.5
    img_h_c, img_w_c = camera.perspectiveProj_cubesCorner(Ps, xyz, cube_D_mm=cube_D_mm, return_int_hw=False)
    ctr_h, ctr_w = camera.perspectiveProj(Ps, xyz + cube_D_mm / 2., return_int_hw=False)
    out.update(sc_P=Ps, sc_xyz=xyz, sc_D=np.asarray(cube_D_mm), sc_hw=np.asarray(hw), sc_seeds=np.asarray([700, 701, 702]),
               sc_img_h=img_h_c, sc_img_w=img_w_c, sc_ctr_h=ctr_h, sc_ctr_w=ctr_w)
    N_views, N_cubes = img_h_c.shape[:2]
    ins = np.stack([image.img_hw_cubesCorner_inScopeCheck(hw, img_h_c[v], img_w_c[v]) for v in range(N_views)])
    out["sc_inscope"] = ins
    print("in-scope (views x cubes):\n", ins.astype(int))

    # ---- cropImgPatches exactly as earlyRejection.py:50 calls it (plus centres that hang over every image border) ----
    ch = np.asarray([300.7, 31.2, -5.5, 599.9, 650.0, 10.0, 32.0, 567.99])
    cw = np.asarray([400.2, 31.9, 20.0, 799.5, 790.0, 900.0, 32.5, 767.01])
    rh = np.stack([ch - 20, ch + 20], axis=1)
    patches = image.cropImgPatches(img=imgs[1], range_h=rh, range_w=rh, patchSize=64, pyramidRate=1, interp_order=2, cubeCenter_hw=(ch, cw))
    out["crop_ch"], out["crop_cw"], out["crop_view"], out["crop_out"] = ch, cw, np.asarray(1), patches
    # cubeCenter_hw=None branch: centre = mean of the ranges
    rw = np.stack([cw - 10, cw + 30], axis=1)
    out["crop_rh"], out["crop_rw"] = rh, rw
    out["crop_out_ranges"] = image.cropImgPatches(img=imgs[1], range_h=rh, range_w=rw, patchSize=64, pyramidRate=1, interp_order=2)

    # ---- earlyRejection with the exactly-rounded stand-in callables ----
    emb, inscope = er.patch2embedding(imgs, img_h_c, img_w_c, golden_util.toy_embedding, mean, N_cubes, N_views, 128, patchSize=64, batchSize=3,
                                      cubeCenter_hw=np.stack([ctr_h, ctr_w], axis=0))
    assert np.array_equal(inscope.T, ins)
    viewPairs = ref_utils.k_combination_np(range(N_views), k=2)
    dis = er.embeddingPairs2simil(embeddings=emb, embeddingPair2simil_fn=golden_util.toy_pair_simil, inScope_cubes_vs_views=inscope,
                                  viewPairs=viewPairs, N_views=N_views, batchSize=4)
    sel = {n: er.selectFromSimilarity(dis, n) for n in (1, 2, 3)}
    out.update(er_emb=emb, er_inscope=inscope, er_dis=dis, er_sel1=sel[1], er_sel2=sel[2], er_sel3=sel[3])
    print("dissimilarity:\n", np.round(dis, 3), "\nselected (N=1,2,3):", [int(sel[n].sum()) for n in (1, 2, 3)])
    np.savez_compressed(os.path.join(OUT, "simil_cases.npz"), **out)
    print("simil_cases.npz %d bytes" % os.path.getsize(os.path.join(OUT, "simil_cases.npz")))


if __name__ == "__main__":
    main()
