#!/usr/bin/env python3
"""tools/simil_view_timing.py - where the early-rejection stage of a whole scene spends its wall time per view (GPU box): the C call (sn_crop_embed) in the worker
thread against the main thread's wait for it, the scatter of its rows, and the gaps between two calls. python tools/simil_view_timing.py [--config dtu_scan9]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--config", default="dtu_scan9"); ap.add_argument("--views", type=int, default=12)
    a = ap.parse_args()
    import surfacenet_amd
    from surfacenet_amd import synthetic, weights, camera, image, runtime
    P, imgs, cubes, cube_D_mm, Dc, n_vp = synthetic.dataset_scene(a.config, 32, 0)
    ctx = runtime.context_for(32)
    ctx.load_simil_param_values(weights.synthetic_simil_param_values(0))
    runtime.bind_scene(ctx, P, imgs)
    t0 = time.perf_counter()
    hc, wc = camera.perspectiveProj_cubesCorner(projection_M=P, cube_xyz_min=cubes['xyz'], cube_D_mm=cube_D_mm, return_int_hw=False, return_depth=False)
    h0, w0 = camera.perspectiveProj(projection_M=P, xyz_3D=cubes['xyz'] + cube_D_mm / 2., return_int_hw=False, return_depth=False)
    print("projections %.3f s" % (time.perf_counter() - t0))
    centers = np.stack([h0, w0], axis=0)
    N, V = len(cubes), len(imgs)
    mean = np.asarray([103.939, 116.779, 123.68], np.float32)
    emb_all = np.zeros((N, V, 128), np.float32)
    tot_c = tot_s = tot_i = tot_p = 0.0
    n_all = 0
    t_all = time.perf_counter()
    for v in range(min(V, a.views)):
        t = time.perf_counter()
        ins = image.img_hw_cubesCorner_inScopeCheck(hw_shape=imgs[v].shape[:2], img_h_cubesCorner=hc[v], img_w_cubesCorner=wc[v])
        t1 = time.perf_counter()
        c = centers[:, v, ins]
        ch, cw = np.ascontiguousarray(c[0]), np.ascontiguousarray(c[1])
        t2 = time.perf_counter()
        e = ctx.crop_embed(v, ch, cw, mean)
        t3 = time.perf_counter()
        emb_all[ins, v] = e
        t4 = time.perf_counter()
        n = int(ins.sum()); n_all += n
        tot_i += t1 - t; tot_p += t2 - t1; tot_c += t3 - t2; tot_s += t4 - t3
        print("view %2d: %6d patches  inscope %.3f  prep %.3f  C call %.3f (%.1f k/s)  scatter %.3f" % (v, n, t1 - t, t2 - t1, t3 - t2, n / (t3 - t2) / 1e3, t4 - t3), flush=True)
    wall = time.perf_counter() - t_all
    print("serial: %d patches in %.2f s = %.1f k/s; C calls alone %.1f k/s; inscope %.2f prep %.2f scatter %.2f s" % (n_all, wall, n_all / wall / 1e3, n_all / tot_c / 1e3, tot_i, tot_p, tot_s))


if __name__ == "__main__":
    main()
