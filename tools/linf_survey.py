#!/usr/bin/env python3
"""tools/linf_survey.py — device L_inf of the surface probabilities against the fp64 oracle over MANY inputs (GPU box; TEST INFRASTRUCTURE: uses
oracle/ and tests/synth.py as the checker). The parity tests assert a tolerance on a handful of seeded cases; this prints the distribution behind it:
    python tools/linf_survey.py [--noise 12] [--scene 12] [--out gpurun_out/linf_survey.json]
noise: BN-calibrated random nets (tests/synth.py, seeds cycling 0..2) on fresh noise inputs at s = 32; scene: cubes sampled from the DTU scan9 / Middlebury
dino grids with random view pairs (noise views, partly out of view - the inputs that decided round 5's conv4 question), CVC by the device (bit-exact
against the oracle elsewhere). With SURFACENET_HIP_LIB=<test-only twin> and SN_C4_M8=0 the same inputs run the round-4 arithmetic."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--noise", type=int, default=12)
    ap.add_argument("--scene", type=int, default=12)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import surfacenet_amd
    import synth
    from oracle import net_oracle
    from surfacenet_amd import synthetic
    s = 32
    rows = []
    for i in range(a.noise):
        values = list(synth.calibrated_params(i % 3))
        X = synth.random_cvc(2, s, 1000 + i)
        with surfacenet_amd.Context(cube_D=s, max_samples=2) as ctx:
            ctx.load_param_values(values)
            _, unf = ctx.forward(X, None, n_vp=1)
        _, u64 = net_oracle.forward_torch(X, values, n_vp=1)
        rows.append(("noise %d (net %d)" % (i, i % 3), float(np.abs(unf - u64).max())))
        print("%-28s L_inf %.3e" % rows[-1], flush=True)
    values = list(synth.calibrated_params(1))
    rs = np.random.RandomState(7)
    for i in range(a.scene):
        cfg = ("dtu_scan9", "dino")[i % 2]
        P, imgs, cubes, _, _, _ = synthetic.dataset_scene(cfg, s, 600)
        pk = int(rs.randint(0, len(cubes)))
        pairs = np.stack([np.sort(rs.choice(len(imgs), 2, replace=False)) for _ in range(4)])[None].astype(np.int64)
        with surfacenet_amd.Context(cube_D=s, max_samples=4) as ctx:
            ctx.load_param_values(values); ctx.set_cameras(P); ctx.set_images(imgs)
            _, unf, cvc = ctx.cvc_forward(pairs, cubes["xyz"][pk:pk + 1], cubes["resol"][pk:pk + 1], np.full((1, 4), 0.25, np.float32), return_cvc=True)
        _, u64 = net_oracle.forward_torch(cvc, values, n_vp=1)
        inview = float((np.abs(cvc + synthetic.MEAN6[None, :, None, None, None]).reshape(cvc.shape[0], 2, 3, -1).max(axis=2) > 0).mean())
        rows.append(("scene %s cube %d (in view %.2f)" % (cfg, pk, inview), float(np.abs(unf.reshape(u64.shape) - u64).max())))
        print("%-44s L_inf %.3e" % rows[-1], flush=True)
    v = np.asarray([r[1] for r in rows])
    print("== %d cases: median %.3e  90 %% %.3e  max %.3e" % (len(v), np.median(v), np.percentile(v, 90), v.max()))
    if a.out:
        json.dump({"cases": rows, "median": float(np.median(v)), "p90": float(np.percentile(v, 90)), "max": float(v.max()),
                   "lib": os.environ.get("SURFACENET_HIP_LIB", "product"), "SN_C4_M8": os.environ.get("SN_C4_M8", "")}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
