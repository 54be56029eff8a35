#!/usr/bin/env python3
"""tools/ab_outputs.py — are the outputs of two builds of libsurfacenet_hip.so bit-identical on the same inputs?
    python tools/ab_outputs.py save out.npz            (uses SURFACENET_HIP_LIB or the in-tree library)
    python tools/ab_outputs.py cmp a.npz b.npz
Used when a kernel is re-written in a way that is meant to keep its arithmetic (e.g. the z-run upsampler)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if sys.argv[1] == "cmp":
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        for k in a.files:
            same = np.array_equal(a[k], b[k])
            print(k, "identical" if same else "DIFFER max |d| = %.3e" % np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max())
        return
    import surfacenet_amd
    from surfacenet_amd import synthetic, weights
    out = {}
    for s, n, n_vp in ((32, 3, 2), (16, 2, 3), (12, 2, 1)):
        sc = synthetic.synthetic_scene(n, n_vp, s=s, seed=s, hw=(600, 800))
        for prec in ("f16x3", "f16x3p", "f16m8", "f16"):
            with surfacenet_amd.Context(cube_D=s, max_samples=n * n_vp, precision=prec) as ctx:
                ctx.load_param_values(weights.synthetic_param_values(1))
                ctx.set_cameras(sc["cams"]); ctx.set_images(sc["imgs"])
                fused, unfused, _ = ctx.cvc_forward(sc["pairs"], sc["xyz"], sc["resol"], sc["w"])
            out["s%d_%s_unfused" % (s, prec)] = unfused
    # the 2-D form of the kernel: similarityNet embeddings of 40 patches in the three arithmetic modes
    sc = synthetic.synthetic_scene(1, 1, s=16, seed=1, hw=(300, 400))
    rs = np.random.RandomState(2)
    ch, cw = rs.uniform(0, 300, 40), rs.uniform(0, 400, 40)
    mean = np.asarray([103.939, 116.779, 123.68], dtype=np.float32)
    for prec in ("f16x3", "f16m8", "f16"):
        with surfacenet_amd.Context(cube_D=16, max_samples=2, precision=prec) as ctx:
            ctx.set_cameras(sc["cams"]); ctx.set_images(sc["imgs"])
            ctx.load_simil_param_values(weights.synthetic_simil_param_values(0))
            out["simil_%s" % prec] = ctx.crop_embed(0, ch, cw, mean)
    np.savez(sys.argv[2], **out)


if __name__ == "__main__":
    main()
