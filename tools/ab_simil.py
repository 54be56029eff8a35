#!/usr/bin/env python3
"""tools/ab_simil.py — same-box interleaved A/B of library builds on the similarityNet leg alone (bench.py -> similarity_net: crop + preprocess + embedding of 2,040
patches per step): patches/s, every s_conv* kernel's ms, and the embeddings of each variant against the first one's.
    python tools/ab_simil.py [--rounds 3] [--steps 6] NAME ...        (NAME: a directory under gpurun_abl/, or `tree`)"""
import argparse
import json
import os
import statistics
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(steps, out):
    import bench
    import surfacenet_amd
    from surfacenet_amd import synthetic, weights
    scene = synthetic.synthetic_scene(2, 2, s=32, seed=0)
    with surfacenet_amd.Context(cube_D=32, max_samples=4) as ctx:
        ctx.set_cameras(scene["cams"]); ctx.set_images(scene["imgs"])
        r = bench.simil_net(surfacenet_amd, ctx, scene, steps)
        ctx.load_simil_param_values(weights.synthetic_simil_param_values(0))
        rs = np.random.RandomState(3)
        emb = ctx.crop_embed(0, rs.uniform(0, 1200, 300), rs.uniform(0, 1600, 300), np.asarray([103.939, 116.779, 123.68], np.float32))
    np.save(out, emb)
    print(json.dumps(r))


def main():
    if sys.argv[1] == "--child":
        return child(int(sys.argv[2]), sys.argv[3])
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    libs = [(v, os.path.join(ROOT, "surfacenet_amd", "libsurfacenet_hip.so") if v == "tree" else os.path.join(ROOT, "gpurun_abl", v, "libsurfacenet_hip.so")) for v in a.variants]
    res = {v: [] for v, _ in libs}
    embs = {}
    for r in range(a.rounds):
        for v, lib in libs:
            out = "/tmp/ab_simil_%s.npy" % v
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(a.steps), out], env=dict(os.environ, SURFACENET_HIP_LIB=lib), capture_output=True, text=True)
            if p.returncode != 0:
                print("%s failed:\n%s" % (v, p.stderr[-1500:]), flush=True)
                continue
            j = json.loads(p.stdout.strip().splitlines()[-1])
            res[v].append(j)
            embs[v] = np.load(out)
            k = j["kernels_ms_per_step"]
            print("round %d %-10s %8.0f patches/s  convs %.3f ms  %s  (vs oracle %.2e)" % (r, v, j["value"], j["convs_ms_per_step"],
                  " ".join("%s %.3f" % (x[2:], k[x]) for x in sorted(k) if x.startswith("s_conv")), j["check_Linf_vs_oracle_f32"]), flush=True)
    print("== medians")
    first = a.variants[0]
    for v, _ in libs:
        if not res[v]:
            continue
        ks = sorted(x for x in res[v][0]["kernels_ms_per_step"] if x.startswith("s_conv"))
        same = "" if v == first or first not in embs or v not in embs else ("  embeddings vs %s: %s" % (first, "identical" if np.array_equal(embs[v], embs[first]) else "max |d| %.2e" % np.abs(embs[v] - embs[first]).max()))
        print("%-10s %8.0f patches/s  convs %.3f ms  %s%s" % (v, statistics.median(j["value"] for j in res[v]), statistics.median(j["convs_ms_per_step"] for j in res[v]),
              " ".join("%s %.3f" % (x[2:], statistics.median(j["kernels_ms_per_step"][x] for j in res[v])) for x in ks), same), flush=True)


if __name__ == "__main__":
    main()
