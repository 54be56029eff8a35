#!/usr/bin/env python3
"""tools/simil_chunk_size.py - similarityNet per-patch layer times against the chunk size (GPU box): does a conv1_1 output that fits the 256 MB Infinity Cache
(n <= 256 patches: 268 MB) make s_conv1_2 faster? Round 6: no - s_conv1_1 0.285 us/patch at every size, s_conv1_2 0.73 -> 0.82 us/patch at n = 256: these layers are
not waiting for HBM; sub-chunking the first layers through the cache is closed."""
import sys, json
sys.path.insert(0, '/root/repo')
import numpy as np
import bench, surfacenet_amd
from surfacenet_amd import synthetic
scene = synthetic.synthetic_scene(2, 2, s=32, seed=0)
with surfacenet_amd.Context(cube_D=32, max_samples=4) as ctx:
    ctx.set_cameras(scene["cams"]); ctx.set_images(scene["imgs"])
    for n in (2040, 1024, 512, 256, 128, 2040):
        r = bench.simil_net(surfacenet_amd, ctx, scene, 6, n=n)
        k = r["kernels_ms_per_step"]
        print("n %5d  %7.0f patches/s  us/patch: " % (n, r["value"]) + "  ".join("%s %.3f" % (x[2:], 1e3 * k[x] / n) for x in ("s_conv1_1", "s_conv1_2", "s_conv2_1", "s_conv2_2", "s_conv3_2", "s_conv4_2")), flush=True)
