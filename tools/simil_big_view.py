import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import surfacenet_amd
from surfacenet_amd import synthetic, weights
scene = synthetic.synthetic_scene(2, 2, s=32, seed=0)
mean = np.asarray([103.939, 116.779, 123.68], dtype=np.float32)
with surfacenet_amd.Context(cube_D=32, max_samples=4) as ctx:
    ctx.set_cameras(scene["cams"]); ctx.set_images(scene["imgs"])
    ctx.load_simil_param_values(weights.synthetic_simil_param_values(0))
    rs = np.random.RandomState(1)
    for n in (2040, 2040 * 8, 126000, 126000):
        ch, cw = rs.uniform(0, 1200, n), rs.uniform(0, 1600, n)
        ctx.crop_embed(0, ch[:2040], cw[:2040], mean)
        t0 = time.perf_counter(); e = ctx.crop_embed(0, ch, cw, mean); dt = time.perf_counter() - t0
        print("n %7d  %.1f k patches/s  (%.3f s)" % (n, n / dt / 1e3, dt), flush=True)
    # sustained: 20 x 2040 back to back
    t0 = time.perf_counter()
    for _ in range(40): ctx.crop_embed(0, ch[:2040], cw[:2040], mean)
    dt = time.perf_counter() - t0
    print("40 x 2040: %.1f k patches/s" % (40 * 2040 / dt / 1e3))
    tf, ghz = ctx.mfma_probe(10.0); print("box %.0f TF %.3f GHz" % (tf, ghz))
