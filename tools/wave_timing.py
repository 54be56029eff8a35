#!/usr/bin/env python3
"""tools/wave_timing.py — where do the conv kernels' waves wait? Needs a diagnostic build of the library:
    hipcc ... -DSN_TIMING=1 (conv3d_mfma.h) -> SURFACENET_HIP_LIB=<that .so> python tools/wave_timing.py
Every wave accumulates shader-clock totals (whole kernel, the vmcnt wait in front of each per-piece barrier, the barrier itself); the
script runs the headline batch a few times and prints, per layer, the share of wave time spent in the two waits.
-DSN_TIMING=1..4 (the one-wave-per-SIMD loop, i.e. merge_conv_a/b): the two columns hold 1: {burst A, burst B}, 2: {vmcnt wait, barrier}, 3: {burst M,
whole piece} per piece, 4: per slab {slab head, piece loop}; "pieces" counts what the mode counts. -DSN_TIMING=10 (every conv kernel): per TILE {store
epilogue, K loop}. (The per-segment modes of the ping-pong loops and the sub-piece modes of the round-2 loop were removed with the round-5 pruning; their
results are in profiles/r2 .. r4/README.md.) A stamp costs ~40 clocks and its lgkmcnt(0) also waits for the wave's in-flight prefetch reads:
instrumented builds run 10-25 % slower and shift time between neighbouring columns.
--simil: the similarityNet's layers instead of the SurfaceNet's."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import surfacenet_amd
    from surfacenet_amd import _lib, synthetic, weights
    s, n, n_vp, steps = 32, 64, 2, 5
    sc = synthetic.synthetic_scene(n, n_vp, s=s, seed=0)
    with surfacenet_amd.Context(cube_D=s, max_samples=n * n_vp) as ctx:
        ctx.load_param_values(weights.synthetic_param_values(0))
        ctx.set_cameras(sc["cams"]); ctx.set_images(sc["imgs"])
        d = [ctx.upload(sc[k]) for k in ("pairs", "xyz", "resol", "w")]
        d_fused = ctx.dev_alloc(n * s ** 3 * 4)
        lib = _lib.load()
        fn = lib.sn_debug_timing
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        out = np.zeros((32, 4), dtype=np.uint64)
        names = ctypes.create_string_buffer(2048)
        ctx.cvc_forward_dev(n, n_vp, d[0], d[1], d[2], d[3], d_fused)
        fn(ctx._h, out.ctypes.data_as(ctypes.c_void_p), 32, names, 2048)          # warm-up pass: read and clear
        for _ in range(steps):
            ctx.cvc_forward_dev(n, n_vp, d[0], d[1], d[2], d[3], d_fused)
        fn(ctx._h, out.ctypes.data_as(ctypes.c_void_p), 32, names, 2048)
        if "--simil" in sys.argv:                   # the similarityNet's 2-D conv layers (names beyond the 31st share status bit 31 = slot 31)
            values = weights.synthetic_simil_param_values(0)
            ctx.load_simil_param_values(values)
            mean = np.asarray([103.939, 116.779, 123.68], dtype=np.float32)
            rs = np.random.RandomState(1)
            H, W = sc["imgs"][0].shape[:2]
            ch, cw = rs.uniform(0, H, 2040), rs.uniform(0, W, 2040)
            ctx.crop_embed(0, ch, cw, mean)
            fn(ctx._h, out.ctypes.data_as(ctypes.c_void_p), 32, names, 2048)
            for _ in range(steps):
                ctx.crop_embed(0, ch, cw, mean)
            fn(ctx._h, out.ctypes.data_as(ctypes.c_void_p), 32, names, 2048)
            nms = [x for x in names.value.decode().split(",") if x]
            for i in range(32):
                k, vm, bar, pieces = (float(v) for v in out[i])
                if k > 0:
                    print("%-14s wave-cycles %.3e  column-1 %5.1f %%  column-2 %5.1f %%  pieces %.0f  cycles/piece %.0f  (per piece: %.0f, %.0f)"
                          % (nms[i] if i < len(nms) else "slot %d (shared)" % i, k, 100 * vm / k, 100 * bar / k, pieces, k / max(pieces, 1), vm / max(pieces, 1), bar / max(pieces, 1)))
            return
        for i, nm in enumerate([x for x in names.value.decode().split(",") if x]):
            k, vm, bar, pieces = (float(v) for v in out[i])
            if k > 0:
                print("%-14s wave-cycles %.3e  vmcnt-wait %5.1f %%  barrier-wait %5.1f %%  pieces/wave-launch %.0f  cycles/piece %.0f  (wait/piece: vm %.0f, barrier %.0f)"
                      % (nm, k, 100 * vm / k, 100 * bar / k, pieces / (steps * 256 * 8), k / max(pieces, 1), vm / max(pieces, 1), bar / max(pieces, 1)))


        tr = np.zeros((2048, 8, 2), dtype=np.int64)
        lib.sn_debug_trace.restype = ctypes.c_int
        lib.sn_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.sn_debug_trace(ctx._h, tr.ctypes.data_as(ctypes.c_void_p))
        if not tr.any():
            return                                           # (builds without the per-piece trace, e.g. the ping-pong kernels)
        tr = tr[200:1800]                                   # steady state of workgroup 0 in merge_conv_b (EPI_FINAL)
        arr, rel = tr[:, :, 0], tr[:, :, 1]
        piece = np.diff(rel.max(axis=1)).astype(np.float64)                     # release-to-release = piece time
        last = arr.max(axis=1, keepdims=True)
        print("merge_conv_b workgroup 0: piece time %.0f +- %.0f cycles" % (piece.mean(), piece.std()))
        print("  arrival before the last wave, per wave (cycles, mean):", np.round((last - arr).mean(axis=0)).astype(int).tolist())
        print("  release - last arrival: %.0f" % (rel.min(axis=1) - last[:, 0]).mean())
        for pairing in ("w,w+4", "w,w+1"):
            if pairing == "w,w+4":
                pa = np.maximum(arr[:, :4], arr[:, 4:])
            else:
                pa = np.maximum(arr[:, 0::2], arr[:, 1::2])
            idle = (last - pa).mean(axis=0)
            print("  if SIMD pairs are (%s): SIMD idle before the last arrival (cycles, mean) %s -> %.1f %% of the piece" % (pairing, np.round(idle).astype(int).tolist(), 100 * idle.mean() / piece.mean()))
        order = np.argsort(arr, axis=1)
        print("  how often each wave arrives last:", np.bincount(order[:, -1], minlength=8).tolist(), " first:", np.bincount(order[:, 0], minlength=8).tolist())


if __name__ == "__main__":
    main()
