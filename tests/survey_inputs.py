"""Seeded input families of the L_inf survey (tools/linf_survey.py) - TEST INFRASTRUCTURE. One `Case` = (family, key) -> network values + the network input
(a CVC tensor, produced by the device's CVC warp where the case is a cube of a scene: that kernel is bit-exact against the reference-run vectors elsewhere),
so that the survey's worst cases can be replayed by name from tests/test_gpu_parity.py. Nothing here touches /root/reference.

families (Next #4): noise | structured | real (the two real-pixel windows of tests/golden/real_cases.npz, shifted by whole voxels / flipped) |
scene (cubes of the DTU scan9 / Middlebury dino grids incl. the grid's border cubes, 2 random view pairs; `scene5` / `scene16`: N_vp = 5 / 16) |
stress (the x3.2 net of tests/test_gpu_numerics.py, calibrated by the library on the case's own batch)."""
import numpy as np

import golden_util
import synth

S = 32
_MEAN = golden_util.MEAN6[None, :, None, None, None]


def _structured(kind, seed):
    s = S
    rs = np.random.RandomState(seed)
    if kind == "all_out_of_view":
        x = np.zeros((2, 6, s, s, s), np.float32)
    elif kind == "constant_colour":
        x = np.empty((2, 6, s, s, s), np.float32)
        x[:] = rs.randint(0, 256, 6).astype(np.float32)[None, :, None, None, None]
    elif kind == "step_edge":
        x = np.empty((2, 6, s, s, s), np.float32)
        a, b = rs.randint(0, 256, 6).astype(np.float32), rs.randint(0, 256, 6).astype(np.float32)
        cut = int(rs.randint(4, s - 4))
        x[:, :, :cut] = a[None, :, None, None, None]
        x[:, :, cut:] = b[None, :, None, None, None]
        x[1:] = np.swapaxes(x[1:], 2, 4)
    elif kind == "smooth_heavy_tail":
        coarse = rs.rand(2, 6, s // 4, s // 4, s // 4).astype(np.float32)
        x = np.repeat(np.repeat(np.repeat(coarse, 4, axis=2), 4, axis=3), 4, axis=4) * 120 + 60
        tail = rs.rand(*x.shape) < 0.01
        x[tail] = np.where(rs.rand(int(tail.sum())) < 0.5, 0.0, 255.0)
        x = np.rint(x).astype(np.float32)
    else:
        raise KeyError(kind)
    return x - _MEAN


STRUCTURED_KINDS = ("all_out_of_view", "constant_colour", "step_edge", "smooth_heavy_tail")
_FLIPS = ((), (2,), (3,), (4,), (2, 3), (3, 4), (2, 4), (2, 3, 4))


def case_list(n_noise=36, n_real=48, n_scene=80, n_scene5=10, n_scene16=10, n_stress=6):
    """-> list of (family, key) in a fixed order; key is everything `make_case` needs."""
    out = [("noise", (i % 3, 5000 + i)) for i in range(n_noise)]
    out += [("structured", (net, kind, 40 + net)) for net in range(3) for kind in STRUCTURED_KINDS]
    wins = ("dtu_real", "mid_real")
    for i in range(n_real):
        out.append(("real", (i % 3, wins[(i // 3) % 2], i // 6)))           # variant i // 6: shift + flip table below
    rs = np.random.RandomState(77)
    for i in range(n_scene):
        cfg = ("dtu_scan9", "dino")[i % 2]
        # a third of the cubes from the grid's first / last rows (border cubes: partly out of every view), the rest anywhere
        where = ("first", "last", "any")[i % 3] if i < 2 * n_scene // 3 else "any"
        out.append(("scene", (i % 3, cfg, where, int(rs.randint(0, 1 << 30)))))
    out += [("scene5", (i % 3, "dtu_scan9", "any", int(rs.randint(0, 1 << 30)))) for i in range(n_scene5)]
    out += [("scene16", (i % 3, "dino", "any", int(rs.randint(0, 1 << 30)))) for i in range(n_scene16)]
    out += [("stress", (i % 3, 7000 + i)) for i in range(n_stress)]
    return out


def net_values(net, stress=False):
    values = [np.array(v) for v in synth.calibrated_params(net)]
    if stress:          # BatchNorm statistics of merge_conv_a that under-estimate the spread 3.2-fold (function unchanged: same fp64 oracle)
        from oracle import net_oracle
        ix = {(l, p): i for i, (l, p, _) in enumerate(net_oracle.PARAM_LAYOUT)}
        values[ix[("merge_conv_a", "inv_std")]] = values[ix[("merge_conv_a", "inv_std")]] * np.float32(3.2)
    return values


_scene_cache = {}


def _scene(cfg):
    if cfg not in _scene_cache:
        from surfacenet_amd import synthetic
        P, imgs, cubes, _, _, _ = synthetic.dataset_scene(cfg, S, 0)
        _scene_cache[cfg] = (P, imgs, cubes)
    return _scene_cache[cfg]


def make_case(family, key, ctx_for):
    """-> (net index, stress flag, X (n_samples,6,s,s,s) float32 network input, label). `ctx_for(tag, P, imgs)` returns a Context with those cameras /
    images set (cached by the caller) - used for the CVC warp of real / scene cases."""
    if family == "noise":
        net, seed = key
        return net, False, synth.random_cvc(2, S, seed), "noise net %d seed %d" % key
    if family == "stress":
        net, seed = key
        return net, True, synth.random_cvc(4, S, seed), "x3.2 net %d seed %d (calibrated)" % key
    if family == "structured":
        net, kind, seed = key
        return net, False, _structured(kind, seed), "structured %s net %d" % (kind, net)
    if family == "real":
        net, win, var = key
        c = golden_util.real_cases()[win]
        assert int(c["s"]) == S
        rs = np.random.RandomState(900 + var)
        shift = rs.randint(-6, 7, 3).astype(np.float32) if var else np.zeros(3, np.float32)
        xyz = (np.asarray(c["xyz"], np.float32) + shift[None] * np.asarray(c["resol"], np.float32).reshape(-1, 1)).astype(np.float32)
        ctx = ctx_for(win, c["P"], golden_util.case_images(c))
        cvc = ctx.cvc(c["pairs"], xyz, c["resol"], mean=golden_util.MEAN6)
        fl = _FLIPS[var % len(_FLIPS)]
        X = np.ascontiguousarray(np.flip(cvc[:1], axis=fl)) if fl else cvc[:1]
        return net, False, X, "real %s net %d shift %s flip %s" % (win, net, shift.astype(int).tolist(), list(fl))
    if family in ("scene", "scene5", "scene16"):
        net, cfg, where, seed = key
        P, imgs, cubes = _scene(cfg)
        rs = np.random.RandomState(seed)
        n = len(cubes)
        edge = max(1, n // 50)
        pk = int(rs.randint(0, edge)) if where == "first" else (n - 1 - int(rs.randint(0, edge)) if where == "last" else int(rs.randint(0, n)))
        n_vp = {"scene": 2, "scene5": 5, "scene16": 16}[family]
        pairs = np.stack([np.sort(rs.choice(len(imgs), 2, replace=False)) for _ in range(n_vp)])[None].astype(np.int64)
        ctx = ctx_for(cfg, P, imgs)
        cvc = ctx.cvc(pairs, cubes["xyz"][pk:pk + 1], cubes["resol"][pk:pk + 1], mean=golden_util.MEAN6)
        inview = float((np.abs(cvc + _MEAN).reshape(cvc.shape[0], 2, 3, -1).max(axis=2) > 0).mean())
        return net, False, cvc, "%s %s cube %d (%s, in view %.2f) net %d" % (family, cfg, pk, where, inview, net)
    raise KeyError(family)
