"""CPU: host-side logic of the product (weights, batching, sharding, drop-in argument handling) — no compute calls."""
import os
import pickle

import numpy as np
import pytest

import golden_util
from surfacenet_amd import reconstruct, weights


def test_batch_selectors_match_reference_goldens():
    z = np.load(os.path.join(golden_util.GOLDEN, "batch_cases.npz"))
    for i in range(5):
        sel = reconstruct.gen_non0Batch_npBool(z["c%d/ind" % i], int(z["c%d/bs" % i]))
        assert np.array_equal(sel, z["c%d/sel" % i])
    assert reconstruct.gen_non0Batch_npBool(np.zeros(5, bool), 3).shape[0] == 0       # "Empty!" case, main_reconstruct.py:128


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 64, 65, 1000):
        for world in (1, 2, 3, 8):
            spans = [reconstruct.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) == -(-n // world) if n else True


def test_weight_blob_and_pickle_roundtrip(tmp_path):
    vals = weights.synthetic_param_values(5)
    assert len(vals) == 105
    blob, descs = weights.to_blob(vals)
    assert blob.dtype == np.float32 and blob.size == sum(v.size for v in vals)
    for d, v in zip(descs, vals):
        assert tuple(d.shape[: d.ndim]) == v.shape
        assert np.array_equal(blob[d.offset: d.offset + v.size].reshape(v.shape), v)
    p = tmp_path / "net.model"
    with open(p, "wb") as f:
        pickle.dump([np.asarray(v) for v in vals], f, protocol=2)      # the reference writes a py2 pickle of a flat list
    back = weights.load_lasagne_pickle(str(p))
    assert all(np.array_equal(a, b) for a, b in zip(back, vals))
    with pytest.raises(ValueError):
        weights.validate(vals[:50])
    bad = list(vals); bad[0] = bad[0][:, :5]
    with pytest.raises(ValueError):
        weights.validate(bad)


def test_preprocess_augmentation_contract():
    from surfacenet_amd import CVC
    c = golden_util.cvc_cases()["dtu_s8_vp1"]
    raw = c["out_u8"].astype(np.float32)
    gt, out = CVC.preprocess_augmentation(None, raw, golden_util.MEAN6[None, :, None, None, None], augment_ON=False, crop_ON=False)
    assert gt is None and np.array_equal(out, c["pre_f32"]) and out.flags.writeable and out is not raw
    out += golden_util.MEAN6[None, :, None, None, None]                # the caller's in-place add (main_reconstruct.py:150)
    with pytest.raises(NotImplementedError):
        CVC.preprocess_augmentation(None, raw, golden_util.MEAN6[None, :, None, None, None])


def test_synthetic_scene_is_in_scope():
    from oracle import cvc_oracle
    sc = golden_util.synthetic_scene(4, 2, s=8, seed=0)
    out = cvc_oracle.gen_coloredCubes(sc["pairs"], sc["xyz"], sc["resol"], sc["cams"], sc["imgs"], 8)
    assert (out.reshape(8, 2, 3, -1).max(axis=2) > 0).mean() > 0.99    # all voxels in scope


def test_cube_grid_matches_reference_initializeCubes():
    """synthetic.cube_grid (the cube table the scene benches / tests feed the hot path) against tables produced by the reference's
    own scene.initializeCubes (oracle/gen_golden_scene.py -> tests/golden/scene_cases.npz): DTU scan9 at s=32 (195,360 cubes =
    BASELINE config 3) and s=64 (24,420, the count in the reference's log q.log/inference.0000022:116), Middlebury dino, doctest input."""
    import os
    from surfacenet_amd import synthetic
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "scene_cases.npz"))
    cases = (("scan9_s32", (np.float32(0.4), 32, 26, 0.5, G["scan9_BB"])), ("scan9_s64", (np.float32(0.4), 64, 52, 0.5, G["scan9_BB"])),
             ("dino_s32", (np.float32(0.00025), 32, 26, 0.5, G["dino_BB"])), ("doc", (1, 22, 10, 0.5, G["doc_BB"])))
    for pre, args in cases:
        cubes, dmm = synthetic.cube_grid(*args)
        idx = G[pre + "_idx"]
        assert cubes.dtype == synthetic.CUBE_DTYPE and cubes.shape[0] == int(G[pre + "_n"]) and float(dmm) == float(G[pre + "_cube_D_mm"])
        assert np.array_equal(cubes["ijk"].max(axis=0) + 1, G[pre + "_grid"])
        assert np.array_equal(cubes["xyz"][idx], G[pre + "_xyz"]) and np.array_equal(cubes["ijk"][idx], G[pre + "_ijk"])
        assert np.array_equal(cubes["resol"][idx], G[pre + "_resol"])
        w = np.arange(1, cubes.shape[0] + 1, dtype=np.float64)                       # checksums over ALL rows
        assert np.array_equal(cubes["xyz"].astype(np.float64).sum(axis=0), G[pre + "_xyz_sum"])
        assert np.array_equal((cubes["xyz"].astype(np.float64) * w[:, None]).sum(axis=0), G[pre + "_xyz_wsum"])
    assert int(G["scan9_s32_n"]) == 195360 and int(G["scan9_s64_n"]) == 24420
    assert G["P_dtu49"].shape == (49, 3, 4) and G["P_mid16"].shape == (16, 3, 4)
    assert np.array_equal(G["P_dtu49"][:2], synthetic.P_DTU_12)


def test_genuine_python2_pickle_weight_file(tmp_path):
    """weights.load_lasagne_pickle on the byte format the reference's `.model` files really have (Python 2.7 cPickle protocol 2,
    numpy 1.13: `numpy.core.multiarray._reconstruct`, py2 `str` payloads; nets/SurfaceNet.py:397-400) - tests/py2pickle.py emits
    that stream opcode by opcode. A Python-3 `pickle.load` without encoding='latin1' cannot read it."""
    import py2pickle
    vals = weights.synthetic_param_values(7)
    blob = py2pickle.dumps_py2(vals)
    assert b"numpy.core.multiarray\n_reconstruct" in blob and blob[:2] == b"\x80\x02"
    with pytest.raises(UnicodeDecodeError):
        pickle.loads(blob)
    p = tmp_path / "2D_2_3D-19-0.918_0.951.model"                          # params.py:106 file name
    p.write_bytes(blob)
    back = weights.load_lasagne_pickle(str(p))
    assert len(back) == 105 and all(b.dtype == np.float32 and np.array_equal(a, b) for a, b in zip(vals, back))
    sv = weights.synthetic_simil_param_values(2)
    q = tmp_path / "epoch33_acc_tr0.707_val0.791.model"                     # params.py:91
    q.write_bytes(py2pickle.dumps_py2(sv))
    back = weights.load_simil_pickle(str(q))                                # same format (nets/similarityNet.py:240-242)
    assert len(back) == 30 and all(np.array_equal(a, b) for a, b in zip(sv, back))


def test_python2_protocol0_weight_file_and_bn_order_guard(tmp_path):
    """The other format a Python-2 `pickle.dump(values, f)` writes - protocol 0, the ASCII form, the only one that survives the
    reference's text-mode open (nets/SurfaceNet.py:398 `open(model_file)`) on every platform - with the array bytes as repr()-escaped
    py2 strings; and the one check that can tell a layer's four same-shaped BatchNorm vectors apart: inv_std > 0."""
    import py2pickle
    vals = weights.synthetic_param_values(9)
    blob = py2pickle.dumps_py2_proto0(vals)
    assert blob[:3] == b"(lp" and b"cnumpy.core.multiarray\n_reconstruct" in blob and max(blob) < 0x80      # pure ASCII
    with pytest.raises(UnicodeDecodeError):
        pickle.loads(blob)
    p = tmp_path / "proto0.model"
    p.write_bytes(blob)
    back = weights.load_lasagne_pickle(str(p))
    assert len(back) == 105 and all(b.dtype == np.float32 and np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(vals, back))
    sv = weights.synthetic_simil_param_values(3)
    q = tmp_path / "simil_proto0.model"
    q.write_bytes(py2pickle.dumps_py2_proto0(sv))
    assert all(np.array_equal(a, b) for a, b in zip(sv, weights.load_simil_pickle(str(q))))
    # a file whose BatchNorm vectors are in another order (mean <-> inv_std) has every shape right: the positivity of inv_std catches it
    idx = {(l, n): i for i, (l, n, _) in enumerate(weights.PARAM_LAYOUT)}
    swapped = list(vals)
    i, j = idx[("conv2_2", "mean")], idx[("conv2_2", "inv_std")]
    swapped[i], swapped[j] = swapped[j], swapped[i]
    assert (swapped[j] < 0).any()
    p.write_bytes(py2pickle.dumps_py2(swapped))
    with pytest.raises(ValueError, match="inv_std"):
        weights.load_lasagne_pickle(str(p))
    p.write_bytes(py2pickle.dumps_py2({"not": "a list"}.keys() and [vals[0]]))
    with pytest.raises(ValueError):
        weights.load_lasagne_pickle(str(p))


def test_interpolation_kernel_pinned_to_reference_W_5D():
    """The fixed stencil of Bilinear_3DInterpolation is the one CNN constant that can be pinned: oracle/gen_golden_w5d.py executed
    the reference's own `__W_5D__` (nets/layers.py:361-372) -> tests/golden/w5d_cases.npz. The product's weight-file entry, the
    oracle's restatement and (GPU test) the closed form inside upsample3_cat_kernel must all equal it."""
    import os
    from oracle import net_oracle
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "w5d_cases.npz"))
    for f, k in ((2, 3), (4, 5)):
        W = G["f%d_W" % f]
        assert int(G["f%d_k" % f]) == k and W.shape == (1, 1, k, k, k) and W.dtype == np.float32
        assert np.array_equal(weights.interpolation_kernel(k), W)
        w1 = net_oracle.w5d(k)
        assert np.array_equal((w1[:, None, None] * w1[None, :, None] * w1[None, None, :]).astype(np.float32), W[0, 0])
    vals = weights.synthetic_param_values(0)
    ups = [v for (layer, p, shape), v in zip(weights.PARAM_LAYOUT, vals) if layer.endswith("_deconv")]
    assert [u.shape[2] for u in ups] == [3, 5, 5] and np.array_equal(ups[0], G["f2_W"]) and np.array_equal(ups[1], G["f4_W"])


class _FakeNumericsCtx(object):
    """Stand-in of a Context for the NumericsGuard's bookkeeping (no library call): `status` is what numeric_status() returns next, `sat` the saturated
    fraction a measure-only calibrate reports."""
    precision = "f16x3"

    def __init__(self):
        from surfacenet_amd.context import Context
        self._numerics = Context._fresh_numerics()
        self._fresh = Context._fresh_numerics
        self.status, self.sat, self.calls = [], 0.0, []

    def numeric_status(self):
        self.calls.append("status")
        return list(self.status)

    def calibrate(self, n, frac):
        self.calls.append("measure" if frac < 0 else "calibrate")
        rep = dict(s_act_before=0, s_cat_before=2, s_act=(0 if frac < 0 else -2), s_cat=2, sat_act_before=self.sat, sat_cat_before=0.0, sat_act=0.0, sat_cat=0.0,
                   max_act=20.0, max_cat=1.0)
        if frac >= 0:
            self._numerics["calibrated"], self._numerics["report"] = True, rep
        return rep

    def load_param_values(self):
        self._numerics = self._fresh()


def test_numerics_guard_state_lives_on_the_context():
    """the guard's knowledge (calibrated / report / back-off) belongs to the Context whose exponents it describes - contexts are cached and
    shared (runtime._contexts): two callers see ONE calibration, new weights reset it for every live guard, a clean-but-noisy net is probed more and
    more sparsely, fp8-plane layers (nothing to calibrate) get one warning of their own."""
    import warnings
    from surfacenet_amd.context import NumericsGuard
    ctx = _FakeNumericsCtx()
    a, b = NumericsGuard(ctx), NumericsGuard(ctx)
    assert a.check() is None and ctx.calls == ["status"]                       # clean word: one read-back, nothing else
    ctx.status, ctx.sat = ["merge_conv_a"], 0.05
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        rep = a.check("loop")
        assert rep is not None and rep["s_act"] == -2 and len(w) == 1 and "recalibrated" in str(w[0].message)
        assert a.calibrated and b.calibrated and b.report is rep               # the other caller of the context knows
        assert b.check() is None and a.check() is None and len(w) == 1         # calibrated: the word is read and cleared, nothing more
    ctx.load_param_values()                                                    # new weights: static exponents again ...
    assert not a.calibrated and not b.calibrated and a.report is None          # ... and every live guard watches them afresh
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert b.check() is not None and len(w) == 1
    # a net that saturates a few per mille of its codes raises the word on every batch: measure-only probes, then back-off
    ctx.load_param_values()
    ctx.sat, ctx.calls = 0.002, []
    for _ in range(3 + 2 * NumericsGuard.BACKOFF_EVERY):
        assert a.check() is None
    assert ctx.calls.count("measure") == 3 + 2 and ctx.calls.count("calibrate") == 0
    ctx.sat = 0.2                                                              # it gets worse later: the sparse probe still catches it
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = [a.check() for _ in range(NumericsGuard.BACKOFF_EVERY)]
        assert sum(g is not None for g in got) == 1 and len(w) == 1
    # fp8 planes: warned once per set of weights, never calibrated
    ctx.load_param_values()
    ctx.status, ctx.calls = ["conv4_1", "conv3_3"], []
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert a.check() is None and b.check() is None
        assert len(w) == 1 and "fp8" in str(w[0].message) and "conv4_fp8=False" in str(w[0].message)
    assert "measure" not in ctx.calls and "calibrate" not in ctx.calls
    assert NumericsGuard(ctx, enabled=False).check() is None


def test_argmaxN_fast_path_equals_the_reference_expression_incl_ties():
    """viewPairSelection.__argmaxN_viewPairs__ (utils/viewPairSelection.py:8-41) selects by argpartition since round 6; wherever the reference's
    `w.argsort(axis=1)[:, -N:]` could break a tie its own way (equal weights inside the selection or at its boundary, NaN) the row is redone by
    that very expression: results identical on random, heavily tied, constant and NaN-bearing weights."""
    from surfacenet_amd import viewPairSelection as V
    pairs = V.k_combination_np(range(49), 2)
    rs = np.random.RandomState(3)

    def ref(w, N):
        ic, _ = np.indices((w.shape[0], N))
        idx = w.argsort(axis=1)[:, -N:]
        return pairs[idx], w[ic, idx]
    for trial in range(6):
        w = rs.rand(400, 1176).astype(np.float32)
        if trial == 1:
            w[:, 100:140] = w[:, :1]
        if trial == 2:
            w = np.round(w, 2)
        if trial == 3:
            w[5, 7] = np.nan; w[9, :] = 0.5
        if trial == 4:
            w[:] = 0.25
        for N in (5, 16):
            a, b = V.__argmaxN_viewPairs__(pairs, w, N), ref(w, N)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True), (trial, N)
    w = rs.rand(7, 12).astype(np.float32)               # few pairs: the reference's expression itself
    a, b = V.__argmaxN_viewPairs__(V.k_combination_np(range(5), 2)[:12] if False else np.arange(24).reshape(12, 2), w, 5), None
    assert a[0].shape == (7, 5, 2) and np.all(np.diff(a[1], axis=1) >= 0)
