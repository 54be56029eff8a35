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
