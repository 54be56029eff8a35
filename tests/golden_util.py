"""Shared helpers of the test-suite: golden fixtures (tests/golden, produced by oracle/gen_golden.py from the
reference's own code) and the synthetic inputs of SURVEY §8(d)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MEAN6 = np.asarray([123.68, 116.779, 103.939, 123.68, 116.779, 103.939]).astype(np.float32)


def synth_image(seed, H, W):
    return np.random.RandomState(int(seed)).randint(0, 256, (int(H), int(W), 3)).astype(np.uint8)


def cvc_cases():
    z = np.load(os.path.join(GOLDEN, "cvc_cases.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    out = {}
    for n in names:
        out[n] = {k.split("/")[1]: z[k] for k in z.files if k.startswith(n + "/")}
    return out


def case_images(case):
    if "imgs" in case:                    # real-pixel cases (real_cases.npz) carry their decoded windows
        return [np.ascontiguousarray(im) for im in case["imgs"]]
    H, W = case["HW"]
    return [synth_image(sd, H, W) for sd in case["seeds"]]


def real_cases():
    """CVC golden cases on windows of real DTU scan9 / Middlebury dino views (oracle/gen_golden_real.py)."""
    z = np.load(os.path.join(GOLDEN, "real_cases.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return {n: {k.split("/")[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}


def cameras():
    return np.load(os.path.join(GOLDEN, "cameras.npz"))


def synthetic_scene(n, n_vp=2, s=32, seed=0, hw=(1200, 1600)):
    """SURVEY §8(d) synthetic workload (surfacenet_amd/synthetic.py); the cameras are the reference-read DTU pair of the fixture."""
    from surfacenet_amd import synthetic
    sc = synthetic.synthetic_scene(n, n_vp, s=s, seed=seed, hw=hw)
    assert np.array_equal(sc["cams"], cameras()["P_dtu"][:2])          # the literals in synthetic.py ARE pos_001/002.txt
    return sc


# Exactly-rounded stand-ins for the two network callables of utils/earlyRejection.py, used when the reference's host
# logic was recorded (oracle/gen_golden_simil.py) and when it is replayed (tests/test_oracle_simil.py): float64 sums of
# float32 values spanning < 2^29 are exact in any order, fsum is exactly rounded, IEEE division is correctly rounded.
def toy_embedding(patches):
    x = np.asarray(patches, dtype=np.float64).reshape(patches.shape[0], 128, -1)      # (n, 128, 96)
    return (x.sum(axis=2) / 1000.0).astype(np.float32)


def toy_pair_simil(emb_pairs):
    import math
    e = np.asarray(emb_pairs, dtype=np.float64).reshape(-1, 2, emb_pairs.shape[-1])
    d = np.asarray([math.fsum(np.abs(a - b)) for a, b in e])
    return (d / (d + 300.0)).astype(np.float32)[:, None]


def cvc_config_cases():
    """CVC cases AT THE SIZES OF BASELINE.json's configs (oracle/gen_golden_configs.py ran the reference's CVC.py on them): name -> (scene dict, s,
    digest dict: sha256, idx, val, chan_sum, shape, inscope)."""
    import hashlib
    from surfacenet_amd import synthetic
    z = np.load(os.path.join(GOLDEN, "cvc_config_cases.npz"))
    out = {}
    for name in sorted({k.split("/")[0] for k in z.files}):
        c = {k.split("/")[1]: z[k] for k in z.files if k.startswith(name + "/")}
        s = int(c["s"])
        if name == "edge_s64":
            sc = synthetic.synthetic_scene(2, 3, s=64, seed=5)
            sc["xyz"], sc["resol"] = c["xyz"], c["resol"]
            assert np.array_equal(sc["pairs"], c["pairs"])
        else:
            sc = synthetic.synthetic_scene(int(c["n"]), int(c["n_vp"]), s=s, seed=int(c["seed"]))
        out[name] = (sc, s, c)
    return out


def check_cvc_digest(out_f32, c):
    """out_f32: a (N, 6, s, s, s) float32 CVC tensor (raw colours) vs the digest of the reference's output."""
    import hashlib
    assert out_f32.dtype == np.float32 and tuple(out_f32.shape) == tuple(int(v) for v in c["shape"])
    u8 = out_f32.astype(np.uint8)
    assert np.array_equal(u8.astype(np.float32), out_f32)                            # integer-valued 0..255
    assert np.array_equal(u8.reshape(-1)[c["idx"]], c["val"])
    assert np.array_equal(u8.reshape(u8.shape[0], 6, -1).sum(axis=2, dtype=np.int64), c["chan_sum"])
    assert hashlib.sha256(np.ascontiguousarray(u8).tobytes()).digest() == bytes(c["sha256"])      # every voxel of every sample
