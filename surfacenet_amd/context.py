"""`Context`: one GPU's SurfaceNet inference state (images, cameras, weights, workspace) over the C ABI."""
import ctypes

import numpy as np

from . import _lib, weights as _weights

MEAN_CVC_RGBRGB = np.asarray([123.68, 116.779, 103.939, 123.68, 116.779, 103.939]).astype(np.float32)  # params.py:129


def allgatherv_two_step(n_local, counts_fn, payload_fn):
    """The rank-uniform protocol of `Context.allgatherv_bytes`, separated from the C calls so that the multi-rank CPU tests can drive it over gloo:
    `counts_fn(n_local)` -> the per-rank byte counts (one collective), `payload_fn(n_local, total)` -> (counts, the `total` gathered bytes) (the
    collectives of sn_allgatherv_bytes_dev). Every rank calls the two in this order exactly once: the total is the same number everywhere, so the
    sizing of the destination can never send one rank back into a collective its peers have left."""
    counts = counts_fn(n_local)
    total = int(sum(counts))
    counts2, out = payload_fn(n_local, total)
    if list(counts2) != list(counts):
        raise _lib.SurfaceNetHipError("allgatherv: the ranks' counts changed between the two steps (%r -> %r): mismatched collectives" % (counts, counts2))
    ends = np.cumsum(counts)
    return [out[e - c: e] for c, e in zip(counts, ends)]


class NumericsGuard(object):
    """The numerics safety net of the default mode for ONE caller (a drop-in callable, a cube loop, a scene): the two merge layers read their inputs
    as fp16 + 6-bit codes under per-tensor premultipliers that are static - sized from the BatchNorm parameters a trained net obeys. A net that does
    not obey them saturates codes (each such value loses its own correction term); the library raises a WARNING word for that (`numeric_status`) and
    can derive the premultipliers from data instead (`calibrate`). `check()` - called by the caller right after its first batch, then once per
    scene / per call - reads the word; when more than `trigger_fraction` of a code plane's values saturate it calibrates on the batch that was just run, emits ONE `RuntimeWarning`
    naming the layers and the exponents chosen, and returns the calibration report: the caller then REDOES that batch (nets/SurfaceNet.py:385-402
    is where the only weights that will ever matter are loaded; main_reconstruct.py:145-146 is the call this protects). Afterwards the word is still
    read (and cleared) but a calibrated context reports nothing more: the calibration itself tolerates `max_sat_fraction` of saturated values.

    What the guard KNOWS lives on the Context, next to the exponents it describes (`Context._numerics`): contexts are cached and shared
    (runtime._contexts), `load_param_values` resets the exponents to the static ones - and with them this state, so a reused loop / callable watches
    new weights afresh; two callers of one context see one calibration (a caller with auto_calibrate=False does not calibrate, but runs under whatever
    exponents another caller of the same context set - `ctx.calibration_report()` says which). Cost: while the net stays clean the word is zero and a
    check is one 8-byte read-back; a net that saturates a few per mille of its codes (every uncalibrated random net does) raises the word on every
    batch, so after `BACKOFF_AFTER` measure-only probes below the trigger in a row the guard re-probes only every `BACKOFF_EVERY`-th check."""
    FP8_PLANES = frozenset(("conv3_3", "conv4_1", "conv4_2"))      # tensors of the conv4 chain: fp8 e4m3 code planes (hi codes to 448 * 2^-s, lo codes safe to 256 * 2^-s: where the warning starts; nothing to calibrate)
    BACKOFF_AFTER, BACKOFF_EVERY = 3, 64

    def __init__(self, ctx, enabled=True, max_sat_fraction=1e-3, trigger_fraction=1e-2):
        self.ctx, self.enabled, self.max_sat_fraction, self.trigger_fraction = ctx, bool(enabled), float(max_sat_fraction), float(trigger_fraction)
        self.checks = 0

    # (state shared by every guard of the context; reset by Context.load_param_values)
    @property
    def _st(self):
        st = getattr(self.ctx, "_numerics", None)
        if st is None:
            st = self.ctx._numerics = Context._fresh_numerics()
        return st

    @property
    def calibrated(self):
        return self._st["calibrated"]

    @property
    def report(self):
        return self._st["report"]

    def check(self, where="forward"):
        if not self.enabled:
            return None
        self.checks += 1
        st = self._st
        names = self.ctx.numeric_status()
        if not names or st["calibrated"]:
            return None
        import warnings
        fp8 = [n for n in names if n in self.FP8_PLANES]
        if fp8 and not st["fp8_warned"]:
            st["fp8_warned"] = True
            warnings.warn("surfacenet_amd (%s): stored activations of %s exceed 256 * 2^-s: the fp8 correction codes of the dilated chain conv4_1 .. conv4_3 "
                          "saturate for those values (each loses its own correction term; nothing to calibrate - fp8 has the range, the values are outliers of "
                          "the net's BatchNorm statistics). Context(conv4_fp8=False) keeps the chain on three fp16 MFMAs." % (where, ", ".join(fp8)),
                          RuntimeWarning, stacklevel=3)
        names = [n for n in names if n not in self.FP8_PLANES]
        if not names:
            return None
        if self.ctx.precision != "f16x3":
            if not st["mode_warned"]:
                st["mode_warned"] = True
                warnings.warn("surfacenet_amd (%s): stored activations of %s exceed the range of their 6-bit code planes; precision mode %r has no "
                              "data-driven premultipliers - accuracy degrades towards plain fp16 for those values" % (where, ", ".join(names), self.ctx.precision),
                              RuntimeWarning, stacklevel=3)
            return None
        # The warning word fires at the FIRST saturated value. Nets that roughly obey their BatchNorm statistics saturate a few per mille of
        # merge_conv_a's outputs (measured: 0.1 .. 0.3 % on uncalibrated random nets, 0.22 % on the worst structured input - L_inf unchanged at 5e-5);
        # the guard acts when more than `trigger_fraction` (1 %) of a plane's non-zero values saturate (x3.2 stress net: 5 %, L_inf 2e-4).
        if st["clean_probes"] >= self.BACKOFF_AFTER:
            st["skipped"] += 1
            if st["skipped"] % self.BACKOFF_EVERY:
                return None
        probe = self.ctx.calibrate(0, -1.0)
        st["probes"] += 1
        if max(probe["sat_act_before"], probe["sat_cat_before"]) < self.trigger_fraction:
            st["clean_probes"] += 1              # nothing to redo, nothing to report - the watch goes on (more and more sparsely)
            return None
        st["clean_probes"] = 0
        cal = self.ctx.calibrate(0, self.max_sat_fraction)
        st["calibrated"], st["report"] = True, cal        # one calibration, one warning per set of weights
        warnings.warn("surfacenet_amd (%s): stored activations of %s exceeded the range of their 6-bit code planes (%.2f %% of merge_conv_a's non-zero "
                      "outputs, %.2f %% of the concat buffer; the network's BatchNorm statistics under-estimate their spread). Premultipliers "
                      "recalibrated on this batch: s_act %d -> %d, s_cat %d -> %d (saturated fraction now %.3f %% / %.3f %%); the batch is recomputed. "
                      "Pass auto_calibrate=False to keep the static exponents."
                      % (where, ", ".join(names), 100 * cal["sat_act_before"], 100 * cal["sat_cat_before"], cal["s_act_before"], cal["s_act"],
                         cal["s_cat_before"], cal["s_cat"], 100 * cal["sat_act"], 100 * cal["sat_cat"]), RuntimeWarning, stacklevel=3)
        return cal


class Context(object):
    PRECISIONS = {"f16": 0, "f16x3": 1, "f16m8": 2, "f16x3p": 3}      # f16x3p: f16x3 without the MX tail (see surfacenet_hip.h)

    def __init__(self, cube_D=32, max_samples=64, device=0, precision="f16x3", conv4_fp8=True):
        """precision: "f16x3" (default; fp32-class results, operands as hi+lo fp16 pairs, 3 MFMAs per term) or
        "f16" (3x faster, L_inf ~2e-3 vs the fp64 oracle on BN-normalised nets: above the 1e-3 parity bar).
        conv4_fp8 (default mode only; sn_set_conv4_fp8): True / 1 = conv4_1 .. conv4_3 with their correction terms on the fp8 MX MFMA (default),
        2 = conv4_2 and conv4_3 only, False / 0 = the whole chain back on three fp16 MFMAs."""
        if precision not in self.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(self.PRECISIONS))
        self._lib = _lib.load()
        self.cube_D, self.max_samples, self.device = int(cube_D), int(max_samples), int(device)
        self._h = self._lib.sn_create(self.device, self.cube_D, self.max_samples)
        if not self._h:
            raise _lib.SurfaceNetHipError("sn_create failed: %s" % _lib.last_error())
        self.precision = precision
        _lib.check(self._lib.sn_set_precision(self._h, self.PRECISIONS[precision]))
        if int(conv4_fp8) != 1 and precision == "f16x3":
            _lib.check(self._lib.sn_set_conv4_fp8(self._h, int(conv4_fp8)))
        self.n_views = 0
        self.n_cameras = 0
        self._numerics = self._fresh_numerics()

    @staticmethod
    def _fresh_numerics():
        """What the NumericsGuards of this context know about the exponents in force (they are the static ones after every load_param_values)."""
        return {"calibrated": False, "report": None, "clean_probes": 0, "skipped": 0, "probes": 0, "fp8_warned": False, "mode_warned": False}

    def calibration_report(self):
        """The report of the calibration the context runs under (None: the static premultipliers)."""
        return self._numerics["report"]

    # ---- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.sn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def stream_handle(self):
        """hipStream_t of this context as an integer (e.g. for torch.cuda.ExternalStream)."""
        return int(self._lib.sn_stream(self._h) or 0)

    def synchronize(self):
        _lib.check(self._lib.sn_synchronize(self._h))

    # ---- setup ------------------------------------------------------------------------------------
    def load_param_values(self, values):
        """values: the reference weight file's list of arrays (98 or 105, weights.PARAM_LAYOUT order)."""
        blob, descs = _weights.to_blob(values)
        _lib.check(self._lib.sn_load_weights(self._h, _lib.ptr(blob), blob.size, descs, len(values)))
        self._numerics = self._fresh_numerics()          # sn_load_weights restored the static premultipliers: every guard of this context starts over

    def set_images(self, models_img):
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in models_img]
        for im in imgs:
            if im.ndim != 3 or im.shape[2] != 3:
                raise ValueError("images must be (H, W, 3) uint8 RGB, got %s" % (im.shape,))
        V = len(imgs)
        ptrs = (ctypes.c_void_p * V)(*[im.ctypes.data for im in imgs])
        H = (ctypes.c_int * V)(*[im.shape[0] for im in imgs])
        W = (ctypes.c_int * V)(*[im.shape[1] for im in imgs])
        _lib.check(self._lib.sn_set_images(self._h, V, ptrs, H, W))
        self.n_views = V

    def set_cameras(self, cameraPOs):
        P = np.ascontiguousarray(cameraPOs, dtype=np.float64)
        if P.ndim != 3 or P.shape[1:] != (3, 4):
            raise ValueError("cameraPOs must have shape (V, 3, 4), got %s" % (P.shape,))
        _lib.check(self._lib.sn_set_cameras(self._h, P.shape[0], _lib.ptr(P)))
        self.n_cameras = P.shape[0]

    # ---- hot path, host arrays --------------------------------------------------------------------
    def _batch_args(self, selected_viewPairs, xyz, resol):
        pairs = np.ascontiguousarray(selected_viewPairs, dtype=np.int64)
        if pairs.ndim != 3 or pairs.shape[2] != 2:
            raise ValueError("selected_viewPairs must have shape (N_cubes, N_viewPairs, 2), got %s" % (pairs.shape,))
        n, n_vp = pairs.shape[:2]
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(n, 3)
        resol = np.ascontiguousarray(resol, dtype=np.float32).reshape(n)
        return pairs, xyz, resol, n, n_vp

    def cvc(self, selected_viewPairs, xyz, resol, mean=None):
        pairs, xyz, resol, n, n_vp = self._batch_args(selected_viewPairs, xyz, resol)
        s = self.cube_D
        out = np.empty((n * n_vp, 6, s, s, s), dtype=np.float32)
        m = None if mean is None else np.ascontiguousarray(mean, dtype=np.float32).reshape(6)
        _lib.check(self._lib.sn_cvc(self._h, n, n_vp, _lib.ptr(pairs), _lib.ptr(xyz), _lib.ptr(resol), _lib.ptr(m), _lib.ptr(out)))
        return out

    def forward(self, X, w=None, n_vp=1, return_unfused=True):
        s = self.cube_D
        if not isinstance(X, np.ndarray) or X.dtype != np.float32:
            raise TypeError("X must be a float32 ndarray (the reference's Theano function rejects other dtypes)")
        if X.ndim != 5 or X.shape[1:] != (6, s, s, s):
            raise TypeError("X must have shape (N*n_vp, 6, %d, %d, %d), got %s" % (s, s, s, X.shape))
        if X.shape[0] % n_vp:
            raise ValueError("X.shape[0]=%d is not a multiple of n_vp=%d" % (X.shape[0], n_vp))
        n = X.shape[0] // n_vp
        X = np.ascontiguousarray(X)
        if n_vp > 1:
            if not isinstance(w, np.ndarray) or w.dtype != np.float32 or w.shape != (n, n_vp):
                raise TypeError("w must be a float32 ndarray of shape (%d, %d)" % (n, n_vp))
            w = np.ascontiguousarray(w)
        else:
            w = None
        fused = np.empty((n, 1, s, s, s), dtype=np.float32)
        unfused = np.empty((n, n_vp, s, s, s), dtype=np.float32) if return_unfused else None
        _lib.check(self._lib.sn_forward(self._h, n, n_vp, _lib.ptr(X), _lib.ptr(w), _lib.ptr(fused), _lib.ptr(unfused)))
        return fused, unfused

    def cvc_forward(self, selected_viewPairs, xyz, resol, w=None, mean=MEAN_CVC_RGBRGB, return_unfused=True, return_cvc=False):
        pairs, xyz, resol, n, n_vp = self._batch_args(selected_viewPairs, xyz, resol)
        s = self.cube_D
        if n_vp > 1:
            if w is None:
                raise TypeError("cvc_forward: w (n, n_vp) float32 is required when a cube has more than one view pair (the reference passes 1/N_vp per pair "
                                "when weighted fusion is off, main_reconstruct.py:114-115)")
            w = np.ascontiguousarray(w, dtype=np.float32)
            if w.size != n * n_vp:
                raise TypeError("cvc_forward: w must have shape (%d, %d), got %s" % (n, n_vp, w.shape))
            w = w.reshape(n, n_vp)
        else:
            w = None
        m = np.ascontiguousarray(mean, dtype=np.float32).reshape(6)
        fused = np.empty((n, 1, s, s, s), dtype=np.float32)
        unfused = np.empty((n, n_vp, s, s, s), dtype=np.float32) if return_unfused else None
        cvc = np.empty((n * n_vp, 6, s, s, s), dtype=np.float32) if return_cvc else None
        _lib.check(self._lib.sn_cvc_forward(self._h, n, n_vp, _lib.ptr(pairs), _lib.ptr(xyz), _lib.ptr(resol), _lib.ptr(m),
                                            _lib.ptr(w), _lib.ptr(fused), _lib.ptr(unfused), _lib.ptr(cvc)))
        return fused, unfused, cvc

    def relative_weights(self, features, n_vp):
        f = np.ascontiguousarray(features, dtype=np.float32)
        if f.ndim != 2 or f.shape[1] != _weights.D_VIEWPAIR_FEATURE or f.shape[0] % n_vp:
            raise TypeError("features must have shape (n*n_vp, %d)" % _weights.D_VIEWPAIR_FEATURE)
        n = f.shape[0] // n_vp
        out = np.empty((n, n_vp), dtype=np.float32)
        _lib.check(self._lib.sn_relative_weights(self._h, n, n_vp, _lib.ptr(f), _lib.ptr(out)))
        return out

    def viewpair_weights(self, embeddings, dissimilarity, theta):
        """(n_cubes,n_views,128), (n_cubes,P), (n_cubes,P) float32 -> (n_cubes,P) softmax weights over all 2-combinations."""
        e = np.ascontiguousarray(embeddings, dtype=np.float32)
        n, V = e.shape[:2]
        P = V * (V - 1) // 2
        d = np.ascontiguousarray(dissimilarity, dtype=np.float32).reshape(n, P)
        t = np.ascontiguousarray(theta, dtype=np.float32).reshape(n, P)
        out = np.empty((n, P), dtype=np.float32)
        _lib.check(self._lib.sn_viewpair_weights(self._h, n, V, _lib.ptr(e), _lib.ptr(d), _lib.ptr(t), _lib.ptr(out)))
        return out

    def color_fuse(self, cvc_minus_mean, unfused, w, mean=MEAN_CVC_RGBRGB):
        """utils.generate_voxelLevelWeighted_coloredCubes (utils/utils.py:8-42): (n,3,s,s,s) uint8 fused colours from the
        mean-subtracted CVC tensor (n*n_vp,6,s,s,s), the unfused predictions (n,n_vp,s,s,s) and the pair weights (n,n_vp)."""
        s = self.cube_D
        unfused = np.ascontiguousarray(unfused, dtype=np.float32)
        n, n_vp = unfused.shape[:2]
        cvc = np.ascontiguousarray(cvc_minus_mean, dtype=np.float32).reshape(n * n_vp, 6, s, s, s)
        w = np.ascontiguousarray(w, dtype=np.float32).reshape(n, n_vp)
        m = np.ascontiguousarray(mean, dtype=np.float32).reshape(6)
        rgb = np.empty((n, 3, s, s, s), dtype=np.uint8)
        _lib.check(self._lib.sn_color_fuse(self._h, n, n_vp, _lib.ptr(cvc), _lib.ptr(m), _lib.ptr(unfused), _lib.ptr(w), _lib.ptr(rgb)))
        return rgb

    # ---- hot path, device-resident ------------------------------------------------------------------
    # ---- similarityNet / early rejection -------------------------------------------
    def load_simil_param_values(self, values):
        """values: the similarityNet weight file's 30 arrays (weights.SIMIL_PARAM_SHAPES order)."""
        blob, descs = _weights.simil_to_blob(values)
        _lib.check(self._lib.sn_simil_load_weights(self._h, _lib.ptr(blob), blob.size, descs, len(values)))

    def crop_patches(self, view, center_h, center_w):
        """image.cropImgPatches(img=images[view], pyramidRate=1, cubeCenter_hw=(center_h, center_w)) -> (n,64,64,3) uint8."""
        ch = np.ascontiguousarray(center_h, dtype=np.float64).reshape(-1)
        cw = np.ascontiguousarray(center_w, dtype=np.float64).reshape(-1)
        out = np.empty((ch.size, 64, 64, 3), dtype=np.uint8)
        _lib.check(self._lib.sn_crop_patches(self._h, int(view), ch.size, _lib.ptr(ch), _lib.ptr(cw), _lib.ptr(out)))
        return out

    def patch2embedding(self, patches):
        """patch2embedding_fn: preprocessed (n,3,64,64) float32 patches -> (n,128) float32 embeddings."""
        if not isinstance(patches, np.ndarray) or patches.dtype != np.float32 or patches.ndim != 4 or patches.shape[1:] != (3, 64, 64):
            raise TypeError("patches must be a float32 ndarray of shape (n, 3, 64, 64)")
        x = np.ascontiguousarray(patches)
        out = np.empty((x.shape[0], _weights.D_EMBEDDING), dtype=np.float32)
        _lib.check(self._lib.sn_patch2embedding(self._h, x.shape[0], _lib.ptr(x), _lib.ptr(out)))
        return out

    def crop_embed(self, view, center_h, center_w, mean_bgr):
        """crop + preprocess + embedding of n cube centres of one view without leaving HBM -> (n,128)."""
        ch = np.ascontiguousarray(center_h, dtype=np.float64).reshape(-1)
        cw = np.ascontiguousarray(center_w, dtype=np.float64).reshape(-1)
        m = np.ascontiguousarray(mean_bgr, dtype=np.float32).reshape(3)
        out = np.empty((ch.size, _weights.D_EMBEDDING), dtype=np.float32)
        _lib.check(self._lib.sn_crop_embed(self._h, int(view), ch.size, _lib.ptr(ch), _lib.ptr(cw), _lib.ptr(m), _lib.ptr(out)))
        return out

    def embeddingpair2simil(self, emb_pairs):
        """embeddingPair2simil_fn: (2n,128) float32 (rows 2i, 2i+1 = pair i) -> (n,1) float32."""
        if not isinstance(emb_pairs, np.ndarray) or emb_pairs.dtype != np.float32 or emb_pairs.ndim != 2 or emb_pairs.shape[1] != _weights.D_EMBEDDING \
                or emb_pairs.shape[0] % 2:
            raise TypeError("embeddingPair must be a float32 matrix (2n, %d)" % _weights.D_EMBEDDING)
        e = np.ascontiguousarray(emb_pairs)
        out = np.empty((e.shape[0] // 2, 1), dtype=np.float32)
        _lib.check(self._lib.sn_embeddingpair2simil(self._h, e.shape[0] // 2, _lib.ptr(e), _lib.ptr(out)))
        return out

    def embeddings2simil(self, embeddings):
        """(n_cubes, n_views, 128) float32 -> (n_cubes, n_views*(n_views-1)/2) float32, all 2-combinations of views."""
        e = np.ascontiguousarray(embeddings, dtype=np.float32)
        if e.ndim != 3 or e.shape[2] != _weights.D_EMBEDDING or e.shape[1] < 2:
            raise TypeError("embeddings must have shape (n_cubes, n_views >= 2, %d)" % _weights.D_EMBEDDING)
        out = np.empty((e.shape[0], e.shape[1] * (e.shape[1] - 1) // 2), dtype=np.float32)
        _lib.check(self._lib.sn_embeddings2simil(self._h, e.shape[0], e.shape[1], _lib.ptr(e), _lib.ptr(out)))
        return out

    def project(self, xyz_3D, projection_M=None, return_int_hw=True, return_depth=False):
        """camera.perspectiveProj (utils/camera.py:123-184) of (n,3) points through V cameras in one launch: projection_M
        (V,3,4) float64, or None = the cameras of set_cameras -> img_h, img_w [, depth], each (V, n); int64 when
        return_int_hw (numpy's .round().astype(int64)), else float64."""
        pts = np.ascontiguousarray(xyz_3D, dtype=np.float64)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError("points must have shape (n, 3), got %s" % (pts.shape,))
        M = None if projection_M is None else np.ascontiguousarray(projection_M, dtype=np.float64)
        if M is not None and (M.ndim != 3 or M.shape[1:] != (3, 4)):
            raise ValueError("projection_M must have shape (V, 3, 4), got %s" % (M.shape,))
        n, V = pts.shape[0], (self.n_cameras if M is None else M.shape[0])
        h = np.empty((V, n), dtype=np.float64)
        w = np.empty((V, n), dtype=np.float64)
        d = np.empty((V, n), dtype=np.float64) if return_depth else None
        _lib.check(self._lib.sn_project_points(self._h, V, _lib.ptr(M), n, _lib.ptr(pts), 1 if return_int_hw else 0, _lib.ptr(h), _lib.ptr(w),
                                               _lib.ptr(d)))
        if return_int_hw:
            h, w = h.astype(np.int64), w.astype(np.int64)
        return (h, w, d) if return_depth else (h, w)

    # ---- post-pass ----------------------------------------------------------------
    def ray_pool(self, selected_viewPairs, xyz, resol, prediction, prediction_thresh=None):
        """rayPooling.rayPooling_1cube_numpy (utils/rayPooling.py:143-260) for n cubes: prediction (n,s,s,s) (or
        (n,1,s,s,s)) probabilities >= 0, compared as float16 -> votes (n,s,s,s) uint8."""
        pairs, xyz, resol, n, n_vp = self._batch_args(selected_viewPairs, xyz, resol)
        s = self.cube_D
        pred = np.ascontiguousarray(np.asarray(prediction).astype(np.float16).astype(np.float32).reshape(n, s, s, s))
        votes = np.empty((n, s, s, s), dtype=np.uint8)
        use = 0 if prediction_thresh is None else 1
        thr = 0.0 if prediction_thresh is None else float(np.float16(prediction_thresh))
        _lib.check(self._lib.sn_ray_pool(self._h, n, n_vp, _lib.ptr(pairs), _lib.ptr(xyz), _lib.ptr(resol), _lib.ptr(pred), use, thr,
                                         _lib.ptr(votes)))
        return votes

    def dense2sparse(self, prediction, rgb, selected_viewPairs, xyz, resol, min_prob=0.5, rayPool_thresh=0, enable_centerCrop=False,
                     cube_Dcenter=None, enable_rayPooling=False):
        """sparseCubes.dense2sparse (utils/sparseCubes.py:9-77) on the GPU. prediction (n,s,s,s) or (n,1,s,s,s) float;
        rgb (n,3,s,s,s) uint8 or None. Returns (offsets (n+1,) int64, ijk (T,3) u8, pred (T,) f16, rgb (T,3) u8 | None,
        votes (T,) u8 | None): cube i owns rows offsets[i]:offsets[i+1]."""
        pairs, xyz, resol, n, n_vp = self._batch_args(selected_viewPairs, xyz, resol)
        s = self.cube_D
        pred = np.ascontiguousarray(np.asarray(prediction).astype(np.float16).astype(np.float32).reshape(n, s, s, s))
        dc = int(cube_Dcenter) if enable_centerCrop else s
        cap = n * dc ** 3
        cfg = _lib.SparseCfg(float(np.float16(min_prob)), int(rayPool_thresh), int(bool(enable_centerCrop)), dc, int(bool(enable_rayPooling)))
        rgb_in = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint8).reshape(n, 3, s, s, s)
        offsets = np.zeros((n + 1,), dtype=np.int64)
        ijk = np.empty((cap, 3), dtype=np.uint8)
        p16 = np.empty((cap,), dtype=np.float16)
        rgb_out = None if rgb is None else np.empty((cap, 3), dtype=np.uint8)
        votes = np.empty((cap,), dtype=np.uint8) if enable_rayPooling else None
        _lib.check(self._lib.sn_dense2sparse(self._h, n, n_vp, _lib.ptr(pairs), _lib.ptr(xyz), _lib.ptr(resol), _lib.ptr(pred), _lib.ptr(rgb_in),
                                             ctypes.byref(cfg), _lib.ptr(offsets), _lib.ptr(ijk), _lib.ptr(p16), _lib.ptr(rgb_out),
                                             _lib.ptr(votes)))
        T = int(offsets[-1])
        return offsets, ijk[:T], p16[:T], (None if rgb_out is None else rgb_out[:T]), (None if votes is None else votes[:T])

    def dev_alloc(self, nbytes):
        p = self._lib.sn_dev_alloc(self._h, int(nbytes))
        if not p:
            raise _lib.SurfaceNetHipError("sn_dev_alloc failed: %s" % _lib.last_error())
        return p

    def dev_free(self, p):
        _lib.check(self._lib.sn_dev_free(self._h, p))

    def h2d(self, dst_dev, arr):
        arr = np.ascontiguousarray(arr)
        _lib.check(self._lib.sn_memcpy_h2d(self._h, dst_dev, _lib.ptr(arr), arr.nbytes))

    def d2h(self, arr, src_dev):
        assert arr.flags["C_CONTIGUOUS"]
        _lib.check(self._lib.sn_memcpy_d2h(self._h, _lib.ptr(arr), src_dev, arr.nbytes))

    def mark(self, slot):
        """Records point `slot` (0..7) on the context's stream for d2h_after."""
        _lib.check(self._lib.sn_mark(self._h, int(slot)))

    def d2h_after(self, slot, arr, src_dev):
        """d2h that waits only for the work enqueued BEFORE mark(slot), not for what was enqueued since."""
        assert arr.flags["C_CONTIGUOUS"]
        _lib.check(self._lib.sn_memcpy_d2h_after(self._h, int(slot), _lib.ptr(arr), src_dev, arr.nbytes))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.dev_alloc(max(arr.nbytes, 16))
        self.h2d(p, arr)
        return p

    def cvc_forward_dev(self, n, n_vp, pairs_dev, xyz_dev, resol_dev, w_dev, fused_dev, unfused_dev=None, cvc_out_dev=None,
                        mean=MEAN_CVC_RGBRGB):
        m = np.ascontiguousarray(mean, dtype=np.float32).reshape(6)
        _lib.check(self._lib.sn_cvc_forward_dev(self._h, n, n_vp, pairs_dev, xyz_dev, resol_dev, _lib.ptr(m), w_dev, fused_dev,
                                                unfused_dev, cvc_out_dev))

    def color_fuse_dev(self, n, n_vp, cvc_dev, unfused_dev, w_dev, rgb_dev, mean=MEAN_CVC_RGBRGB):
        m = np.ascontiguousarray(mean, dtype=np.float32).reshape(6)
        _lib.check(self._lib.sn_color_fuse_dev(self._h, n, n_vp, cvc_dev, _lib.ptr(m), unfused_dev, w_dev, rgb_dev))

    def ray_pool_dev(self, n, n_vp, pairs_dev, xyz_dev, resol_dev, pred_dev, votes_dev, prediction_thresh=None):
        use = 0 if prediction_thresh is None else 1
        thr = 0.0 if prediction_thresh is None else float(np.float16(prediction_thresh))
        _lib.check(self._lib.sn_ray_pool_dev(self._h, n, n_vp, pairs_dev, xyz_dev, resol_dev, pred_dev, use, thr, votes_dev))

    def dense2sparse_dev(self, n, n_vp, pairs_dev, xyz_dev, resol_dev, pred_dev, rgb_dev, votes_ws_dev, offsets_dev, ijk_dev, pred16_dev,
                         rgb_out_dev, votes_out_dev, min_prob=0.5, rayPool_thresh=0, enable_centerCrop=False, cube_Dcenter=None,
                         enable_rayPooling=False):
        dc = int(cube_Dcenter) if enable_centerCrop else self.cube_D
        cfg = _lib.SparseCfg(float(np.float16(min_prob)), int(rayPool_thresh), int(bool(enable_centerCrop)), dc, int(bool(enable_rayPooling)))
        _lib.check(self._lib.sn_dense2sparse_dev(self._h, n, n_vp, pairs_dev, xyz_dev, resol_dev, pred_dev, rgb_dev, ctypes.byref(cfg),
                                                 votes_ws_dev, offsets_dev, ijk_dev, pred16_dev, rgb_out_dev, votes_out_dev))

    # ---- numerics of the 6-bit code planes (default mode) ---------------------------------------------------
    def calibrate(self, n_samples=0, max_sat_fraction=1e-3):
        """Data-driven premultipliers of the two code planes (sn_calibrate_dev): looks at the activations the LAST forward call left in the
        workspace (`n_samples` of them, 0 = all that call ran - run `forward` / `cvc_forward` on a representative batch first) and sets each plane's exponent to the
        largest one that saturates at most `max_sat_fraction` of the values. Returns a dict (exponents and saturated fractions before / after,
        largest magnitudes). The setting holds until the next load_param_values / precision change."""
        cal = _lib.Calibration()
        _lib.check(self._lib.sn_calibrate_dev(self._h, int(n_samples), float(max_sat_fraction), ctypes.byref(cal)))
        out = {k: getattr(cal, k) for k, _ in _lib.Calibration._fields_}
        if max_sat_fraction >= 0:                        # (a negative bound only measures)
            self._numerics["calibrated"], self._numerics["report"] = True, out
        return out

    def numeric_status(self):
        """Warning-level numeric status: names of the layers whose stored outputs exceeded the range of their 6-bit code plane since the last
        call (such a value loses its own correction term only; not an error). Clears the warning bits."""
        bits = ctypes.c_uint(0)
        names = ctypes.create_string_buffer(2048)
        _lib.check(self._lib.sn_numeric_status(self._h, ctypes.byref(bits), names, 2048))
        nm = [x for x in names.value.decode().split(",") if x]
        return [nm[i] if i < len(nm) else "layer %d" % i for i in range(32) if bits.value & (1 << i)]

    # ---- multi-GPU exchange (RCCL, native; torch.distributed is not required) ---------------------------
    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().sn_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, world, rank, unique_id, timeout_s=None):
        """Joins the RCCL communicator (collective: every rank calls it with rank 0's unique id). timeout_s: bound on the wait for the other ranks
        (sn_comm_init_deadline: SurfaceNetHipError when it expires, the context stays usable without a communicator); None = wait for ever."""
        self.comm_world = self.comm_rank = 0
        _lib.check(self._lib.sn_comm_init_deadline(self._h, int(world), int(rank), ctypes.c_char_p(bytes(unique_id)), float(timeout_s or 0.0)))
        self.comm_world, self.comm_rank = int(world), int(rank)

    @staticmethod
    def comm_info():
        """-> (file the RCCL entry points are bound to [+ whether the process had it mapped already], ncclGetVersion code)."""
        buf, ver = ctypes.create_string_buffer(1024), ctypes.c_int(0)
        _lib.check(_lib.load().sn_comm_info(buf, 1024, ctypes.byref(ver)))
        return buf.value.decode("utf-8", "replace"), int(ver.value)

    def allgather_f32_dev(self, local_dev, n_local, global_dev):
        _lib.check(self._lib.sn_allgather_f32_dev(self._h, local_dev, int(n_local), global_dev))

    def allgather_f32_dev_overlap(self, local_dev, n_local, global_dev, slot):
        """The all-gather on the context's communication stream, behind the kernels submitted so far and overlapping the ones submitted next;
        `comm_wait(slot)` orders the kernel stream behind it again (before the two buffers are reused)."""
        _lib.check(self._lib.sn_allgather_f32_dev_overlap(self._h, local_dev, int(n_local), global_dev, int(slot)))

    def comm_wait(self, slot):
        _lib.check(self._lib.sn_comm_wait(self._h, int(slot)))

    def allgatherv_bytes(self, blob):
        """Variable-length all-gather of one byte string per rank through the library's RCCL binding: `blob` (np.uint8, any length incl. 0) -> list
        of `world` np.uint8 arrays in rank order. The payloads travel device to device over xGMI; only this rank's blob goes up and the gathered
        bytes come down. Collective sequence, identical on every rank whatever the blob sizes are (`allgatherv_two_step`): the counts
        (sn_allgatherv_counts), then counts + payloads into a buffer sized from them (sn_allgatherv_bytes_dev) - no rank ever retries on its own."""
        world = getattr(self, "comm_world", 0)
        if not world:
            raise _lib.SurfaceNetHipError("allgatherv_bytes: comm_init has not been called on this context")
        blob = np.ascontiguousarray(np.asarray(blob, dtype=np.uint8).reshape(-1))
        d_local = self.upload(blob) if blob.size else None

        def counts_fn(n_local):
            counts = (ctypes.c_ulonglong * world)()
            _lib.check(self._lib.sn_allgatherv_counts(self._h, int(n_local), counts))
            return [int(c) for c in counts]

        def payload_fn(n_local, total):
            counts = (ctypes.c_ulonglong * world)()
            try:
                d_all, alloc_err = self.dev_alloc(max(total, 16)), None
            except _lib.SurfaceNetHipError as e:
                # a rank that cannot hold the result must STILL run the call's collectives (its peers are entering them): with no destination the
                # library runs them all and reports "too small" afterwards; the allocation failure is raised then
                d_all, alloc_err = None, e
            try:
                rc = self._lib.sn_allgatherv_bytes_dev(self._h, d_local, int(n_local), d_all, int(total) if d_all else 0, counts)
                if alloc_err is not None:
                    raise alloc_err
                _lib.check(rc)
                out = np.empty((total,), dtype=np.uint8)
                if total:
                    self.d2h(out, d_all)
            finally:
                if d_all:
                    self.dev_free(d_all)
            return [int(c) for c in counts], out
        try:
            return allgatherv_two_step(int(blob.size), counts_fn, payload_fn)
        finally:
            if d_local is not None:
                self.dev_free(d_local)

    # ---- measurement --------------------------------------------------------------------------------
    def mfma_probe(self, target_ms=10.0):
        """What this box sustains on a pure fp16 MFMA stream (sn_mfma_probe) -> (dense fp16 TFLOP/s, shader clock in GHz)."""
        tf, ghz = ctypes.c_double(), ctypes.c_double()
        _lib.check(self._lib.sn_mfma_probe(self._h, float(target_ms), ctypes.byref(tf), ctypes.byref(ghz)))
        return tf.value, ghz.value

    def profile_enable(self, on=True):
        _lib.check(self._lib.sn_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        _lib.check(self._lib.sn_profile_reset(self._h))

    def profile(self):
        """-> {kernel tag: dict(ms, launches, flops, bytes)} accumulated since the last reset."""
        n = self._lib.sn_profile_count(self._h)
        if n < 0:
            _lib.check(n)
        out = {}
        name = ctypes.create_string_buffer(64)
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        cnt = ctypes.c_int64()
        for i in range(n):
            _lib.check(self._lib.sn_profile_get(self._h, i, name, 64, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by)))
            out[name.value.decode()] = dict(ms=ms.value, launches=cnt.value, flops=fl.value, bytes=by.value)
        return out
