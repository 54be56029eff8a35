"""ctypes binding of libsurfacenet_hip.so (include/surfacenet_hip.h).

There is NO CPU fallback: if the shared library is missing or no gfx950 device is visible every
entry point raises `SurfaceNetHipError` — the product path never routes through oracle/ or numpy.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SURFACENET_HIP_LIB") or os.path.join(_HERE, "libsurfacenet_hip.so")   # override: profiling builds

# Every symbol include/surfacenet_hip.h declares (tests/test_abi.py checks the two lists agree).
ABI_SYMBOLS = [
    "sn_create", "sn_destroy", "sn_last_error", "sn_version", "sn_synchronize", "sn_set_precision", "sn_get_precision", "sn_set_conv4_fp8", "sn_stream",
    "sn_load_weights", "sn_set_images", "sn_set_cameras",
    "sn_cvc", "sn_forward", "sn_cvc_forward", "sn_relative_weights", "sn_viewpair_weights", "sn_color_fuse", "sn_color_fuse_dev",
    "sn_dev_alloc", "sn_dev_free", "sn_memcpy_h2d", "sn_memcpy_d2h", "sn_mark", "sn_memcpy_d2h_after",
    "sn_cvc_forward_dev", "sn_cvc_dev", "sn_forward_dev",
    "sn_ray_pool", "sn_ray_pool_dev", "sn_dense2sparse", "sn_dense2sparse_dev",
    "sn_simil_load_weights", "sn_crop_patches", "sn_patch2embedding", "sn_crop_embed", "sn_embeddingpair2simil", "sn_embeddings2simil",
    "sn_project_points",
    "sn_comm_unique_id", "sn_comm_init", "sn_comm_init_deadline", "sn_comm_info", "sn_allgather_f32_dev", "sn_allgather_f32_dev_overlap", "sn_comm_wait", "sn_allgatherv_counts", "sn_allgatherv_bytes_dev",
    "sn_calibrate_dev", "sn_numeric_status",
    "sn_profile_enable", "sn_profile_count", "sn_profile_get", "sn_profile_reset", "sn_mfma_probe",
]


class SurfaceNetHipError(RuntimeError):
    pass


class SparseCfg(ctypes.Structure):
    _fields_ = [("min_prob", ctypes.c_float), ("rayPool_thresh", ctypes.c_int), ("enable_centerCrop", ctypes.c_int),
                ("cube_Dcenter", ctypes.c_int), ("enable_rayPooling", ctypes.c_int)]


class Calibration(ctypes.Structure):
    _fields_ = [("s_act_before", ctypes.c_int), ("s_cat_before", ctypes.c_int), ("s_act", ctypes.c_int), ("s_cat", ctypes.c_int),
                ("sat_act_before", ctypes.c_double), ("sat_cat_before", ctypes.c_double), ("sat_act", ctypes.c_double), ("sat_cat", ctypes.c_double),
                ("max_act", ctypes.c_float), ("max_cat", ctypes.c_float)]


class ParamDesc(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_int64), ("ndim", ctypes.c_int32), ("shape", ctypes.c_int32 * 5)]


_lib = None


def load():
    """Loads the HIP library (once). Raises SurfaceNetHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SurfaceNetHipError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C surfacenet_amd/csrc`). The MI355X path has no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise SurfaceNetHipError("cannot load %s: %s" % (LIB_PATH, e))
    c_void_p, c_int, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    P = ctypes.POINTER
    sig = {
        "sn_create": (c_void_p, [c_int, c_int, c_int]),
        "sn_destroy": (None, [c_void_p]),
        "sn_last_error": (ctypes.c_char_p, []),
        "sn_version": (c_int, []),
        "sn_synchronize": (c_int, [c_void_p]),
        "sn_set_precision": (c_int, [c_void_p, c_int]),
        "sn_get_precision": (c_int, [c_void_p]),
        "sn_set_conv4_fp8": (c_int, [c_void_p, c_int]),
        "sn_stream": (c_void_p, [c_void_p]),
        "sn_load_weights": (c_int, [c_void_p, c_void_p, c_size_t, P(ParamDesc), c_int]),
        "sn_set_images": (c_int, [c_void_p, c_int, P(c_void_p), P(c_int), P(c_int)]),
        "sn_set_cameras": (c_int, [c_void_p, c_int, c_void_p]),
        "sn_cvc": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
        "sn_forward": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
        "sn_cvc_forward": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 8),
        "sn_relative_weights": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
        "sn_viewpair_weights": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
        "sn_color_fuse": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 5),
        "sn_color_fuse_dev": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 5),
        "sn_dev_alloc": (c_void_p, [c_void_p, c_size_t]),
        "sn_dev_free": (c_int, [c_void_p, c_void_p]),
        "sn_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
        "sn_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
        "sn_mark": (c_int, [c_void_p, c_int]),
        "sn_memcpy_d2h_after": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t]),
        "sn_cvc_forward_dev": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 8),
        "sn_cvc_dev": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 5),
        "sn_forward_dev": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 4),
        "sn_ray_pool": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 4 + [c_int, ctypes.c_float, c_void_p]),
        "sn_ray_pool_dev": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 4 + [c_int, ctypes.c_float, c_void_p]),
        "sn_dense2sparse": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 5 + [P(SparseCfg)] + [c_void_p] * 5),
        "sn_dense2sparse_dev": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 5 + [P(SparseCfg)] + [c_void_p] * 6),
        "sn_simil_load_weights": (c_int, [c_void_p, c_void_p, c_size_t, P(ParamDesc), c_int]),
        "sn_crop_patches": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
        "sn_patch2embedding": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
        "sn_crop_embed": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
        "sn_embeddingpair2simil": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
        "sn_embeddings2simil": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
        "sn_project_points": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
        "sn_comm_unique_id": (c_int, [ctypes.c_char_p]),
        "sn_comm_init": (c_int, [c_void_p, c_int, c_int, ctypes.c_char_p]),
        "sn_comm_init_deadline": (c_int, [c_void_p, c_int, c_int, ctypes.c_char_p, ctypes.c_double]),
        "sn_comm_info": (c_int, [ctypes.c_char_p, c_int, P(c_int)]),
        "sn_calibrate_dev": (c_int, [c_void_p, c_int, ctypes.c_double, c_void_p]),
        "sn_numeric_status": (c_int, [c_void_p, c_void_p, ctypes.c_char_p, c_int]),
        "sn_allgather_f32_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
        "sn_allgather_f32_dev_overlap": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_int]),
        "sn_comm_wait": (c_int, [c_void_p, c_int]),
        "sn_allgatherv_counts": (c_int, [c_void_p, c_size_t, c_void_p]),
        "sn_allgatherv_bytes_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
        "sn_profile_enable": (c_int, [c_void_p, c_int]),
        "sn_profile_count": (c_int, [c_void_p]),
        "sn_profile_get": (c_int, [c_void_p, c_int, ctypes.c_char_p, c_int, P(ctypes.c_double), P(ctypes.c_int64),
                                   P(ctypes.c_double), P(ctypes.c_double)]),
        "sn_profile_reset": (c_int, [c_void_p]),
        "sn_mfma_probe": (c_int, [c_void_p, ctypes.c_double, P(ctypes.c_double), P(ctypes.c_double)]),
    }
    assert sorted(sig) == sorted(ABI_SYMBOLS)
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().sn_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise SurfaceNetHipError("libsurfacenet_hip: %s (status %d)" % (last_error(), rc))


def ptr(a):
    """Host pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)
