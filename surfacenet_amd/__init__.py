"""surfacenet_amd — MI355X-native inference hot path of SurfaceNet (CVC warp + 3D-CNN + view-pair fusion).

Only what the hot path needs:
  csrc/            HIP kernels + the C ABI (include/surfacenet_hip.h) -> libsurfacenet_hip.so
  context.Context  one GPU's state over the C ABI (ctypes)
  CVC, SurfaceNet  drop-in mirrors of the reference's utils/CVC.py and nets/SurfaceNet.py entry points
  reconstruct      the cube-batch loop of main_reconstruct.py:126-166 + multi-GPU sharding
  weights          weight-file layout, loader for the reference pickle, synthetic weights
"""
from .context import Context, MEAN_CVC_RGBRGB  # noqa: F401
from ._lib import SurfaceNetHipError  # noqa: F401
