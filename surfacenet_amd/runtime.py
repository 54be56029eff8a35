"""Process-wide runtime state behind the drop-in modules: one `Context` per (device, cube_D), lazily sized,
with scene (cameras + images) and weights bound on demand."""
import numpy as np

from .context import Context, MEAN_CVC_RGBRGB  # noqa: F401

DEFAULT_CUBE_D = 32        # params.py:65 (__cube_D in {32, 64})
DEFAULT_MAX_SAMPLES = 64
_device = 0
_contexts = {}
_param_values = None
_scene_key = {}
_cam_key = {}


def set_device(device):
    global _device
    _device = int(device)


def set_param_values(values):
    """Registers the weight list; (re)loaded into every live context."""
    global _param_values
    _param_values = values
    for ctx in _contexts.values():
        ctx.load_param_values(values)


def context_for(cube_D, n_samples=1):
    key = (_device, int(cube_D))
    ctx = _contexts.get(key)
    want = max(DEFAULT_MAX_SAMPLES, 1)
    if ctx is None:
        ctx = Context(cube_D=cube_D, max_samples=want, device=_device)
        if _param_values is not None:
            ctx.load_param_values(_param_values)
        _contexts[key] = ctx
    return ctx


def bind_scene(ctx, cameraPOs, models_img):
    """Uploads cameras/images when they differ from what the context already holds (identity + cheap checks)."""
    cams = np.ascontiguousarray(cameraPOs, dtype=np.float64)
    key = (id(models_img), len(models_img), tuple(id(im) for im in models_img), cams.tobytes())
    if _scene_key.get(id(ctx)) == key:
        return
    if len(models_img) != cams.shape[0]:
        raise ValueError("cameraPOs has %d views but models_img has %d" % (cams.shape[0], len(models_img)))
    ctx.set_cameras(cams)
    ctx.set_images(models_img)
    _scene_key[id(ctx)] = key
    _cam_key[id(ctx)] = cams.tobytes()


def bind_cameras(ctx, cameraPOs):
    """Cameras only (ray pooling needs no images). Changing them drops the cached scene binding."""
    cams = np.ascontiguousarray(cameraPOs, dtype=np.float64)
    key = cams.tobytes()
    if _cam_key.get(id(ctx)) == key:
        return
    ctx.set_cameras(cams)
    _cam_key[id(ctx)] = key
    _scene_key.pop(id(ctx), None)


def reset():
    for ctx in _contexts.values():
        ctx.close()
    _contexts.clear()
    _scene_key.clear()
    _cam_key.clear()
