"""Process-wide runtime state behind the drop-in modules: one `Context` per (device, cube_D), lazily sized,
with scene (cameras + images) and weights bound on demand."""
import numpy as np

from .context import Context, MEAN_CVC_RGBRGB  # noqa: F401

DEFAULT_CUBE_D = 32        # params.py:65 (__cube_D in {32, 64})
DEFAULT_MAX_SAMPLES = 64
_device = 0
_contexts = {}
_param_values = None
_simil_values = None
_scene_key = {}
_cam_key = {}
_img_key = {}
_single_img_keepalive = {}


def set_device(device):
    global _device
    _device = int(device)


def set_param_values(values):
    """Registers the weight list; (re)loaded into every live context."""
    global _param_values
    _param_values = values
    for ctx in _contexts.values():
        ctx.load_param_values(values)


def set_simil_param_values(values):
    """Registers the similarityNet weight list; (re)loaded into every live context."""
    global _simil_values
    _simil_values = values
    for ctx in _contexts.values():
        ctx.load_simil_param_values(values)


def any_context():
    """A context for work that does not depend on cube_D (similarityNet, patch cropping)."""
    for ctx in _contexts.values():
        return ctx
    return context_for(DEFAULT_CUBE_D)


def bind_images(ctx, models_img):
    """Images only (patch cropping needs no cameras). Changing them drops the cached scene binding."""
    key = (id(models_img), len(models_img), tuple(id(im) for im in models_img))
    if _img_key.get(id(ctx)) == key:
        return
    ctx.set_images(models_img)
    _img_key[id(ctx)] = key
    _scene_key.pop(id(ctx), None)


def bind_single_image(ctx, img):
    """For image.cropImgPatches(img=...): returns the view index of `img` in the context, uploading it as a one-image set
    when it is not one of the bound images."""
    key = _img_key.get(id(ctx))
    if key is not None and id(img) in key[2]:
        return key[2].index(id(img))
    holder = [img]
    _single_img_keepalive[id(ctx)] = holder
    bind_images(ctx, holder)
    return 0


def context_for(cube_D, n_samples=1):
    key = (_device, int(cube_D))
    ctx = _contexts.get(key)
    want = max(DEFAULT_MAX_SAMPLES, 1)
    if ctx is None:
        ctx = Context(cube_D=cube_D, max_samples=want, device=_device)
        if _param_values is not None:
            ctx.load_param_values(_param_values)
        if _simil_values is not None:
            ctx.load_simil_param_values(_simil_values)
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
    _img_key[id(ctx)] = key[:3]


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
    _img_key.clear()
