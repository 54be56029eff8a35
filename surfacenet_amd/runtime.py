"""Process-wide runtime state behind the drop-in modules: one `Context` per (device, cube_D), lazily sized,
with scene (cameras + images) and weights bound on demand."""
import numpy as np

from .context import Context, MEAN_CVC_RGBRGB  # noqa: F401

DEFAULT_CUBE_D = 64        # params.py:65 (__cube_D = 64; {32, 64} are the supported sizes). The drop-in callables infer cube_D from X.shape.
DEFAULT_MAX_SAMPLES = None  # None: by cube size - 128 cube-view-pair samples at s <= 32 (5.2 GB of workspace; 128 samples fill every layer's tile rounds on 256 CUs exactly),
                            # 64 above (s = 64: 21 GB). An int overrides it for contexts created afterwards.
_device = 0
_contexts = {}
_param_values = None
_simil_values = None
_scene_key = {}
_cam_key = {}
_img_key = {}
_img_refs = {}      # id(ctx) -> (list object, [arrays]): strong references, so that the ids in a key can never be recycled


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


_preferred_cube_D = None


def prefer_cube_D(cube_D):
    """Tells the runtime which cube size the scene at hand will use (SurfaceNet_inference(cube_D=...) and reconstruct_scene call this),
    so that cube-size independent work that comes first (projections, patch cropping, the similarityNet) lands in THAT context instead
    of creating a second one - with its own workspace and its own copy of the images - for the default size."""
    global _preferred_cube_D
    _preferred_cube_D = None if cube_D is None else int(cube_D)


_pinned_ctx = None


class scene_context(object):
    """`with runtime.scene_context(ctx=..., cube_D=...):` - for the duration of the block the cube-size independent work of the drop-in modules
    (projections, patch cropping, the similarityNet) runs in the caller's own `ctx` (also one built by hand, which `context_for` does not know)
    or, without one, in the context of `cube_D`. The previous setting - also `prefer_cube_D`'s - is restored on exit, so a scene does not leave
    its choice behind for unrelated calls of the same process."""

    def __init__(self, ctx=None, cube_D=None):
        self.ctx, self.cube_D = ctx, cube_D

    def __enter__(self):
        global _pinned_ctx, _preferred_cube_D
        self._saved = (_pinned_ctx, _preferred_cube_D)
        _pinned_ctx = self.ctx
        if self.ctx is None and self.cube_D is not None:
            _preferred_cube_D = int(self.cube_D)
        return self

    def __exit__(self, *exc):
        global _pinned_ctx, _preferred_cube_D
        _pinned_ctx, _preferred_cube_D = self._saved
        return False


def any_context():
    """A context for work that does not depend on cube_D (projections, similarityNet, patch cropping): the one a running scene pinned
    (scene_context), else the preferred size's (prefer_cube_D), else any live one, else a new one of the smaller size."""
    if _pinned_ctx is not None:
        return _pinned_ctx
    if _preferred_cube_D is not None:
        return context_for(_preferred_cube_D)
    for ctx in _contexts.values():
        return ctx
    return context_for(32)        # cube-size independent work: the smaller workspace


def _fingerprint(im):
    """Cheap content fingerprint of one image: shape, dtype and a checksum of a strided sample of its bytes (<= ~4k samples), so
    that an image mutated in place, or a new array that landed on a freed array's address, is re-uploaded."""
    a = np.asarray(im)
    flat = a.reshape(-1)
    step = max(1, flat.size // 4096)
    sample = np.ascontiguousarray(flat[::step])
    return (a.shape, a.dtype.str, int(sample.view(np.uint8).astype(np.uint64).dot(np.arange(1, sample.nbytes + 1, dtype=np.uint64) % np.uint64(65521))))


def _images_key(models_img):
    return (id(models_img), len(models_img), tuple(id(im) for im in models_img), tuple(_fingerprint(im) for im in models_img))


def invalidate(ctx=None):
    """Forget what is bound (all contexts, or one): the next bind_* call uploads again. For callers that mutate images in
    place beyond what the strided fingerprint can see."""
    for d in (_scene_key, _cam_key, _img_key, _img_refs):
        if ctx is None:
            d.clear()
        else:
            d.pop(id(ctx), None)


def bind_images(ctx, models_img):
    """Images only (patch cropping needs no cameras). Changing them drops the cached scene binding."""
    key = _images_key(models_img)
    if _img_key.get(id(ctx)) == key:
        return
    ctx.set_images(models_img)
    _img_key[id(ctx)] = key
    _img_refs[id(ctx)] = (models_img, list(models_img))
    _scene_key.pop(id(ctx), None)


def bind_single_image(ctx, img):
    """For image.cropImgPatches(img=...): returns the view index of `img` in the context, uploading it as a one-image set
    when it is not one of the bound images."""
    key = _img_key.get(id(ctx))
    if key is not None and id(img) in key[2]:
        v = key[2].index(id(img))
        if key[3][v] == _fingerprint(img):
            return v
    bind_images(ctx, [img])
    return 0


def context_for(cube_D, n_samples=1):
    key = (_device, int(cube_D))
    ctx = _contexts.get(key)
    want = max(int(DEFAULT_MAX_SAMPLES), 1) if DEFAULT_MAX_SAMPLES else (128 if int(cube_D) <= 32 else 64)
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
    ikey = _images_key(models_img)
    key = ikey + (cams.tobytes(),)
    if _scene_key.get(id(ctx)) == key:
        return
    if len(models_img) != cams.shape[0]:
        raise ValueError("cameraPOs has %d views but models_img has %d" % (cams.shape[0], len(models_img)))
    ctx.set_cameras(cams)
    ctx.set_images(models_img)
    _scene_key[id(ctx)] = key
    _cam_key[id(ctx)] = cams.tobytes()
    _img_key[id(ctx)] = ikey
    _img_refs[id(ctx)] = (models_img, list(models_img))


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
    global _preferred_cube_D
    _preferred_cube_D = None
    for ctx in _contexts.values():
        ctx.close()
    _contexts.clear()
    _scene_key.clear()
    _cam_key.clear()
    _img_key.clear()
    _img_refs.clear()
