"""Drop-in for the reference's `utils/earlyRejection.py` (call sites main_reconstruct.py:84-97).

    patch2embedding        utils/earlyRejection.py:6-56    crop patches per view -> preprocess -> similarityNet embedding
    embeddingPairs2simil   utils/earlyRejection.py:59-90   pair dissimilarity for every 2-combination of views
    selectFromSimilarity   utils/earlyRejection.py:92-103  bool mask of the cubes worth reconstructing

Same signatures and return values. With the GPU-backed `patch2embedding_fn` of surfacenet_amd.similarityNet the
crop + preprocess + network chain of a view runs without the patches ever leaving HBM (sn_crop_embed); with any other
callable the reference's three-step protocol is followed literally.
"""
import numpy as np

from . import image, runtime
from .viewPairSelection import k_combination_np, yield_batch_npBool


def patch2embedding(images_list, img_h_cubesCorner, img_w_cubesCorner, patch2embedding_fn, patches_mean_bgr, N_cubes, N_views, D_embedding,
                    patchSize, batchSize, cubeCenter_hw):
    """-> patches_embedding (N_cubes, N_views, D_embedding) float32, inScope_cubes_vs_views (N_cubes, N_views) bool.
    Out-of-scope (cube, view) entries keep the embedding of an all-black patch (utils/earlyRejection.py:31-34)."""
    inScope_cubes_vs_views = np.zeros((N_cubes, N_views), dtype=bool)
    patch_allBlack = image.preprocess_patches(np.zeros((1, patchSize, patchSize, 3), dtype=np.float32), mean_BGR=patches_mean_bgr)
    fused = bool(getattr(patch2embedding_fn, "sn_gpu", False)) and patchSize == 64
    embedding_allBlack = patch2embedding_fn(np.ascontiguousarray(patch_allBlack))[0]
    if fused:
        # (N_cubes, N_views, 128) float32 is 4.9 GB for DTU scan9's full bounding box: filling it with the black-patch embedding up front is 2.5 s of
        # first-touch page faults and memory traffic with the GPU idle. The fused path below writes every row exactly once instead - a view's in-scope rows
        # with its embeddings, its out-of-scope rows with the black-patch embedding - while the GPU embeds the next view (round 6).
        patches_embedding = np.empty((N_cubes, N_views, D_embedding), dtype=np.float32)
    else:
        patches_embedding = np.zeros((N_cubes, N_views, D_embedding), dtype=np.float32)
        patches_embedding[:, :] = embedding_allBlack
    if fused:
        ctx = runtime.any_context()
        runtime.bind_images(ctx, images_list)
    proj_h = np.stack([img_h_cubesCorner.min(axis=-1), img_h_cubesCorner.max(axis=-1)], axis=-1)
    proj_w = np.stack([img_w_cubesCorner.min(axis=-1), img_w_cubesCorner.max(axis=-1)], axis=-1)
    if fused:
        # The GPU embeds one view (one blocking C call in a worker thread; ctypes drops the GIL) while this thread does the host work of its neighbours: the
        # in-scope test of the NEXT view (22 ms for 195,360 cubes) and the write-out of the PREVIOUS one - its in-scope rows take the embeddings, its
        # out-of-scope rows the black-patch embedding (strided 0.5 KB rows, first touch of the 4.9 GB array). Nothing but the first view's in-scope test and
        # the last view's write-out is left outside the GPU's shadow (round 6: the up-front fill and the 49 in-scope tests were 3.5 s of DTU scan9's 38 s).
        # The same look-ahead slot also finds the cubes of a view whose centres share a pixel (`prepare`).
        import threading
        from concurrent.futures import ThreadPoolExecutor

        jobs = {}                                                    # view -> (centre rows (2, n_unique) float64, inverse index or None)

        def prepare(v, ins):
            # The crop takes a 64x64 window around the TRUNCATED centre projection (image.py:160-169 `.astype(np.int)`; patch_crop_kernel does the same): cubes
            # whose centres fall on the same pixel of a view get the same patch, hence - the network being independent of batch position, bit for bit
            # (tests/test_gpu_simil.py) - the same embedding. 3.5 % of DTU scan9's 6.75 M in-scope (cube, view) pairs, 2.6 % of dino's: embedded once.
            c = cubeCenter_hw[:, v, ins]
            key = c[0].astype(np.int64) * 4294967296 + c[1].astype(np.int64)
            uniq, first, inverse = np.unique(key, return_index=True, return_inverse=True)
            if uniq.size == key.size:
                first, inverse = slice(None), None
            jobs[v] = (np.ascontiguousarray(c[0][first], dtype=np.float64), np.ascontiguousarray(c[1][first], dtype=np.float64), inverse)

        def embed(v, started):
            try:
                ch, cw, _ = jobs[v]
            finally:
                started.set()                  # from here on the worker is a few bytecodes away from the C call, which drops the GIL (set even if the
            return ctx.crop_embed(v, ch, cw, patches_mean_bgr)      # preparation raised: the caller must not wait for ever - it meets the exception in result())

        def submit(pool, v):
            # (the caller goes on to numpy calls that hold the GIL for tens of ms each: wait until the worker has picked its job up, so that the GPU is
            # busy before this thread is)
            started = threading.Event()
            fut = pool.submit(embed, v, started)
            started.wait()
            return fut

        def write_out(v, emb):
            m = inScope_cubes_vs_views[:, v]
            inverse = jobs.pop(v)[2]
            patches_embedding[m, v] = emb if inverse is None else emb[inverse]
            patches_embedding[~m, v] = embedding_allBlack
        with ThreadPoolExecutor(max_workers=1) as pool:
            inflight = None                                          # (view, future) of the call the GPU is working on
            for v in list(range(len(images_list))) + [None]:
                has = False
                if v is not None:
                    ins = image.img_hw_cubesCorner_inScopeCheck(hw_shape=images_list[v].shape[:2], img_h_cubesCorner=img_h_cubesCorner[v],
                                                                img_w_cubesCorner=img_w_cubesCorner[v])
                    inScope_cubes_vs_views[:, v] = ins
                    has = bool(ins.any())
                    if has:
                        prepare(v, ins)
                retired = None
                if inflight is not None and (v is None or has):
                    retired = (inflight[0], inflight[1].result())
                    inflight = None
                if v is not None:
                    if has:
                        inflight = (v, submit(pool, v))
                    else:
                        patches_embedding[:, v] = embedding_allBlack      # no cube projects into this view
                if retired is not None:
                    write_out(*retired)
        for v in range(len(images_list), N_views):                   # (fewer images than views: the reference leaves those columns black)
            patches_embedding[:, v] = embedding_allBlack
        return patches_embedding, inScope_cubes_vs_views
    for _view, _image in enumerate(images_list):
        _inScope = image.img_hw_cubesCorner_inScopeCheck(hw_shape=_image.shape[:2], img_h_cubesCorner=img_h_cubesCorner[_view],
                                                         img_w_cubesCorner=img_w_cubesCorner[_view])
        inScope_cubes_vs_views[:, _view] = _inScope
        n_in = int(_inScope.sum())
        if not n_in:
            continue
        centers = cubeCenter_hw[:, _view, _inScope]
        patches = image.cropImgPatches(img=_image, range_h=proj_h[_view][_inScope], range_w=proj_w[_view][_inScope], patchSize=patchSize,
                                       pyramidRate=1, interp_order=2, cubeCenter_hw=centers)
        pre = np.ascontiguousarray(image.preprocess_patches(patches.astype(np.float32), mean_BGR=patches_mean_bgr))
        emb = np.zeros((n_in, D_embedding), dtype=np.float32)
        for _batch in yield_batch_npBool(N_all=n_in, batch_size=batchSize):
            emb[_batch] = patch2embedding_fn(pre[_batch])
        patches_embedding[_inScope, _view] = emb
    return patches_embedding, inScope_cubes_vs_views


def embeddingPairs2simil(embeddings, N_views, inScope_cubes_vs_views, embeddingPair2simil_fn, batchSize, viewPairs):
    """embeddings (N_cubes, N_views, D) -> dissimilarity (N_cubes, N_viewPairs) over all 2-combinations of range(N_views)
    (the `viewPairs` argument is recomputed, as in the reference: utils/earlyRejection.py:74)."""
    viewPairs = k_combination_np(range(N_views), k=2)
    N_viewPairs, N_cubes = viewPairs.shape[0], embeddings.shape[0]
    if getattr(embeddingPair2simil_fn, "sn_gpu", False) and embeddings.shape[1] == N_views:
        # one upload of the embeddings, every 2-combination on the GPU (bit-identical to the batched protocol below)
        return runtime.any_context().embeddings2simil(embeddings)
    # rows (cube i, view j) in cube-major, pair-major, (first, second) order == the reference's yield_batch_ij_npBool walk
    ii = np.repeat(np.arange(N_cubes), 2 * N_viewPairs)
    jj = np.tile(viewPairs.flatten(), N_cubes)
    step = int(batchSize * 2)
    out = []
    for a in range(0, ii.size, step):
        out.append(embeddingPair2simil_fn(np.ascontiguousarray(embeddings[ii[a:a + step], jj[a:a + step]])))
    return np.vstack(out).reshape((N_cubes, N_viewPairs))


def selectFromSimilarity(dissimilarityProb, N_viewPairs4inference):
    """(N_cubes,) bool: at least N_viewPairs4inference pairs with 0.1 < dissimilarity < 0.5 (utils/earlyRejection.py:92-103)."""
    similarityBool = (dissimilarityProb < 0.5) & (dissimilarityProb > 0.1)
    return (similarityBool.sum(axis=1) >= N_viewPairs4inference).astype(bool)
