"""Drop-in for the inference entry point of the reference's `nets/similarityNet.py` on the MI355X.

`similarityNet_inference(model_file, imgPatch_hw_size)` (nets/similarityNet.py:229-244) returns
    patch2embedding_fn(patches (n,3,64,64) float32, BGR - mean)  -> (n,128) float32      (:219-221)
    embeddingPair2simil_fn(embeddingPair (2n,128) float32)       -> (n,1) float32         (:223-226)
with the calling conventions of the two compiled Theano functions (TypeError on dtype/ndim mismatch). The network is
the VGG16-style stack of :23-58 (13 x conv3x3+ReLU on the 2-D form of the MFMA kernel, 5 max-pools, centre-crop features,
L2 norm, dense 5888->128), then Euclidean distance + logistic unit (:71-77).
"""
import numpy as np

from . import runtime, weights


def similarityNet_inference(model_file, imgPatch_hw_size=(64, 64), param_values=None):
    """model_file: the reference's similarityNet `*.model` pickle; `param_values` (30 arrays, weights.SIMIL_PARAM_SHAPES)
    may be given instead, e.g. weights.synthetic_simil_param_values(seed)."""
    if tuple(imgPatch_hw_size) != (64, 64):
        raise NotImplementedError("imgPatch_hw_size must be (64, 64) (params.py:92): the feature layout 5888 = f(64) is compiled in")
    values = param_values if param_values is not None else weights.load_simil_pickle(model_file)
    runtime.set_simil_param_values(values)

    def patch2embedding_fn(patch):
        if not isinstance(patch, np.ndarray) or patch.dtype != np.float32 or patch.ndim != 4:
            raise TypeError("patch must be a float32 4-D ndarray")
        return runtime.any_context().patch2embedding(patch)

    def embeddingPair2simil_fn(embeddingPair):
        if not isinstance(embeddingPair, np.ndarray) or embeddingPair.dtype != np.float32 or embeddingPair.ndim != 2:
            raise TypeError("embeddingPair must be a float32 matrix")
        return runtime.any_context().embeddingpair2simil(embeddingPair)

    patch2embedding_fn.sn_gpu = True       # lets earlyRejection.patch2embedding fuse crop + preprocess + embedding in HBM
    embeddingPair2simil_fn.sn_gpu = True   # lets earlyRejection.embeddingPairs2simil evaluate every pair in one call
    return patch2embedding_fn, embeddingPair2simil_fn
