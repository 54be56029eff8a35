/*
 * surfacenet_hip.h — C ABI of libsurfacenet_hip.so (MI355X / gfx950).
 *
 * The reference (mjiUST/SurfaceNet) has no FFI: its hot path is three Python callables used in
 * main_reconstruct.py:134-146. This header is what a ctypes binding of those callables needs; each
 * entry point cites the reference interface it replaces (paths relative to the reference root).
 * The Python side that presents the reference's own signatures on top of this ABI lives in
 * surfacenet_amd/CVC.py and surfacenet_amd/SurfaceNet.py.
 *
 * Conventions: plain C; every pointer is caller-owned HOST memory unless the parameter name ends in
 * `_dev` (device memory of the context's GPU); functions returning int give 0 on success and a
 * negative sn_status on failure, with a human-readable message from sn_last_error() (thread-local).
 * One sn_ctx per GPU per host thread; a context is not thread-safe, the library has no global state.
 * All work of a context is issued on one HIP stream owned by the context; host-pointer entry points
 * are synchronous, `_dev` entry points are asynchronous until sn_synchronize().
 */
#ifndef SURFACENET_HIP_H
#define SURFACENET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SN_ABI_VERSION 2   /* 2 (round 6): + sn_mfma_probe, sn_set_conv4_fp8; sn_calibrate_dev refuses the all-MX mode with SN_ERR_STATE (since round 5) */

/* The library is built with -fvisibility=hidden: the functions below are its WHOLE dynamic symbol table
 * (tests/test_abi.py compares `nm -D` with this header). */
#define SN_API __attribute__((visibility("default")))

typedef struct sn_ctx sn_ctx;

enum sn_status {
    SN_OK = 0,
    SN_ERR_ARG = -1,     /* bad argument (shape, null pointer, view id out of range ...) */
    SN_ERR_STATE = -2,   /* call order: weights / images / cameras not set                */
    SN_ERR_HIP = -3,     /* HIP runtime error; text carries hipGetErrorString             */
    SN_ERR_NOMEM = -4,
    SN_ERR_COMM = -5,
    SN_ERR_RANGE = -6    /* a conv layer stored a non-finite value or one beyond the fp16 range of its storage format */
};

/* One parameter array inside the weight blob (reference pickle = flat list of arrays,
 * nets/SurfaceNet.py:397-400; order documented in SURVEY.md App. B / DESIGN.md). */
typedef struct {
    int64_t offset;   /* in floats, into `blob` */
    int32_t ndim;
    int32_t shape[5];
} sn_param_desc;

/* ---- lifetime -------------------------------------------------------------------------------- */
/* cube_D = s of the s^3 colored voxel cube (params.py:65, 32 or 64; any multiple of 4 in [8,96]
 * except 36 and 68 is accepted); max_samples = largest n*n_vp processed per internal pass (activation
 * workspace is sized for it; larger calls are chunked). Returns NULL on failure (sn_last_error). */
SN_API sn_ctx *sn_create(int device_id, int cube_D, int max_samples);
SN_API void sn_destroy(sn_ctx *ctx);
SN_API const char *sn_last_error(void);
SN_API int sn_version(void);
SN_API int sn_synchronize(sn_ctx *ctx);

/* Arithmetic of the 3D-CNN (the CVC warp is always the reference's fp64/int arithmetic):
 *   SN_PRECISION_F16X3 (default): operands carried as hi+lo pairs of fp16 (22 significant bits), three
 *       MFMAs per product term, fp32 accumulate -> fp32-class results. The two LAST 3x3x3 layers (merge_conv_a,
 *       merge_conv_b: 56 % of a step) compute their two correction terms on one MX-scaled MFMA with 6-bit
 *       (fp6 e2m3) operands, which issues at twice the fp16 rate: 1.5 MFMA units per product instead of 3;
 *       the three dilated layers conv4_1 .. conv4_3 (round 5) compute theirs on one MX-scaled MFMA with fp8 e4m3
 *       operands: 2 units (fp8, not fp6: their activations need the exponent range).
 *       L_inf vs the fp64 oracle 3e-5 .. 1.7e-4 (asserted 2e-4, bar 1e-3);
 *   SN_PRECISION_F16X3_PURE: all three MFMAs in fp16 in every layer (L_inf ~1e-5);
 *   SN_PRECISION_F16: operands rounded to fp16, fp32 accumulate -> 3x faster, L_inf ~2e-3 on BN-normalised
 *       nets, i.e. above the 1e-3 parity bar; opt-in fast mode.
 * Call before sn_load_weights (weights are packed for the selected mode). */
#define SN_PRECISION_F16 0
#define SN_PRECISION_F16X3 1
/*   SN_PRECISION_F16M8 (experimental; the default dominates it in speed and accuracy): EVERY layer with the main term
 *       on the f16 MFMA and the two 2^-11 correction terms on one MX-scaled 6-bit MFMA
 *       (v_mfma_scale_f32_16x16x128_f8f6f4); L_inf 1e-4 .. 4e-4 (bar 1e-3). */
#define SN_PRECISION_F16M8 2
#define SN_PRECISION_F16X3_PURE 3
SN_API int sn_set_precision(sn_ctx *ctx, int mode);
SN_API int sn_get_precision(sn_ctx *ctx);
/* SN_PRECISION_F16X3 only: which layers of the dilated chain conv4_1 .. conv4_3 (nets/layers.py:200-253) compute their correction terms on the fp8 MX
 * MFMA (2 MFMA units per product) instead of three fp16 MFMAs. on = 1 (default): all three - worst observed L_inf 1.83e-4 over the 202-input survey
 * of round 6, conv4_x 1.15 ms per 128 samples; on = 2: conv4_2 and conv4_3 only (conv4_1 on three fp16 MFMAs: <= 1.54e-4 on the survey's worst
 * inputs, +0.07 ms); on = 0: none - the round-4 arithmetic (<= 9.5e-5 on the same inputs, conv4_x 1.5 ms). The merge layers keep their 6-bit
 * correction step in every setting. Call after sn_set_precision (which resets it to 1) and before sn_load_weights (a change discards packed
 * weights: SN_ERR_STATE from the forward calls until they are loaded again). */
SN_API int sn_set_conv4_fp8(sn_ctx *ctx, int on);

/* ---- one-time setup -------------------------------------------------------------------------- */
/* Replaces lasagne.layers.set_all_param_values(...) in SurfaceNet_inference
 * (nets/SurfaceNet.py:385-402). `descs` lists the 105 arrays of the reference pickle in its order
 * (98 for the network alone: the relative-weight MLP arrays may be omitted). BN folding and the
 * fp16 MFMA-fragment packing happen inside. */
SN_API int sn_load_weights(sn_ctx *ctx, const float *blob, size_t n_floats, const sn_param_desc *descs, int n_params);
/* models_img of CVC.gen_coloredCubes (utils/CVC.py:56): V images, (H[v], W[v], 3) uint8 RGB. */
SN_API int sn_set_images(sn_ctx *ctx, int V, const uint8_t *const *imgs, const int *H, const int *W);
/* cameraPOs of CVC.gen_coloredCubes: (V,3,4) float64 row-major projection matrices. */
SN_API int sn_set_cameras(sn_ctx *ctx, int V, const double *P);

/* ---- hot path, host buffers ------------------------------------------------------------------ */
/* CVC.gen_coloredCubes (utils/CVC.py:56-104) [+ CVC.preprocess_augmentation, utils/CVC.py:108-111,
 * when mean6 != NULL]. view_pairs (n, n_vp, 2) int64 indices into the image/camera lists;
 * xyz (n,3) float32 cube min corners; resol (n,) float32; out (n*n_vp, 6, s,s,s) float32. */
SN_API int sn_cvc(sn_ctx *ctx, int n, int n_vp, const int64_t *view_pairs, const float *xyz, const float *resol,
           const float *mean6, float *out);
/* nViewPair_SurfaceNet_fn (nets/SurfaceNet.py:365-382; call at main_reconstruct.py:145-146).
 * X (n*n_vp, 6, s,s,s) float32 mean-subtracted; w (n, n_vp) float32 (NULL iff n_vp == 1);
 * fused (n,1,s,s,s); unfused (n,n_vp,s,s,s) or NULL. */
SN_API int sn_forward(sn_ctx *ctx, int n, int n_vp, const float *X, const float *w, float *fused, float *unfused);
/* The loop body main_reconstruct.py:134-146 in one call: CVC warp -> mean subtraction -> CNN ->
 * fusion, nothing but the cube parameters crossing PCIe. cvc_out (optional) receives the
 * mean-subtracted CVC tensor the reference keeps for colour fusion (main_reconstruct.py:150). */
SN_API int sn_cvc_forward(sn_ctx *ctx, int n, int n_vp, const int64_t *view_pairs, const float *xyz, const float *resol,
                   const float *mean6, const float *w, float *fused, float *unfused, float *cvc_out);
/* viewPair_relativeImpt_fn (nets/SurfaceNet.py:334-338; used at utils/viewPairSelection.py:77):
 * features (n*n_vp, 258) float32 -> softmax weights (n, n_vp). */
SN_API int sn_relative_weights(sn_ctx *ctx, int n, int n_vp, const float *features, float *weights);

/* The weight computation of viewPairSelection.viewPairSelection (utils/viewPairSelection.py:63-77) for every 2-combination of
 * views (itertools.combinations order) in one call: embeddings (n_cubes,n_views,128), dissimilarity and theta (n_cubes,P)
 * float32 -> softmax weights (n_cubes,P). Bit-identical to building the (n_cubes*P,258) feature rows and calling
 * sn_relative_weights with n_vp = P. */
SN_API int sn_viewpair_weights(sn_ctx *ctx, int n_cubes, int n_views, const float *embeddings, const float *dissimilarity,
                        const float *theta, float *weights);

/* utils.generate_voxelLevelWeighted_coloredCubes (utils/utils.py:8-42; call at main_reconstruct.py:150-152), float32 op for
 * op: cvc (n*n_vp,6,s,s,s) is the MEAN-SUBTRACTED tensor of sn_cvc_forward (the caller's `X += mean` is applied inside);
 * unfused (n,n_vp,s,s,s), w (n,n_vp) -> rgb (n,3,s,s,s) uint8. */
SN_API int sn_color_fuse(sn_ctx *ctx, int n, int n_vp, const float *cvc, const float *mean6, const float *unfused, const float *w,
                  unsigned char *rgb);
SN_API int sn_color_fuse_dev(sn_ctx *ctx, int n, int n_vp, const float *cvc_dev, const float *mean6, const float *unfused_dev,
                      const float *w_dev, unsigned char *rgb_dev);

/* ---- hot path, device-resident (asynchronous on the context's stream) ------------------------- */
SN_API void *sn_dev_alloc(sn_ctx *ctx, size_t bytes);
SN_API int sn_dev_free(sn_ctx *ctx, void *p_dev);
SN_API int sn_memcpy_h2d(sn_ctx *ctx, void *dst_dev, const void *src, size_t bytes);
SN_API int sn_memcpy_d2h(sn_ctx *ctx, void *dst, const void *src_dev, size_t bytes);
/* Pipelined readback for loops that enqueue the next batch before they fetch the previous one (the reference's hot loop collects a
 * sparse list per cube and batch, main_reconstruct.py:126-160): sn_mark records point `slot` (0..7) on the context's stream;
 * sn_memcpy_d2h_after copies on a second stream as soon as that point has been reached and returns when the copy is done - work
 * enqueued on the context's stream AFTER the mark keeps running meanwhile (sn_memcpy_d2h would wait for all of it). */
SN_API int sn_mark(sn_ctx *ctx, int slot);
SN_API int sn_memcpy_d2h_after(sn_ctx *ctx, int slot, void *dst, const void *src_dev, size_t bytes);
/* The HIP stream (hipStream_t) every asynchronous entry point of this context is ordered on, for interop: record / wait
 * events on it, or wrap it (e.g. torch.cuda.ExternalStream) to order collectives against the kernels without host syncs. */
SN_API void *sn_stream(sn_ctx *ctx);
/* Same as sn_cvc_forward with every array already in HBM. n*n_vp <= max_samples. mean6 is host. */
SN_API int sn_cvc_forward_dev(sn_ctx *ctx, int n, int n_vp, const int64_t *view_pairs_dev, const float *xyz_dev,
                       const float *resol_dev, const float *mean6, const float *w_dev, float *fused_dev,
                       float *unfused_dev, float *cvc_out_dev);
SN_API int sn_cvc_dev(sn_ctx *ctx, int n, int n_vp, const int64_t *view_pairs_dev, const float *xyz_dev,
               const float *resol_dev, const float *mean6, float *out_dev);
SN_API int sn_forward_dev(sn_ctx *ctx, int n, int n_vp, const float *X_dev, const float *w_dev, float *fused_dev,
                   float *unfused_dev);

/* ---- post-pass of the loop body (main_reconstruct.py:153-160) ------------------- */
/* rayPooling.rayPooling_1cube_numpy (utils/rayPooling.py:143-260) for n cubes at once: pred (n,s,s,s) float32
 * probabilities >= 0 (rounded to float16 inside, as append_dense_2sparseList does at utils/sparseCubes.py:136 before
 * the call), view_pairs (n,n_vp,2) -> votes (n,s,s,s) uint8 (max 2*n_vp). use_thresh = 0 is prediction_thresh=None;
 * otherwise voxels with fp16(pred) > fp16(min_prob) take part. Needs sn_set_cameras only. SN_ERR_ARG if a projected
 * pixel / depth bin falls outside the int32 range (a cube on the camera plane; the reference has no such limit). */
SN_API int sn_ray_pool(sn_ctx *ctx, int n, int n_vp, const int64_t *view_pairs, const float *xyz, const float *resol,
                const float *pred, int use_thresh, float min_prob, unsigned char *votes);
/* Device-resident, asynchronous; the range error is reported by the next sn_synchronize. */
SN_API int sn_ray_pool_dev(sn_ctx *ctx, int n, int n_vp, const int64_t *view_pairs_dev, const float *xyz_dev,
                    const float *resol_dev, const float *pred_dev, int use_thresh, float min_prob,
                    unsigned char *votes_dev);

/* sparseCubes.dense2sparse (utils/sparseCubes.py:9-77) keyword arguments. */
typedef struct sn_sparse_cfg {
    float min_prob;          /* compared in float16, as numpy does for a float16 array */
    int rayPool_thresh;
    int enable_centerCrop;
    int cube_Dcenter;        /* used when enable_centerCrop != 0; (s - cube_Dcenter) / 2 voxels are cut on each side */
    int enable_rayPooling;
} sn_sparse_cfg;
/* pred (n,s,s,s) float32 fused probabilities, rgb (n,3,s,s,s) uint8 (sn_color_fuse's output; may be NULL) ->
 * packed voxel lists of all cubes, cube after cube, voxels in ascending flat index of the (cropped) cube:
 *   offsets (n+1) int64: cube i owns [offsets[i], offsets[i+1]); empty cubes have zero length (the reference skips them)
 *   ijk (total,3) uint8 | pred16 (total) float16 bits | rgb_out (total,3) uint8 | votes_out (total) uint8
 * Output arrays must hold n*Dc^3 entries (Dc = cube_Dcenter when cropping, else s); rgb_out / votes_out may be NULL.
 * votes_out is filled only when enable_rayPooling. The caller shifts xyz by resol*(s-Dc)/2 (sparseCubes.py:55). */
SN_API int sn_dense2sparse(sn_ctx *ctx, int n, int n_vp, const int64_t *view_pairs, const float *xyz, const float *resol,
                    const float *pred, const unsigned char *rgb, const sn_sparse_cfg *cfg, int64_t *offsets,
                    unsigned char *ijk, uint16_t *pred16, unsigned char *rgb_out, unsigned char *votes_out);
/* Same with every array in HBM (offsets too); asynchronous. votes_ws_dev (n,s,s,s) uint8 scratch is required when
 * enable_rayPooling. */
SN_API int sn_dense2sparse_dev(sn_ctx *ctx, int n, int n_vp, const int64_t *view_pairs_dev, const float *xyz_dev,
                        const float *resol_dev, const float *pred_dev, const unsigned char *rgb_dev,
                        const sn_sparse_cfg *cfg, unsigned char *votes_ws_dev, int64_t *offsets_dev,
                        unsigned char *ijk_dev, uint16_t *pred16_dev, unsigned char *rgb_out_dev,
                        unsigned char *votes_out_dev);

/* ---- similarityNet / early rejection (main_reconstruct.py:76-97) ---------------- */
/* pickle.load + set_all_param_values([embedding layer, similarity layer]) of similarityNet_inference
 * (nets/similarityNet.py:229-244): 30 arrays in order — 13 x (conv W (Cout,Cin,3,3), b (Cout,)) for conv1_1 .. conv5_3
 * (cross-correlation, as Conv2DDNNLayer), embedding W (5888,128), b (128,), similarity W (1,1), b (1,). */
SN_API int sn_simil_load_weights(sn_ctx *ctx, const float *blob, size_t n_floats, const sn_param_desc *descs, int n_params);
/* image.cropImgPatches(img = view's image, pyramidRate = 1, cubeCenter_hw = (center_h, center_w)) (utils/image.py:92-183, as
 * called at utils/earlyRejection.py:50): n patches (n,64,64,3) uint8 RGB around the truncated centre projections,
 * coordinates clamped to the image. center_h / center_w: float64 (n,). Needs sn_set_images. */
SN_API int sn_crop_patches(sn_ctx *ctx, int view, int n, const double *center_h, const double *center_w, unsigned char *patches);
/* patch2embedding_fn (nets/similarityNet.py:219-221): preprocessed patches (n,3,64,64) float32 (BGR - mean) -> (n,128). */
SN_API int sn_patch2embedding(sn_ctx *ctx, int n, const float *patches, float *embeddings);
/* The inner loop of earlyRejection.patch2embedding (utils/earlyRejection.py:50-53) without leaving HBM: crop +
 * image.preprocess_patches (utils/image.py:9-36, mean_bgr[3]) + patch2embedding_fn for n cube centres of one view. */
SN_API int sn_crop_embed(sn_ctx *ctx, int view, int n, const double *center_h, const double *center_w, const float *mean_bgr,
                  float *embeddings);
/* embeddingPair2simil_fn (nets/similarityNet.py:223-226): rows 2i, 2i+1 of emb_pairs (2*n_pairs,128) -> (n_pairs,1)
 * sigmoid(w * ||e1 - e2||_2 + b). */
SN_API int sn_embeddingpair2simil(sn_ctx *ctx, int n_pairs, const float *emb_pairs, float *similarity);
/* earlyRejection.embeddingPairs2simil (utils/earlyRejection.py:59-90) in one call: embeddings (n_cubes, n_views, 128) ->
 * similarity (n_cubes, n_views*(n_views-1)/2), pairs in itertools.combinations order; bit-identical to feeding the same pairs
 * through sn_embeddingpair2simil, without shipping every embedding once per pair across PCIe. */
SN_API int sn_embeddings2simil(sn_ctx *ctx, int n_cubes, int n_views, const float *embeddings, float *similarity);

/* camera.perspectiveProj (utils/camera.py:123-184; calls at main_reconstruct.py:62-65 through perspectiveProj_cubesCorner):
 * V cameras x n points in one launch. P (V,3,4) float64 row-major, or NULL = the cameras of sn_set_cameras (V ignored);
 * xyz (n,3) float64 -> img_h, img_w (V,n) float64 (row v = camera v), depth (V,n) or NULL. Same arithmetic as the CVC
 * warp's projection (fp64 FMA chain over k, IEEE divide); round_int != 0 applies numpy's .round() (half-to-even) - the
 * caller casts to int64. */
SN_API int sn_project_points(sn_ctx *ctx, int V, const double *P, int n, const double *xyz, int round_int, double *img_h,
                      double *img_w, double *depth);

/* ---- numerics of the default mode's 6-bit code planes ------------------------------------------------------------
 * The two merge layers read their inputs as fp16 + 6-bit e2m3 codes scaled by a per-tensor premultiplier 2^s ("cat": the concat buffer of
 * sigmoid side outputs; "act": merge_conv_a's ReLU output). (The dilated layers conv4_1 .. conv4_3 run the same two-term correction on fp8 e4m3
 * codes since round 5: those have the exponent range, there is nothing to calibrate.) s is static - sized from the layers' BatchNorm parameters, which trained nets obey
 * (nets/SurfaceNet.py:33-74, every batch_norm) - unless the caller calibrates it on data:
 *   sn_calibrate_dev looks at the activations the LAST forward call left in the workspace (n_samples of them, <= 0: all that call ran; run
 *   sn_forward / sn_cvc_forward on a representative batch first - SN_ERR_STATE when none has run since the weights / the mode were set, or when it
 *   ran fewer samples; default precision mode only), and sets each tensor's s to the largest value whose saturated fraction (|v| * 2^s > 7.5) stays <= max_sat_fraction.
 *   The new exponents apply to every later call of this context; sn_load_weights / sn_set_precision restore the static ones. A negative
 *   max_sat_fraction only MEASURES: the report holds the saturated fractions under the exponents in force (s_* = s_*_before), nothing changes.
 * A value beyond the code range loses (only) its own correction term; it is not an error. sn_numeric_status reports - and clears - the WARNING bits:
 * which layers stored such values since the last call (names: comma-separated layer names, bit i = the i-th). */
typedef struct sn_calibration {
    int s_act_before, s_cat_before, s_act, s_cat;                /* premultiplier exponents before / after */
    double sat_act_before, sat_cat_before, sat_act, sat_cat;     /* saturated fraction of the non-zero values under them */
    float max_act, max_cat;                                      /* largest stored magnitude */
} sn_calibration;
SN_API int sn_calibrate_dev(sn_ctx *ctx, int n_samples, double max_sat_fraction, sn_calibration *out);
SN_API int sn_numeric_status(sn_ctx *ctx, unsigned *saturated_bits, char *names, int names_cap);

/* ---- multi-GPU (one process per GPU): the path's only exchange is an all-gather of the per-cube fused probabilities
 * (the reference is single-GPU, no counterpart). RCCL over xGMI; librccl is dlopen'ed on first use.
 * Rank 0 calls sn_comm_unique_id and ships the 128 bytes to the other ranks by any means; every rank then calls
 * sn_comm_init (collective). sn_allgather_f32_dev is asynchronous on the context's stream. */
SN_API int sn_comm_unique_id(char *id128);
SN_API int sn_comm_init(sn_ctx *ctx, int world, int rank, const char *id128);
/* The same with a bound on the wait (ncclCommInitRank blocks until EVERY rank has called it): after timeout_s seconds the call returns
 * SN_ERR_COMM, the context stays without a communicator and remains usable for everything but the exchange (the helper thread that is still
 * inside RCCL is abandoned). timeout_s <= 0: no deadline (= sn_comm_init; the caller vouches that all ranks arrive). */
SN_API int sn_comm_init_deadline(sn_ctx *ctx, int world, int rank, const char *id128, double timeout_s);
/* Which RCCL the entry points are bound to: file name + whether the host process had it mapped already (then that copy is used: a process
 * must not run two RCCL copies) and the ncclGetVersion code. Loads librccl if nothing has yet. */
SN_API int sn_comm_info(char *file, int file_cap, int *version_code);
SN_API int sn_allgather_f32_dev(sn_ctx *ctx, const float *local_dev, size_t n_local, float *global_dev);
/* The same on the context's own communication stream, ordered behind everything submitted to the kernel stream so far: the all-gather of
 * batch i overlaps the kernels of batch i + 1. slot (0..7) names its completion; sn_comm_wait(ctx, slot) makes the kernel stream wait for it
 * (call it before local_dev / global_dev are written again). sn_synchronize waits for both streams. */
SN_API int sn_allgather_f32_dev_overlap(sn_ctx *ctx, const float *local_dev, size_t n_local, float *global_dev, int slot);
SN_API int sn_comm_wait(sn_ctx *ctx, int slot);
/* Variable-length all-gather of bytes - the exchange of the packed sparse voxel lists of a sharded scene (SURVEY §8e "counts then
 * all-gather-v"; utils/sparseCubes.py:9-77 produces the lists, main_reconstruct.py:153-160 accumulates them): every rank contributes
 * n_local bytes of device memory (0 allowed, different per rank); global_dev receives the contributions back to back in rank order and
 * counts[r] (host, `world` entries) their sizes. Synchronous. Every rank issues the same collectives whatever its own arguments are (counts; one
 * 8-byte status word per rank - a rank that cannot allocate its staging buffer says so there and EVERY rank returns the error without entering
 * the payload step; payloads): a destination that cannot hold the total is reported AFTER the payload all-gather (SN_ERR_ARG, counts[] filled in)
 * and must NOT be answered by a retry of this rank alone. Size the destination first with sn_allgatherv_counts (collective: the 8-byte counts all-gather alone). */
SN_API int sn_allgatherv_counts(sn_ctx *ctx, size_t n_local, unsigned long long *counts);
SN_API int sn_allgatherv_bytes_dev(sn_ctx *ctx, const void *local_dev, size_t n_local, void *global_dev, size_t global_cap,
                                   unsigned long long *counts);

/* ---- measurement ------------------------------------------------------------------------------ */
/* Per-kernel HIP-event timing on the context's stream. While enabled every kernel launch is
 * bracketed by events; sn_profile_get drains them. idx enumerates kernel tags (layer names);
 * returns 1 past the last. flops / bytes are the ALGORITHMIC work of the recorded launches. */
SN_API int sn_profile_enable(sn_ctx *ctx, int on);
SN_API int sn_profile_count(sn_ctx *ctx);
SN_API int sn_profile_get(sn_ctx *ctx, int idx, char *name, int name_cap, double *ms_total, int64_t *launches,
                   double *flops, double *bytes);
SN_API int sn_profile_reset(sn_ctx *ctx);
/* What THIS box sustains on a pure stream of v_mfma_f32_16x16x32_f16 (all CUs, one wave per SIMD, random fp16 operands in registers): boxes of
 * the same SKU fall into speed classes 5-8 % apart (power / clock management), and every absolute number of a run - cubes/s, kernel times,
 * `roofline.frac` against the nominal 2.5 PF - moves with it. Runs a few launches of ~target_ms (<= 0: 10 ms) on the context's stream (DVFS settles
 * within the first), returns the last one's rate in dense fp16 TFLOP/s and the shader clock it ran at (cycle counter / event time). Synchronous.
 * No reference counterpart: measurement infrastructure (bench.py -> "box"). */
SN_API int sn_mfma_probe(sn_ctx *ctx, double target_ms, double *tflops, double *ghz);

#ifdef __cplusplus
}
#endif
#endif /* SURFACENET_HIP_H */
