// sn_simil.hip — C ABI of the similarityNet / early-rejection stage over simil.h and the 2-D form
// of conv3d_f16_mfma.
#include "sn_internal.h"
#include "simil.h"

// ---- similarityNet + patch cropping ---------------------------------------------------------------
// 13 x (3x3 conv + bias + ReLU) on the 2-D form of the MFMA kernel; a chunk of n patches is one volume (x = patch index).
static const int kSimC[14] = {3, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};   // channel chain
static const int kSimStage[13] = {0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4};                          // H = 64 >> stage
static const char *const kSimName[13] = {"s_conv1_1", "s_conv1_2", "s_conv2_1", "s_conv2_2", "s_conv3_1", "s_conv3_2", "s_conv3_3",
                                         "s_conv4_1", "s_conv4_2", "s_conv4_3", "s_conv5_1", "s_conv5_2", "s_conv5_3"};
static constexpr int kSimParams = 30, kSimNF = 4;
// patches per pass: two 64x64 group planes of a chunk (2040 * 4096 * 16 B * 2 = 267.4 MB) must stay below the 2^28 - 16 offset field of the
// conv kernel's buffer-addressed halo staging (conv3d_mfma.h)
static constexpr int kSimChunk = 2040;
// channel groups per slab / K-chunks per weight piece: f16x3 (two activation planes) 2 / 2 (3 measured equal); f16 (one plane)
// 4 / 3, i.e. 36 groups = exactly 9 chunks per slab: +18 % in that mode
#define SIMCS8 (SP == 0 ? 4 : 2)
#ifndef SN_SIMPCH
#define SN_SIMPCH 2
#endif
#define SIMPCH (SP == 0 ? 3 : SN_SIMPCH)
static int simil_cs8(int mode) { return mode == 0 ? 4 : 2; }
#define SCONV 3, 1, 4, kSimNF, EPI_STORE, SP, SIMCS8, SIMPCH, 8, 0, 1
// the 4x4 maps of conv5_x: one MFMA voxel fragment = one image (K2D = 2), 16 images x 128 output channels per workgroup
#define SCONV5 3, 1, 2, 8, EPI_STORE, SP, 2, 2, 8, 0, 2
// the last conv of every block writes the 2x2 max-pooled map directly (EPI_POOL2D): the unpooled map is never stored
#define SCONVP 3, 1, 4, kSimNF, EPI_POOL2D, SP, SIMCS8, SIMPCH, 8, 0, 1
#define SCONV5P 3, 1, 2, 8, EPI_POOL2D, SP, 2, 2, 8, 0, 2
// SN_SIM_NF8 (round 4): 128 output channels per workgroup with ONE-group channel slabs for the f16x3 layers with >= 128 outputs (two-group slabs + 128-channel weight pieces
// would need 169 KB of LDS): half the halo DMAs per MFMA, a slab boundary every 2.25 chunks. Same-box (profiles/r4/ab_r4ak_nf8.log): s_conv2_1 -6 %, s_conv3_x -2..5 %, s_conv4_x
// -3..6 %, s_conv2_2 (pooled, 128 outputs) +5 % -> not that one; similarityNet +2.5 % patches/s. 0: every layer on the 64-channel kernels; 1: unpooled layers only.
#ifndef SN_SIM_NF8
#define SN_SIM_NF8 2
#endif
// SN_SIM_MF8 (round 6, experiment): the 64-output layers of the 64x64 / 32x32 maps (s_conv1_1, s_conv1_2, s_conv2_2 in two cout splits) with EIGHT voxel fragments per wave over
// one-group slabs - 16 patches x 8x8 pixels x 64 channels per workgroup: bursts of 96 instead of 48 MFMAs per segment, 24 instead of 32 operand reads and half the weight DMAs per
// 96 MFMAs, the same halo bytes per MFMA
#ifndef SN_SIM_MF8
#define SN_SIM_MF8 0
#endif
#define SCONV16 3, 1, 8, kSimNF, EPI_STORE, SP, 1, 2, 8, 0, 1
#define SCONV16P 3, 1, 8, kSimNF, EPI_POOL2D, SP, 1, 2, 8, 0, 1
#define SCONV8 3, 1, 4, 8, EPI_STORE, SP, 1, 2, 8, 0, 1
#define SCONV8P 3, 1, 4, 8, EPI_POOL2D, SP, 1, 2, 8, 0, 1
static bool simil_wide(int i, int mode)
{
    const bool last = (i == 12 || kSimStage[i + 1] != kSimStage[i]);
    return SN_SIM_NF8 && mode == 1 && kSimStage[i] < 4 && kSimC[i + 1] >= 128 && (!last || (SN_SIM_NF8 >= 2 && kSimC[i + 1] >= 256));
}
static int simil_nf(int i, int mode = 0) { return kSimStage[i] == 4 || simil_wide(i, mode) ? 8 : kSimNF; }
static bool simil_mf8(int i, int mode) { return SN_SIM_MF8 && mode == 1 && kSimStage[i] < 2 && !simil_wide(i, mode); }

static int simil_mode(sn_ctx *c) { return c->split == 0 ? 0 : 1; }   // f16m8 contexts run this net in f16x3 (own workspace)

static int simil_pack(sn_ctx *c)
{
    const int want = simil_mode(c);
    if (c->simil_split == want) return SN_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc;
    for (int i = 0; i < 13; ++i) {
        PackedConv &L = c->sconv[i];
        dev_free_owned(c, L.wpack); dev_free_owned(c, L.scale); dev_free_owned(c, L.shift);
        L = PackedConv();
        L.name = kSimName[i]; L.cin = kSimC[i]; L.cout = kSimC[i + 1]; L.ks = 3; L.dil = 1; L.act = 0; L.k2d = 1;
        static const bool no_bridge = sn_ab_switch("SN_SIMIL_NO_BRIDGE") != nullptr;      // (A/B switch)
        L.bridge = (want == 1 && !no_bridge) ? 1 : 0;       // f16x3: two-group slabs = 4.5 K-chunks -> 9 chunks per slab pair (pack_conv_host decides per layer)
        const float *W = c->simil_host.data() + c->simil_descs[2 * i].offset, *b = c->simil_host.data() + c->simil_descs[2 * i + 1].offset;
        std::vector<float> one((size_t)L.cout, 1.f), zero((size_t)L.cout, 0.f);
        if ((rc = pack_conv(c, L, W, b, one.data(), zero.data(), one.data(), simil_nf(i, want), L.cout / (16 * simil_nf(i, want)), kSimStage[i] == 4 ? 2 : ((simil_wide(i, want) || simil_mf8(i, want)) ? 1 : simil_cs8(want)), want)) != SN_OK) return rc;
    }
    c->simil_split = want;
    return SN_OK;
}

extern "C" int sn_simil_load_weights(sn_ctx *c, const float *blob, size_t n_floats, const sn_param_desc *descs, int n_params)
{
    if (!c || !blob || !descs) return fail(SN_ERR_ARG, "null argument");
    if (n_params != kSimParams) return fail(SN_ERR_ARG, "similarityNet has %d parameter arrays, got %d", kSimParams, n_params);
    HIPCHK(hipSetDevice(c->device));
    auto count = [](const sn_param_desc &d) { size_t n = 1; for (int i = 0; i < d.ndim; ++i) n *= (size_t)d.shape[i]; return n; };
    for (int i = 0; i < n_params; ++i)
        if (descs[i].ndim < 1 || descs[i].ndim > 5 || descs[i].offset < 0 || (size_t)descs[i].offset + count(descs[i]) > n_floats)
            return fail(SN_ERR_ARG, "param %d: bad descriptor", i);
    for (int i = 0; i < 13; ++i) {
        if (!shape_is(descs[2 * i], {kSimC[i + 1], kSimC[i], 3, 3})) return fail(SN_ERR_ARG, "%s: W must be (%d,%d,3,3)", kSimName[i], kSimC[i + 1], kSimC[i]);
        if (!shape_is(descs[2 * i + 1], {kSimC[i + 1]})) return fail(SN_ERR_ARG, "%s: b must be (%d,)", kSimName[i], kSimC[i + 1]);
    }
    if (!shape_is(descs[26], {kSimilFeat, kEmb}) || !shape_is(descs[27], {kEmb})) return fail(SN_ERR_ARG, "embedding: W must be (%d,%d), b (%d,)", kSimilFeat, kEmb, kEmb);
    if (!shape_is(descs[28], {1, 1}) || !shape_is(descs[29], {1})) return fail(SN_ERR_ARG, "similarity: W must be (1,1), b (1,)");
    c->simil_host.assign(blob, blob + n_floats);
    c->simil_descs.assign(descs, descs + n_params);
    c->simil_split = -1; c->simil_loaded = false;
    int rc;
    if (!c->semb_W) { if ((rc = dev_alloc(c, &c->semb_W, (size_t)kSimilFeat * kEmb)) != SN_OK) return rc; if ((rc = dev_alloc(c, &c->semb_b, kEmb)) != SN_OK) return rc; }
    HIPCHK(hipMemcpy(c->semb_W, blob + descs[26].offset, sizeof(float) * kSimilFeat * kEmb, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->semb_b, blob + descs[27].offset, sizeof(float) * kEmb, hipMemcpyHostToDevice));
    c->ssim_w = blob[descs[28].offset]; c->ssim_b = blob[descs[29].offset];
    if ((rc = simil_pack(c)) != SN_OK) return rc;
    c->simil_loaded = true;
    return SN_OK;
}

// Workspace of one chunk: every tensor [C/8][cap][H][H][8] fp16 (+ second plane right behind), carved from one allocation.
struct SimilWs {
    Act p0, a[5][2], pool[5];
    float *feat, *emb, *part; double *centers; unsigned char *patches;
};
static size_t simil_carve(sn_ctx *c, int cap, int npl, SimilWs *w)
{
    size_t off = 0;
    char *base = static_cast<char *>(c->sws);
    auto act = [&](int ch, int H) {
        const size_t halfs = (size_t)ch * H * H * cap;
        Act t{base ? reinterpret_cast<_Float16 *>(base + off) : nullptr, (long long)halfs};
        off += halfs * 2 * npl;
        off = (off + 255) / 256 * 256;
        return t;
    };
    static const int C[5] = {64, 128, 256, 512, 512};
    SimilWs t;
    t.p0 = act(8, kPatch);
    for (int st = 0; st < 5; ++st) {
        const int H = kPatch >> st;
        t.a[st][0] = act(C[st], H); t.a[st][1] = act(C[st], H); t.pool[st] = act(C[st], H / 2);
    }
    auto raw = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; };
    t.feat = reinterpret_cast<float *>(raw((size_t)cap * kSimilFeat * 4));
    t.emb = reinterpret_cast<float *>(raw((size_t)cap * kEmb * 4));
    t.part = reinterpret_cast<float *>(raw((size_t)cap * kEmb * 4 * kDenseKS));
    t.centers = reinterpret_cast<double *>(raw((size_t)cap * 2 * 8));
    t.patches = reinterpret_cast<unsigned char *>(raw((size_t)cap * kPatch * kPatch * 3));
    if (w) *w = t;
    return off;
}
static int simil_workspace(sn_ctx *c, int n, SimilWs *w)
{
    const int cap = std::min(std::max(n, 8), kSimChunk), npl = simil_mode(c) ? 2 : 1;
    if (!c->sws || c->sws_n < cap || c->sws_split != npl) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->sws) dev_free_owned(c, c->sws);
        c->sws = nullptr; c->sws_n = 0;
        const size_t bytes = simil_carve(c, cap, npl, nullptr);
        unsigned char *p = nullptr;
        int rc = dev_alloc(c, &p, bytes);
        if (rc != SN_OK) return rc;
        c->sws = p; c->sws_bytes = bytes; c->sws_n = cap; c->sws_split = npl;
    }
    simil_carve(c, c->sws_n, npl, w);
    return SN_OK;
}

extern "C++" {
template <int SP>
static int run_simil_t(sn_ctx *c, const SimilWs &w, int n)
{
    int rc;
    Act cur = w.p0;
    int cur_cs = 8;
    int flip[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 13; ++i) {
        const int st = kSimStage[i], H = kPatch >> st;
        const bool last = (i == 12 || kSimStage[i + 1] != st);
        if (last) {      // conv + bias + ReLU + Pool2DLayer(2) in one kernel
            Act out = w.pool[st];
            bool done = false;
            if constexpr (SN_SIM_NF8 >= 2 && SP == 1) {
                if (simil_wide(i, 1)) { rc = launch_conv<SCONV8P>(c, c->sconv[i], cur, cur_cs, out, kSimC[i + 1], 0, kSimC[i + 1], nullptr, 1, H, n); done = true; }
            }
            if constexpr (SN_SIM_MF8 && SP == 1) {
                if (!done && simil_mf8(i, 1)) { rc = launch_conv<SCONV16P>(c, c->sconv[i], cur, cur_cs, out, kSimC[i + 1], 0, kSimC[i + 1], nullptr, 1, H, n); done = true; }
            }
            if (!done)
            rc = st == 4 ? launch_conv<SCONV5P>(c, c->sconv[i], cur, cur_cs, out, kSimC[i + 1], 0, kSimC[i + 1], nullptr, 1, H, n)
                         : launch_conv<SCONVP>(c, c->sconv[i], cur, cur_cs, out, kSimC[i + 1], 0, kSimC[i + 1], nullptr, 1, H, n);
            if (rc != SN_OK) return rc;
            cur = out; cur_cs = kSimC[i + 1];
            continue;
        }
        Act out = w.a[st][flip[st]];
        flip[st] ^= 1;
        if constexpr (SN_SIM_NF8 && SP == 1) {
            if (simil_wide(i, 1)) {
                rc = launch_conv<SCONV8>(c, c->sconv[i], cur, cur_cs, out, kSimC[i + 1], 0, kSimC[i + 1], nullptr, 1, H, n);
                if (rc != SN_OK) return rc;
                cur = out; cur_cs = kSimC[i + 1];
                continue;
            }
        }
        if constexpr (SN_SIM_MF8 && SP == 1) {
            if (simil_mf8(i, 1)) {
                rc = launch_conv<SCONV16>(c, c->sconv[i], cur, cur_cs, out, kSimC[i + 1], 0, kSimC[i + 1], nullptr, 1, H, n);
                if (rc != SN_OK) return rc;
                cur = out; cur_cs = kSimC[i + 1];
                continue;
            }
        }
        rc = st == 4 ? launch_conv<SCONV5>(c, c->sconv[i], cur, cur_cs, out, kSimC[i + 1], 0, kSimC[i + 1], nullptr, 1, H, n)
                     : launch_conv<SCONV>(c, c->sconv[i], cur, cur_cs, out, kSimC[i + 1], 0, kSimC[i + 1], nullptr, 1, H, n);
        if (rc != SN_OK) return rc;
        cur = out; cur_cs = kSimC[i + 1];
    }
    {
        SimilFeatArgs fa;
        for (int k = 0; k < 5; ++k) { fa.pool[k] = w.pool[k].p; fa.lo_off[k] = w.pool[k].lo; }
        fa.feat = w.feat; fa.n = n;
        ProfScope ps(c, "s_features", 0, (double)n * kSimilFeat * (2.0 * (SP ? 2 : 1) + 4.0));
        hipLaunchKernelGGL(simil_features_kernel<SP>, dim3((unsigned)n), dim3(256), 0, c->stream, fa);
        HIPCHK(hipGetLastError());
    }
    {
        ProfScope ps(c, "s_dense", 2.0 * n * kSimilFeat * kEmb, (double)n * (kSimilFeat + kEmb) * 4.0 + (double)kSimilFeat * kEmb * 4.0);
        hipLaunchKernelGGL(simil_dense_kernel, dim3((unsigned)((n + 31) / 32), kDenseKS), dim3(256), 0, c->stream, w.feat, c->semb_W, w.part, n);
        hipLaunchKernelGGL(simil_dense_reduce_kernel, dim3((unsigned)((n * kEmb + 255) / 256)), dim3(256), 0, c->stream, w.part, c->semb_b, w.emb, n);
        HIPCHK(hipGetLastError());
    }
    return SN_OK;
}
}   // extern "C++"
static int run_simil(sn_ctx *c, const SimilWs &w, int n) { return simil_mode(c) ? run_simil_t<1>(c, w, n) : run_simil_t<0>(c, w, n); }

static int simil_ready(sn_ctx *c)
{
    if (!c->simil_loaded) return fail(SN_ERR_STATE, "sn_simil_load_weights has not been called");
    return simil_pack(c);     // re-packs after a precision switch
}

static int launch_crop(sn_ctx *c, int view, int n, const double *ch_dev, const double *cw_dev, unsigned char *patches_dev, Act p0,
                       const float *mean_bgr)
{
    const long long total = (long long)n * kPatch * kPatch;
    const uint8_t *img = c->img_base + c->h_img_off[view];
    const float mb = mean_bgr ? mean_bgr[0] : 0.f, mg = mean_bgr ? mean_bgr[1] : 0.f, mr = mean_bgr ? mean_bgr[2] : 0.f;
    const int sp = simil_mode(c);
    ProfScope ps(c, "patch_crop", 0, (double)total * (3.0 + (patches_dev ? 3.0 : 0.0) + (p0.p ? 16.0 * (sp ? 2 : 1) : 0.0)));
    if (sp) hipLaunchKernelGGL(patch_crop_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, img, c->h_img_h[view], c->h_img_w[view],
                               ch_dev, cw_dev, n, patches_dev, p0.p, p0.lo, mb, mg, mr);
    else hipLaunchKernelGGL(patch_crop_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, img, c->h_img_h[view], c->h_img_w[view],
                            ch_dev, cw_dev, n, patches_dev, p0.p, p0.lo, mb, mg, mr);
    HIPCHK(hipGetLastError());
    return SN_OK;
}

static int check_view(sn_ctx *c, int view)
{
    if (!c->img_base) return fail(SN_ERR_STATE, "sn_set_images must be called first");
    if (view < 0 || view >= c->V_img) return fail(SN_ERR_ARG, "view %d out of range for %d images", view, c->V_img);
    return SN_OK;
}

extern "C" int sn_crop_patches(sn_ctx *c, int view, int n, const double *center_h, const double *center_w, unsigned char *patches)
{
    if (!c || !center_h || !center_w || !patches) return fail(SN_ERR_ARG, "null argument");
    if (n < 0) return fail(SN_ERR_ARG, "bad n");
    if (n == 0) return SN_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = check_view(c, view)) != SN_OK) return rc;
    SimilWs w;
    for (int i0 = 0; i0 < n; i0 += kSimChunk) {
        const int m = std::min(kSimChunk, n - i0);
        if ((rc = simil_workspace(c, m, &w)) != SN_OK) return rc;
        HIPCHK(hipMemcpyAsync(w.centers, center_h + i0, sizeof(double) * m, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(w.centers + c->sws_n, center_w + i0, sizeof(double) * m, hipMemcpyHostToDevice, c->stream));
        if ((rc = launch_crop(c, view, m, w.centers, w.centers + c->sws_n, w.patches, Act{nullptr, 0}, nullptr)) != SN_OK) return rc;
        HIPCHK(hipMemcpyAsync(patches + (size_t)i0 * kPatch * kPatch * 3, w.patches, (size_t)m * kPatch * kPatch * 3, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return SN_OK;
}

extern "C" int sn_patch2embedding(sn_ctx *c, int n, const float *patches, float *embeddings)
{
    if (!c || !patches || !embeddings) return fail(SN_ERR_ARG, "null argument");
    if (n < 0) return fail(SN_ERR_ARG, "bad n");
    if (n == 0) return SN_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = simil_ready(c)) != SN_OK) return rc;
    TmpDev t;
    const size_t per = (size_t)3 * kPatch * kPatch;
    float *d_x = t.get<float>(per * std::min(n, kSimChunk));
    if (!d_x) return fail(SN_ERR_NOMEM, "sn_patch2embedding: device allocation failed");
    SimilWs w;
    for (int i0 = 0; i0 < n; i0 += kSimChunk) {
        const int m = std::min(kSimChunk, n - i0);
        if ((rc = simil_workspace(c, m, &w)) != SN_OK) return rc;
        HIPCHK(hipMemcpyAsync(d_x, patches + (size_t)i0 * per, sizeof(float) * per * m, hipMemcpyHostToDevice, c->stream));
        {
            const long long total = (long long)m * kPatch * kPatch;
            ProfScope ps(c, "nchw_to_p0", 0, (double)total * (12.0 + 16.0 * (simil_mode(c) ? 2 : 1)));
            if (simil_mode(c)) hipLaunchKernelGGL(nchw_to_p0_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_x, m, w.p0.p, w.p0.lo);
            else hipLaunchKernelGGL(nchw_to_p0_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_x, m, w.p0.p, w.p0.lo);
            HIPCHK(hipGetLastError());
        }
        if ((rc = run_simil(c, w, m)) != SN_OK) return rc;
        HIPCHK(hipMemcpyAsync(embeddings + (size_t)i0 * kEmb, w.emb, sizeof(float) * kEmb * m, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return SN_OK;
}

extern "C" int sn_crop_embed(sn_ctx *c, int view, int n, const double *center_h, const double *center_w, const float *mean_bgr, float *embeddings)
{
    if (!c || !center_h || !center_w || !mean_bgr || !embeddings) return fail(SN_ERR_ARG, "null argument");
    if (n < 0) return fail(SN_ERR_ARG, "bad n");
    if (n == 0) return SN_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = check_view(c, view)) != SN_OK) return rc;
    if ((rc = simil_ready(c)) != SN_OK) return rc;
    // centres up and embeddings down ONCE per call: the chunks of a view (62 for a DTU image) run back to back on the stream instead of
    // each waiting for two uploads, a download and a host synchronisation (7 % of the early-rejection stage)
    const size_t need = (size_t)n * (2 * sizeof(double) + kEmb * sizeof(float)) + 256;
    if (c->sview_bytes < need) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->sview) dev_free_owned(c, c->sview);
        c->sview = nullptr; c->sview_bytes = 0;
        unsigned char *p = nullptr;
        const size_t cap = need + need / 4;
        if ((rc = dev_alloc(c, &p, cap)) != SN_OK) return rc;
        c->sview = p; c->sview_bytes = cap;
    }
    double *d_ch = static_cast<double *>(c->sview), *d_cw = d_ch + n;
    float *d_emb = reinterpret_cast<float *>(d_cw + n);
    HIPCHK(hipMemcpyAsync(d_ch, center_h, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_cw, center_w, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    SimilWs w;
    for (int i0 = 0; i0 < n; i0 += kSimChunk) {
        const int m = std::min(kSimChunk, n - i0);
        if ((rc = simil_workspace(c, m, &w)) != SN_OK) return rc;
        w.emb = d_emb + (size_t)i0 * kEmb;                       // the embedding kernels of this chunk write straight into the call's buffer
        if ((rc = launch_crop(c, view, m, d_ch + i0, d_cw + i0, nullptr, w.p0, mean_bgr)) != SN_OK) return rc;
        if ((rc = run_simil(c, w, m)) != SN_OK) return rc;
    }
    HIPCHK(hipMemcpyAsync(embeddings, d_emb, sizeof(float) * kEmb * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SN_OK;
}

extern "C" int sn_embeddingpair2simil(sn_ctx *c, int n_pairs, const float *emb_pairs, float *similarity)
{
    if (!c || !emb_pairs || !similarity) return fail(SN_ERR_ARG, "null argument");
    if (n_pairs < 0) return fail(SN_ERR_ARG, "bad n_pairs");
    if (n_pairs == 0) return SN_OK;
    if (!c->simil_loaded) return fail(SN_ERR_STATE, "sn_simil_load_weights has not been called");
    HIPCHK(hipSetDevice(c->device));
    TmpDev t;
    float *d_e = t.get<float>((size_t)2 * n_pairs * kEmb), *d_s = t.get<float>(n_pairs);
    if (!d_e || !d_s) return fail(SN_ERR_NOMEM, "sn_embeddingpair2simil: device allocation failed");
    HIPCHK(hipMemcpyAsync(d_e, emb_pairs, sizeof(float) * 2 * n_pairs * kEmb, hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, "pair_simil", 0, (double)n_pairs * (2.0 * kEmb + 1.0) * 4.0);
        hipLaunchKernelGGL(pair_simil_kernel, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), 0, c->stream, d_e, d_s, n_pairs, c->ssim_w, c->ssim_b);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(similarity, d_s, sizeof(float) * n_pairs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SN_OK;
}

extern "C" int sn_embeddings2simil(sn_ctx *c, int n_cubes, int n_views, const float *embeddings, float *similarity)
{
    if (!c || !embeddings || !similarity) return fail(SN_ERR_ARG, "null argument");
    if (n_cubes < 0 || n_views < 2) return fail(SN_ERR_ARG, "need n_cubes >= 0 and n_views >= 2");
    if (n_cubes == 0) return SN_OK;
    if (!c->simil_loaded) return fail(SN_ERR_STATE, "sn_simil_load_weights has not been called");
    HIPCHK(hipSetDevice(c->device));
    const int P = n_views * (n_views - 1) / 2;
    std::vector<int> pairs;
    pairs.reserve(2 * (size_t)P);
    for (int i = 0; i < n_views; ++i)
        for (int j = i + 1; j < n_views; ++j) { pairs.push_back(i); pairs.push_back(j); }      // itertools.combinations order
    TmpDev t;
    const size_t ne = (size_t)n_cubes * n_views * kEmb, ns = (size_t)n_cubes * P;
    float *d_e = t.get<float>(ne), *d_s = t.get<float>(ns);
    int *d_p = t.get<int>(pairs.size());
    if (!d_e || !d_s || !d_p) return fail(SN_ERR_NOMEM, "sn_embeddings2simil: device allocation failed");
    HIPCHK(hipMemcpyAsync(d_e, embeddings, sizeof(float) * ne, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_p, pairs.data(), sizeof(int) * pairs.size(), hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, "pair_simil_all", 0, (double)ns * (2.0 * kEmb + 1.0) * 4.0);
        hipLaunchKernelGGL(pair_simil_all_kernel, dim3((unsigned)((ns + 3) / 4)), dim3(256), 0, c->stream, d_e, d_p, d_s, (long long)ns, n_views, P,
                           c->ssim_w, c->ssim_b);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(similarity, d_s, sizeof(float) * ns, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SN_OK;
}
