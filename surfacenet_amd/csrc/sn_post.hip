// sn_post.hip — C ABI of the post-pass: ray pooling + dense2sparse over postpass.h.
#include "sn_internal.h"
#include "postpass.h"

// ---- post-pass: ray pooling + dense2sparse --------------------------------------------------------
static int check_pairs_cam(sn_ctx *c, long long count, const int64_t *pairs, std::vector<int64_t> &wrapped)
{
    wrapped.assign(pairs, pairs + count);
    for (long long i = 0; i < count; ++i) {
        int64_t v = wrapped[i];
        if (v < -(int64_t)c->V_cam || v >= (int64_t)c->V_cam)
            return fail(SN_ERR_ARG, "view index %lld out of range for %d cameras (the reference raises IndexError here)", (long long)v, c->V_cam);
        if (v < 0) wrapped[i] = v + c->V_cam;
    }
    return SN_OK;
}

static int launch_ray_pool(sn_ctx *c, int n, int n_vp, const int64_t *pairs_dev, const float *xyz_dev, const float *resol_dev,
                           const float *pred_dev, int use_thresh, float min_prob, unsigned char *votes_dev)
{
    if (!c->cams) return fail(SN_ERR_STATE, "sn_set_cameras must be called before ray pooling");
    if (2 * n_vp > 255) return fail(SN_ERR_ARG, "2*n_vp = %d votes do not fit the uint8 result", 2 * n_vp);
    const size_t s3 = (size_t)c->s * c->s * c->s;
    size_t cap = 64;
    while (cap < 2 * s3) cap <<= 1;
    const size_t per_wg = cap * (8 + 8 + 8 + 4) + s3 * 4;
    const int E = 2 * n_vp;
    const size_t budget = (size_t)1 << 30;                      // hash-table workspace per launch
    int cubes = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, budget / (per_wg * E)));
    const size_t need = per_wg * E * cubes;
    if (!c->d_err) { int rc = dev_alloc(c, &c->d_err, 1); if (rc != SN_OK) return rc; HIPCHK(hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream)); }
    if (c->rp_ws_bytes < need) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->rp_ws) dev_free_owned(c, c->rp_ws);
        c->rp_ws = nullptr; c->rp_ws_bytes = 0;
        unsigned char *w = nullptr;
        int rc = dev_alloc(c, &w, need);
        if (rc != SN_OK) return rc;
        c->rp_ws = w; c->rp_ws_bytes = need;
    }
    HIPCHK(hipMemsetAsync(votes_dev, 0, (size_t)n * s3, c->stream));
    for (int i0 = 0; i0 < n; i0 += cubes) {
        const int m = std::min(cubes, n - i0);
        const size_t wgs = (size_t)m * E;
        RayPoolArgs a;
        memset(&a, 0, sizeof a);
        a.pairs = pairs_dev + (size_t)i0 * E; a.xyz = xyz_dev + 3 * (size_t)i0; a.resol = resol_dev + i0; a.cams = c->cams;
        a.pred = pred_dev + (size_t)i0 * s3; a.votes = votes_dev + (size_t)i0 * s3;
        unsigned char *w = static_cast<unsigned char *>(c->rp_ws);
        a.pix_key = reinterpret_cast<unsigned long long *>(w); w += wgs * cap * 8;
        a.pix_best = reinterpret_cast<unsigned long long *>(w); w += wgs * cap * 8;
        a.cell_key = reinterpret_cast<unsigned long long *>(w); w += wgs * cap * 8;
        a.cell_idx = reinterpret_cast<unsigned *>(w); w += wgs * cap * 4;
        a.cslot = reinterpret_cast<unsigned *>(w);
        a.err = c->d_err;
        a.n_vp = n_vp; a.s = c->s; a.V = c->V_cam; a.cap_max = (int)cap; a.use_thresh = use_thresh; a.thresh = min_prob;
        // algorithmic bytes: the prediction cube is read once per distinct view (<= E), the votes are written once
        ProfScope ps(c, "ray_pool", 0, (double)m * s3 * (4.0 * E + 1.0));
        hipLaunchKernelGGL(ray_pool_kernel, dim3((unsigned)E, (unsigned)m), dim3(RP_NT), 0, c->stream, a);
        HIPCHK(hipGetLastError());
    }
    return SN_OK;
}

extern "C" int sn_ray_pool_dev(sn_ctx *c, int n, int n_vp, const int64_t *pairs_dev, const float *xyz_dev, const float *resol_dev,
                               const float *pred_dev, int use_thresh, float min_prob, unsigned char *votes_dev)
{
    if (!c || !pairs_dev || !xyz_dev || !resol_dev || !pred_dev || !votes_dev) return fail(SN_ERR_ARG, "null argument");
    if (n < 1 || n_vp < 1) return fail(SN_ERR_ARG, "bad n / n_vp");
    HIPCHK(hipSetDevice(c->device));
    return launch_ray_pool(c, n, n_vp, pairs_dev, xyz_dev, resol_dev, pred_dev, use_thresh, min_prob, votes_dev);
}

static int sparse_geometry(sn_ctx *c, const sn_sparse_cfg *cfg, int &lo, int &dc)
{
    lo = 0; dc = c->s;
    if (cfg->enable_centerCrop) {
        if (cfg->cube_Dcenter < 1 || cfg->cube_Dcenter > c->s) return fail(SN_ERR_ARG, "cube_Dcenter %d not in [1,%d]", cfg->cube_Dcenter, c->s);
        lo = (c->s - cfg->cube_Dcenter) / 2; dc = cfg->cube_Dcenter;
    }
    if (dc > 256) return fail(SN_ERR_ARG, "voxel indices are uint8 (utils/sparseCubes.py:68): cube edge %d > 256", dc);
    return SN_OK;
}

extern "C" int sn_dense2sparse_dev(sn_ctx *c, int n, int n_vp, const int64_t *pairs_dev, const float *xyz_dev, const float *resol_dev,
                                   const float *pred_dev, const unsigned char *rgb_dev, const sn_sparse_cfg *cfg,
                                   unsigned char *votes_ws_dev, int64_t *offsets_dev, unsigned char *ijk_dev, uint16_t *pred16_dev,
                                   unsigned char *rgb_out_dev, unsigned char *votes_out_dev)
{
    if (!c || !cfg || !pred_dev || !offsets_dev || !ijk_dev || !pred16_dev) return fail(SN_ERR_ARG, "null argument");
    if (n < 1 || n_vp < 1) return fail(SN_ERR_ARG, "bad n / n_vp");
    if (rgb_out_dev && !rgb_dev) return fail(SN_ERR_ARG, "rgb_out requested without rgb");
    HIPCHK(hipSetDevice(c->device));
    int lo, dc, rc;
    if ((rc = sparse_geometry(c, cfg, lo, dc)) != SN_OK) return rc;
    const bool by_votes = cfg->enable_rayPooling && cfg->rayPool_thresh != 0;      // sparseCubes.py:60-62
    const bool need_votes = cfg->enable_rayPooling && (by_votes || votes_out_dev);
    if (need_votes) {
        if (!pairs_dev || !xyz_dev || !resol_dev || !votes_ws_dev) return fail(SN_ERR_ARG, "ray pooling needs view pairs, xyz, resol and the votes scratch");
        if ((rc = launch_ray_pool(c, n, n_vp, pairs_dev, xyz_dev, resol_dev, pred_dev, 1, cfg->min_prob, votes_ws_dev)) != SN_OK) return rc;
    }
    if (c->d_counts_cap < n) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->d_counts) dev_free_owned(c, c->d_counts);
        c->d_counts = nullptr; c->d_counts_cap = 0;
        if ((rc = dev_alloc(c, &c->d_counts, (size_t)n)) != SN_OK) return rc;
        c->d_counts_cap = n;
    }
    SparseArgs a;
    memset(&a, 0, sizeof a);
    a.pred = pred_dev; a.rgb = rgb_out_dev ? rgb_dev : nullptr; a.votes = need_votes ? votes_ws_dev : nullptr;
    a.offsets = reinterpret_cast<long long *>(offsets_dev); a.counts = c->d_counts;
    a.ijk = ijk_dev; a.pred16 = pred16_dev; a.rgb_out = rgb_out_dev; a.votes_out = votes_out_dev;
    a.s = c->s; a.lo = lo; a.dc = dc;
    a.by_votes = by_votes ? 1 : 0; a.vote_thresh = cfg->rayPool_thresh; a.min_prob = cfg->min_prob;
    const double vox = (double)n * dc * dc * dc;
    // algorithmic bytes: the keep rule reads 4 B (pred) or 1 B (votes) per voxel twice (count + write); kept voxels add <= 9 B
    ProfScope ps(c, "dense2sparse", 0, vox * (by_votes ? 1.0 : 4.0) * 2.0);
    hipLaunchKernelGGL(d2s_count_kernel, dim3((unsigned)n), dim3(D2S_NT), 0, c->stream, a);
    hipLaunchKernelGGL(d2s_scan_kernel, dim3(1), dim3(64), 0, c->stream, c->d_counts, a.offsets, n);
    hipLaunchKernelGGL(d2s_write_kernel, dim3((unsigned)n), dim3(D2S_NT), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    return SN_OK;
}

// Host-array forms: stage through temporary device buffers, synchronous.
extern "C" int sn_ray_pool(sn_ctx *c, int n, int n_vp, const int64_t *pairs, const float *xyz, const float *resol, const float *pred,
                           int use_thresh, float min_prob, unsigned char *votes)
{
    if (!c || !pairs || !xyz || !resol || !pred || !votes) return fail(SN_ERR_ARG, "null argument");
    if (n < 0 || n_vp < 1) return fail(SN_ERR_ARG, "bad n / n_vp");
    if (n == 0) return SN_OK;
    if (!c->cams) return fail(SN_ERR_STATE, "sn_set_cameras must be called before ray pooling");
    HIPCHK(hipSetDevice(c->device));
    std::vector<int64_t> wp;
    int rc;
    if ((rc = check_pairs_cam(c, (long long)n * n_vp * 2, pairs, wp)) != SN_OK) return rc;
    const size_t s3 = (size_t)c->s * c->s * c->s;
    TmpDev t;
    int64_t *d_p = t.get<int64_t>((size_t)n * n_vp * 2); float *d_x = t.get<float>(3 * (size_t)n), *d_r = t.get<float>(n), *d_pr = t.get<float>(n * s3);
    unsigned char *d_v = t.get<unsigned char>(n * s3);
    if (!d_p || !d_x || !d_r || !d_pr || !d_v) return fail(SN_ERR_NOMEM, "sn_ray_pool: device allocation failed");
    HIPCHK(hipMemcpyAsync(d_p, wp.data(), sizeof(int64_t) * 2 * n * n_vp, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_x, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_r, resol, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_pr, pred, sizeof(float) * n * s3, hipMemcpyHostToDevice, c->stream));
    if ((rc = launch_ray_pool(c, n, n_vp, d_p, d_x, d_r, d_pr, use_thresh, min_prob, d_v)) != SN_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    HIPCHK(hipMemcpyAsync(votes, d_v, n * s3, hipMemcpyDeviceToHost, c->stream));
    return sn_synchronize(c);
}

extern "C" int sn_dense2sparse(sn_ctx *c, int n, int n_vp, const int64_t *pairs, const float *xyz, const float *resol, const float *pred,
                               const unsigned char *rgb, const sn_sparse_cfg *cfg, int64_t *offsets, unsigned char *ijk,
                               uint16_t *pred16, unsigned char *rgb_out, unsigned char *votes_out)
{
    if (!c || !cfg || !pred || !offsets || !ijk || !pred16) return fail(SN_ERR_ARG, "null argument");
    if (rgb_out && !rgb) return fail(SN_ERR_ARG, "rgb_out requested without rgb");
    if (n < 0 || n_vp < 1) return fail(SN_ERR_ARG, "bad n / n_vp");
    if (n == 0) { offsets[0] = 0; return SN_OK; }
    HIPCHK(hipSetDevice(c->device));
    int lo, dc, rc;
    if ((rc = sparse_geometry(c, cfg, lo, dc)) != SN_OK) return rc;
    const size_t s3 = (size_t)c->s * c->s * c->s, cap = (size_t)n * dc * dc * dc;
    TmpDev t;
    int64_t *d_p = nullptr; float *d_x = nullptr, *d_r = nullptr; unsigned char *d_vws = nullptr;
    std::vector<int64_t> wp;
    if (cfg->enable_rayPooling) {
        if (!pairs || !xyz || !resol) return fail(SN_ERR_ARG, "ray pooling needs view pairs, xyz and resol");
        if (!c->cams) return fail(SN_ERR_STATE, "sn_set_cameras must be called before ray pooling");
        if ((rc = check_pairs_cam(c, (long long)n * n_vp * 2, pairs, wp)) != SN_OK) return rc;
        d_p = t.get<int64_t>((size_t)n * n_vp * 2); d_x = t.get<float>(3 * (size_t)n); d_r = t.get<float>(n); d_vws = t.get<unsigned char>(n * s3);
        if (!d_p || !d_x || !d_r || !d_vws) return fail(SN_ERR_NOMEM, "sn_dense2sparse: device allocation failed");
        HIPCHK(hipMemcpyAsync(d_p, wp.data(), sizeof(int64_t) * 2 * n * n_vp, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_x, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_r, resol, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    }
    float *d_pr = t.get<float>(n * s3);
    unsigned char *d_rgb = rgb_out ? t.get<unsigned char>(3 * n * s3) : nullptr;
    int64_t *d_off = t.get<int64_t>((size_t)n + 1);
    unsigned char *d_ijk = t.get<unsigned char>(3 * cap), *d_ro = rgb_out ? t.get<unsigned char>(3 * cap) : nullptr;
    unsigned char *d_vo = (votes_out && cfg->enable_rayPooling) ? t.get<unsigned char>(cap) : nullptr;
    uint16_t *d_p16 = t.get<uint16_t>(cap);
    if (!d_pr || !d_off || !d_ijk || !d_p16 || (rgb_out && (!d_rgb || !d_ro)) || (votes_out && cfg->enable_rayPooling && !d_vo))
        return fail(SN_ERR_NOMEM, "sn_dense2sparse: device allocation failed");
    HIPCHK(hipMemcpyAsync(d_pr, pred, sizeof(float) * n * s3, hipMemcpyHostToDevice, c->stream));
    if (rgb_out) HIPCHK(hipMemcpyAsync(d_rgb, rgb, 3 * n * s3, hipMemcpyHostToDevice, c->stream));
    rc = sn_dense2sparse_dev(c, n, n_vp, d_p, d_x, d_r, d_pr, d_rgb, cfg, d_vws, d_off, d_ijk, d_p16, d_ro, d_vo);
    if (rc != SN_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    HIPCHK(hipMemcpyAsync(offsets, d_off, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, c->stream));
    if ((rc = sn_synchronize(c)) != SN_OK) return rc;
    const size_t total = (size_t)offsets[n];
    if (total) {
        HIPCHK(hipMemcpyAsync(ijk, d_ijk, 3 * total, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(pred16, d_p16, 2 * total, hipMemcpyDeviceToHost, c->stream));
        if (rgb_out) HIPCHK(hipMemcpyAsync(rgb_out, d_ro, 3 * total, hipMemcpyDeviceToHost, c->stream));
        if (d_vo) HIPCHK(hipMemcpyAsync(votes_out, d_vo, total, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return SN_OK;
}

// camera.perspectiveProj (utils/camera.py:123-184): V cameras x n points in one launch (see project_points_kernel).
extern "C" int sn_project_points(sn_ctx *c, int V, const double *P, int n, const double *xyz, int round_int, double *img_h, double *img_w,
                                 double *depth)
{
    if (!c || !xyz || !img_h || !img_w) return fail(SN_ERR_ARG, "null argument");
    if (n < 0) return fail(SN_ERR_ARG, "n must be >= 0");
    if (!P) {
        if (!c->cams) return fail(SN_ERR_STATE, "P == NULL selects the cameras of sn_set_cameras, which has not been called");
        V = c->V_cam;
    } else if (V < 1) return fail(SN_ERR_ARG, "V must be >= 1");
    if (n == 0) return SN_OK;
    HIPCHK(hipSetDevice(c->device));
    const size_t tot = (size_t)V * n;
    TmpDev t;
    double *d_x = t.get<double>((size_t)n * 3), *d_h = t.get<double>(tot), *d_w = t.get<double>(tot), *d_d = depth ? t.get<double>(tot) : nullptr;
    double *d_P = P ? t.get<double>((size_t)V * 12) : c->cams;
    if (!d_x || !d_h || !d_w || !d_P || (depth && !d_d)) return fail(SN_ERR_NOMEM, "sn_project_points: device allocation failed");
    if (P) HIPCHK(hipMemcpyAsync(d_P, P, sizeof(double) * 12 * V, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_x, xyz, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, "project_points", 0, (double)n * 24.0 + (double)tot * (depth ? 24.0 : 16.0));
        hipLaunchKernelGGL(project_points_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)V), dim3(256), 0, c->stream, d_P, d_x, n, round_int,
                           d_h, d_w, d_d);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(img_h, d_h, sizeof(double) * tot, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(img_w, d_w, sizeof(double) * tot, hipMemcpyDeviceToHost, c->stream));
    if (depth) HIPCHK(hipMemcpyAsync(depth, d_d, sizeof(double) * tot, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SN_OK;
}

// ---- box-speed probe (include/surfacenet_hip.h: sn_mfma_probe) ------------------------------------------------------------------------
// A pure v_mfma_f32_16x16x32_f16 stream, operands in registers, 14 independent accumulators (no dependent-issue stalls), one wave per SIMD on
// every CU: what the chip SUSTAINS (time, not clocks) - tools/probe/power_probe.hip's first row, inside the library so that every bench line can
// carry its box's speed class.
typedef _Float16 probe_half8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) mfma_probe_kernel(float *out, int iters, unsigned long long *clk)
{
    const int lane = threadIdx.x & 63;
    probe_half8 a[4], b[4];
    unsigned st = 12345u + 747796405u * (unsigned)(threadIdx.x + 1);
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 8; ++e) {      // random operands in [-1, 1]: zero (or constant) operands draw less power and clock higher
            st = st * 1664525u + 1013904223u; a[j][e] = (_Float16)(((int)(st >> 16) % 2001 - 1000) * 0.001f);
            st = st * 1664525u + 1013904223u; b[j][e] = (_Float16)(((int)(st >> 16) % 2001 - 1000) * 0.001f);
        }
    probe_f32x4 acc[14] = {};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 56; ++q)      // tied accumulators in the AGPR file: through the builtin hipcc rotates the 14 tuples through the file (80 register moves per pass)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[q % 14]) : "v"(a[q & 3]), "v"(b[(q >> 2) & 3]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // (inline-asm MFMAs: the wait states in front of the accumulator reads below are ours to keep)
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = (unsigned long long)(t1 - t0);
    float s = 0.f;
    for (int q = 0; q < 14; ++q) s += acc[q][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)lane;
}

extern "C" int sn_mfma_probe(sn_ctx *c, double target_ms, double *tflops, double *ghz)
{
    if (!c || !tflops) return fail(SN_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    if (!(target_ms > 0)) target_ms = 10.0;
    if (target_ms > 200.0) target_ms = 200.0;
    const int cus = c->num_cus > 0 ? c->num_cus : 256;
    float *out = nullptr;
    unsigned long long *clk = nullptr;
    int rc = dev_alloc(c, &out, (size_t)cus * 256);
    if (rc != SN_OK) return rc;
    rc = dev_alloc(c, &clk, 1);
    if (rc != SN_OK) { dev_free_owned(c, out); return rc; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); dev_free_owned(c, out); dev_free_owned(c, clk); };
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { cleanup(); return fail(SN_ERR_HIP, "hipEventCreate failed"); }
    // one wave-instruction = 2 * 16 * 16 * 32 FLOP; 4 waves per CU; ~16 clocks each at ~2 GHz: iterations of 56 MFMAs for target_ms
    const double flop_iter = 2.0 * 16 * 16 * 32 * 56.0 * 4 * cus;
    int iters = (int)(target_ms * 1e-3 * 2.0e9 / (56.0 * 16.1));
    if (iters < 100) iters = 100;
    float ms = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        hipError_t e = hipEventRecord(e0, c->stream);
        hipLaunchKernelGGL(mfma_probe_kernel, dim3(cus), dim3(256), 0, c->stream, out, iters, clk);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) { cleanup(); return fail(SN_ERR_HIP, "sn_mfma_probe: %s", hipGetErrorString(e)); }
    }
    unsigned long long h = 0;
    if (hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost) != hipSuccess) { cleanup(); return fail(SN_ERR_HIP, "sn_mfma_probe: copy failed"); }
    cleanup();
    if (!(ms > 0.f)) return fail(SN_ERR_HIP, "sn_mfma_probe: no elapsed time");
    *tflops = flop_iter * iters / (ms * 1e-3) / 1e12;
    if (ghz) *ghz = (double)h / (ms * 1e-3) / 1e9;
    return SN_OK;
}
