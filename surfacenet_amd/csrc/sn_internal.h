// sn_internal.h — shared by the translation units of libsurfacenet_hip.so (sn_api.hip: context, weights, hot path,
// RCCL, profiling; sn_post.hip: ray pooling + dense2sparse; sn_simil.hip: similarityNet + patch cropping): error state,
// the context, owned device memory, HIP-event profiling, packed conv layers and the launcher of conv3d_f16_mfma.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/surfacenet_hip.h"
#include "conv3d_mfma.h"
#include "cvc_warp.h"
#include "elementwise.h"

using namespace sn;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
inline thread_local std::string g_err;   // one per thread for the whole library (C++17 inline variable)

static int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return fail(SN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)


static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }


// A conv layer prepared for conv3d_f16_mfma: packed fp16 weight fragments + folded BN.
struct PackedConv {
    std::string name;
    int cin = 0, cout = 0, ks = 1, dil = 1, act = 0;
    int cin_p = 0;             // input channels padded to 8
    int nf = 0, nsplit = 1;    // 16-channel fragments per workgroup, workgroup columns
    int cs8max = 4, split = 0, k2d = 0;   // k2d: ks x ks taps over (y,z) only (2-D nets)
    int bridge = 0;            // in: bridge chunks wanted; out of pack_conv: granted (sn_api.hip pack_conv_host, conv3d_mfma.h write_koff_part)
    std::vector<unsigned char> slab_c8;
    long long wsplit_stride = 0;   // halfs
    _Float16 *wpack = nullptr;     // device
    float *scale = nullptr, *shift = nullptr;  // device, nsplit*nf*16
    double macs_per_voxel = 0;
    // 1x1x1 layers only: the normalised weights (cout x cin, after pack_conv's power-of-two scalings) and, when the layer is fused into
    // its producer's epilogue (EPI_SIDEPOOL), its A fragments in the producer's register order (device)
    std::vector<float> w_norm;
    _Float16 *side_frag = nullptr;
};


struct ProfRec { int tag; hipEvent_t e0, e1; double flops, bytes; };
struct ProfStat { std::string name; double ms = 0; int64_t launches = 0; double flops = 0, bytes = 0; };

// A/B switches of the kernels' schedule and packing (SN_NO_BRIDGE, SN_NO_EPI_FUSION, SN_M8_TAIL, SN_MX_S_*, ...): read from the environment by
// the TEST-ONLY twin library alone (Makefile target dbg, -DSN_DEBUG_HOOKS; tests/ and tools/ load it through SURFACENET_HIP_LIB). In the
// product library they do not exist: a stray variable cannot change the K order or the code format behind the caller's back.
static inline const char *sn_ab_switch(const char *name)
{
#ifdef SN_DEBUG_HOOKS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

struct sn_ctx {
    int device = 0, s = 32, max_samples = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;      // sn_memcpy_d2h_after: device-to-host copies that overlap later work on `stream`
    hipEvent_t marks[8] = {};               // sn_mark: points on `stream` those copies wait for
    // images / cameras
    int V_img = 0, V_cam = 0;
    uint8_t *img_base = nullptr; long long *img_off = nullptr; int *img_h = nullptr, *img_w = nullptr;
    double *cams = nullptr;
    // weights
    bool have_weights = false, have_relw = false;
    int split = 1;              // 0: f16 operands; 1: f16x3 (hi/lo split operands, fp32-class results) — default; 2: f16m8
    int mode = 1;               // the SN_PRECISION_* value given to sn_set_precision
    int mx_c4_e8 = kMxC4E8;     // ... of the fp8 code planes of the conv4 chain (conv3_3's, conv4_1's, conv4_2's outputs) when it runs in the f16m8e arithmetic (c4_m8)
    int c4_m8 = 1;              // default mode, round 5: conv4_1 .. conv4_3 with their correction terms on the fp8 e4m3 MX MFMA (2 MFMA units per product instead of 3)
    _Float16 *a3c = nullptr;    // code plane of conv3_3's output, [max_samples][160/8][D/4]^3 slots of 16 bytes
    int mx_act_e8 = kMxActE8, mx_cat_e8 = kMxCatE8;   // mx_format.h; SN_MX_S_ACT / SN_MX_S_CAT in the environment override them in the default (hybrid) mode
    int tail_m8 = 2;            // f16x3: 2 = merge_conv_a + merge_conv_b run their two correction terms on the MX MFMA (the default), 0 = none (f16x3p)
    bool ws_ready = false; int ws_split = -1;
    int last_run_samples = 0;   // samples of the last network run whose activations are still in the workspace (0: none since the weights / mode changed) - sn_calibrate_dev
    std::map<std::string, PackedConv> conv;
    float *w3 = nullptr; float scale3 = 0, shift3 = 0;
    void *zero_page = nullptr; int num_cus = 256;
    void *rccl_comm = nullptr; int comm_world = 0, comm_rank = 0;   // RCCL communicator (lazy dlopen of librccl)
    hipStream_t comm_stream = nullptr;                               // collectives that overlap the kernels (sn_allgather_f32_dev_overlap)
    hipEvent_t comm_ev[8] = {};                                      // ... their completion, per caller slot; comm_fork: "the kernels so far are done"
    hipEvent_t comm_fork = nullptr;
    unsigned char *comm_stage = nullptr; size_t comm_stage_cap = 0;  // sn_allgatherv_bytes_dev: padded payloads of all ranks
    unsigned char *comm_small = nullptr;                             // ... its 8-byte-per-rank exchanges (counts, status): allocated with the communicator
    float *relw_W1 = nullptr, *relw_scale = nullptr, *relw_shift = nullptr, *relw_w2 = nullptr; float relw_b2 = 0;
    // activation workspace (channels-last fp16)
    _Float16 *x0 = nullptr, *a1 = nullptr, *b1 = nullptr, *cat = nullptr, *p1 = nullptr, *a2 = nullptr, *b2 = nullptr,
             *p2 = nullptr, *a3 = nullptr, *b3 = nullptr, *a4 = nullptr, *b4 = nullptr, *s2 = nullptr, *s3 = nullptr,
             *s4 = nullptr, *ma = nullptr;
    std::vector<void *> ws_owned;
    float *unf_ws = nullptr;      // [max_samples][s^3]
    // batch parameter staging
    int64_t *d_pairs = nullptr; float *d_xyz = nullptr, *d_resol = nullptr, *d_w = nullptr;
    // host-API staging
    float *d_X = nullptr;         // [max_samples][6][s^3] fp32 NCDHW
    float *d_fused = nullptr;     // [max_samples][s^3]
    std::vector<long long> h_img_off; std::vector<int> h_img_h, h_img_w;   // host copies (patch cropping addresses one view)
    // similarityNet (N3)
    PackedConv sconv[13]; bool simil_loaded = false; int simil_split = -1;
    float *semb_W = nullptr, *semb_b = nullptr; float ssim_w = 0, ssim_b = 0;
    std::vector<float> simil_host;   // the 13 conv layers' fp32 parameters, kept to re-pack on a precision switch
    std::vector<sn_param_desc> simil_descs;
    void *sws = nullptr; size_t sws_bytes = 0; int sws_n = 0, sws_split = -1;
    void *sview = nullptr; size_t sview_bytes = 0;      // sn_crop_embed: centres and embeddings of one whole call (no per-chunk host round trip)
    // post-pass (ray pooling / dense2sparse) workspace
    unsigned *d_num = nullptr;    // numeric status word: bit i = conv layer i of the launch order stored a non-finite / fp16-overflowing value
    std::vector<std::string> num_names;   // layer name of each status bit
    void *rp_ws = nullptr; size_t rp_ws_bytes = 0; int *d_err = nullptr; int *d_counts = nullptr; int d_counts_cap = 0;
    std::vector<void *> owned;
    // profiling
    bool prof_on = false;
    std::vector<ProfRec> prof_recs;
    std::vector<ProfStat> prof_stats;
    std::map<std::string, int> prof_tags;
    std::vector<hipEvent_t> ev_pool;
};

template <typename T>
static int dev_alloc(sn_ctx *c, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) return fail(SN_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    c->owned.push_back(q);
    *p = static_cast<T *>(q);
    return SN_OK;
}

static int dev_free_owned(sn_ctx *c, void *p)
{
    if (!p) return SN_OK;
    auto it = std::find(c->owned.begin(), c->owned.end(), p);
    if (it != c->owned.end()) c->owned.erase(it);
    HIPCHK(hipFree(p));
    return SN_OK;
}

// ------------------------------------------------------------------------------------------------
// profiling: every launch goes through prof_begin / prof_end
// ------------------------------------------------------------------------------------------------
static hipEvent_t get_event(sn_ctx *c)
{
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    sn_ctx *c; ProfRec r; bool on;
    ProfScope(sn_ctx *ctx, const std::string &tag, double flops, double bytes) : c(ctx), on(ctx->prof_on)
    {
        if (!on) return;
        auto it = c->prof_tags.find(tag);
        int id;
        if (it == c->prof_tags.end()) {
            id = (int)c->prof_stats.size();
            c->prof_tags[tag] = id;
            ProfStat st; st.name = tag;
            c->prof_stats.push_back(st);
        } else id = it->second;
        r.tag = id; r.flops = flops; r.bytes = bytes;
        r.e0 = get_event(c); r.e1 = get_event(c);
        (void)hipEventRecord(r.e0, c->stream);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(r.e1, c->stream);
        c->prof_recs.push_back(r);
    }
};

static int prof_drain(sn_ctx *c)
{
    if (c->prof_recs.empty()) return SN_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &r : c->prof_recs) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, r.e0, r.e1));
        ProfStat &st = c->prof_stats[r.tag];
        st.ms += ms; st.launches += 1; st.flops += r.flops; st.bytes += r.bytes;
        c->ev_pool.push_back(r.e0); c->ev_pool.push_back(r.e1);
    }
    c->prof_recs.clear();
    return SN_OK;
}

// Folds BN, packs the weights of one conv layer into MFMA fragment order (defined in sn_api.hip).
int pack_conv(sn_ctx *c, PackedConv &L, const float *W, const float *beta, const float *gamma, const float *mean,
              const float *inv_std, int nf, int nsplit, int cs8max, int split, const int *in_exp = nullptr, const int *out_exp = nullptr);

static bool shape_is(const sn_param_desc &d, std::initializer_list<int> s)
{
    if (d.ndim != (int)s.size()) return false;
    int i = 0;
    for (int v : s) if (d.shape[i++] != v) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------
// A channels-last fp16 activation tensor: hi plane at p, lo plane (f16x3 mode) at p + lo elements.
struct Act { _Float16 *p; long long lo; long long code = 0; };      // code: a third plane of fp8 code slots (OSPLIT 4), in halfs from p

// What EPI_SIDEPOOL needs beside the conv's own arguments: the fused 1x1x1 layer and the two destinations.
struct SideFuse { const PackedConv *side; Act side_out; int side_cs, side_coff; Act pool_out; int pool_cs; };

template <int KS, int DIL, int MF, int NF, int EPI, int SPLIT, int CS8, int PCH, int NW, int PADV, int K2D = 0, int OSPLIT = -1>
static int launch_conv(sn_ctx *c, const PackedConv &L, Act in, int in_cs, Act out, int out_cs, int out_coff, int out_cp,
                       float *out_f32, int B, int D, int DX = 0, const SideFuse *sf = nullptr)
{
    using C = ConvCfg<KS, DIL, MF, NF, EPI, SPLIT, CS8, PCH, NW, PADV, K2D>;
    if (DX <= 0) DX = D;
    if ((L.k2d != 0) != (K2D != 0)) return fail(SN_ERR_STATE, "%s: packed for a different tap geometry", L.name.c_str());
    if (L.nf != NF || L.ks != KS || L.dil != DIL || L.cs8max != C::CS8MAX || L.split != SPLIT)
        return fail(SN_ERR_STATE, "%s: packed for a different kernel configuration", L.name.c_str());
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.in = in.p; a.in_lo_off = in.lo; a.out = out.p; a.out_lo_off = out.lo; a.out_code_off = out.code; a.out_f32 = out_f32;
    a.wpack = L.wpack; a.scale = L.scale; a.shift = L.shift;
    a.w3 = c->w3; a.scale3 = c->scale3; a.shift3 = c->shift3; a.zero_page = c->zero_page;
    a.wsplit_stride = L.wsplit_stride;
    a.in_cs = in_cs; a.out_cs = out_cs; a.out_coff = out_coff; a.out_cp = out_cp;
    a.D = D; a.DX = DX;
    a.tiles_x = (DX + C::TX - 1) / C::TX; a.tiles_y = (D + C::TY - 1) / C::TY; a.tiles_z = (D + C::TZ - 1) / C::TZ;
    a.total_tiles = B * a.tiles_x * a.tiles_y * a.tiles_z;
    {
        // preconditions of the buffer-addressed halo staging (conv3d_mfma.h, stage_halo_buf)
        auto edge_only = [](int n, int T, int R) { return n <= T || n % T == 0 || n % T >= R; };   // only the first / last tile sees padding
        if ((K2D == 0 && !edge_only(DX, C::TX, C::RX)) || !edge_only(D, C::TY, C::R) || !edge_only(D, C::TZ, C::R))
            return fail(SN_ERR_ARG, "%s: volume extent %d is not supported by this kernel's halo addressing (extent mod 8 must be 0 or >= %d)", L.name.c_str(), D, C::R);
        const long long plane = (long long)DX * D * D * 16, slack = 4LL * C::HVOX * 16;
        if (K2D == 0 && C::CS8MAX * plane + slack >= (1LL << 25))
            return fail(SN_ERR_ARG, "%s: a %d-group channel slab of a %dx%dx%d volume exceeds the 32 MiB this kernel's halo addressing covers", L.name.c_str(), C::CS8MAX, DX, D, D);
        if (K2D != 0 && SPLIT != 0 && C::CS8MAX * plane + slack >= 0x0FFFFFF0LL - (1 << 16))
            return fail(SN_ERR_ARG, "%s: %d images of %dx%d exceed the 256 MiB this kernel's halo addressing covers per channel slab", L.name.c_str(), DX, D, D);
    }
    a.act = L.act;
    if (EPI == EPI_STORE && KS != 1 && L.act != 0)       // conv3d_mfma.h: only the 1x1x1 store kernels carry the sigmoid epilogue
        return fail(SN_ERR_STATE, "%s: a sigmoid activation is only built for the 1x1x1 store kernels", L.name.c_str());
    {
        // static premultipliers of the 6-bit code planes (mx_format.h): the concat buffer holds sigmoid outputs, everything else ReLU(BN(.))
        auto e8_of = [&](const _Float16 *t) {
            if (t && c->c4_m8 && c->split == 1 && (t == c->a3 || t == c->a4 || t == c->b4)) return c->mx_c4_e8;
            return t && t == c->cat ? c->mx_cat_e8 : (t && t == c->x0 ? kMxX0E8 : c->mx_act_e8);
        };
        a.mx_in_e8 = e8_of(in.p); a.mx_out_e8 = e8_of(out.p); a.mx_side_e8 = sf ? e8_of(sf->side_out.p) : c->mx_act_e8;
    }
    if (EPI == EPI_SIDEPOOL) {
        if (!sf || !sf->side || !sf->side->side_frag || sf->side->cin != L.cout || sf->side->cout != 16 || L.nsplit != 1 || L.act != 0 || (D & 1))
            return fail(SN_ERR_STATE, "%s: fused side-conv / pool epilogue misconfigured", L.name.c_str());
        a.side_w = sf->side->side_frag; a.side_scale = sf->side->scale; a.side_shift = sf->side->shift; a.side_act = sf->side->act;
        a.side_out = sf->side_out.p; a.side_lo_off = sf->side_out.lo; a.side_cs = sf->side_cs; a.side_coff = sf->side_coff;
        a.pool_out = sf->pool_out.p; a.pool_lo_off = sf->pool_out.lo; a.pool_cs = sf->pool_cs;
    }
    if (c->d_num) {
        // one status bit per layer name (2-D similarityNet layers share bit 31)
        auto it = std::find(c->num_names.begin(), c->num_names.end(), L.name);
        size_t bit = it - c->num_names.begin();
        if (it == c->num_names.end() && c->num_names.size() < 31) c->num_names.push_back(L.name);
        a.status = c->d_num;
        // saturation warning: only for tensors stored with a 6-bit code plane whose values are not bounded by construction (ReLU outputs)
        constexpr int OS = OSPLIT < 0 ? SPLIT : OSPLIT;
        if (EPI == EPI_STORE && OS >= 2 && L.act == 0) {
            // (fp8 planes: the hi code holds values up to 448 * 2^-s, but the lo code - lo * 2^12 * 2^s with |lo| <= half an ulp of the fp16 hi - can already
            // saturate from |x| = 256 * 2^-s on (ulp 0.25: lo up to 0.125 -> 512 > 448); the warning starts where the first code can saturate. ADVICE r5)
            const _Float16 lim = (_Float16)std::min(60000.f, std::ldexp(OS >= 3 ? 255.875f : (SN_MX_FMT == 2 ? 7.5f : 28.f), a.mx_out_e8 - 127));
            unsigned short bits; memcpy(&bits, &lim, 2);
            a.mx_sat_bits = bits;
        }
        a.status_bit = bit < 31 ? (1u << bit) : (1u << 31);
    }
    a.nslab = (int)L.slab_c8.size();
    if (a.nslab < 1) return fail(SN_ERR_STATE, "%s: no channel slabs", L.name.c_str());      // (the kernel assumes at least one)
    a.bridge = L.bridge;
    if (L.bridge && !sn::sn_conv_has_bridge<KS, SPLIT, NW, PCH, NF, K2D, MF>())
        return fail(SN_ERR_STATE, "%s: packed with bridge chunks, launched on a kernel without them", L.name.c_str());
    if (C::PWM) {      // one-wave-per-SIMD loop: a slab's first piece issues DMAs that its second piece waits for
        for (unsigned char c8n : L.slab_c8)
            if (C::NTAP * c8n < 9) return fail(SN_ERR_STATE, "%s: a channel slab of fewer than two weight pieces", L.name.c_str());
    }
    for (int i = 0; i + 1 < a.nslab; ++i)
        if (L.slab_c8[i] != C::CS8MAX) return fail(SN_ERR_STATE, "%s: channel slab %d holds %d groups, the kernel expects %d in all but the last", L.name.c_str(), i, (int)L.slab_c8[i], C::CS8MAX);
    a.c8_last = L.slab_c8.back();
    const double vox = (double)B * DX * D * D;
    const double bytes = vox * 2.0 * (SPLIT ? 2 : 1) * (L.cin + (EPI == EPI_FINAL ? 2 : (EPI == EPI_SIDEPOOL ? 16 + L.cout / 8.0 : L.cout)));
    ProfScope ps(c, L.name, 2.0 * L.macs_per_voxel * vox, bytes);
    // persistent workgroups: one resident set, each walking tiles blockIdx.x, +gridDim.x, ...
    const int resident = std::max(1, c->num_cus * C::WG_PER_CU / L.nsplit);
    dim3 grid((unsigned)std::min(a.total_tiles, resident), (unsigned)L.nsplit);
    hipLaunchKernelGGL((conv3d_f16_mfma<KS, DIL, MF, NF, EPI, SPLIT, CS8, PCH, NW, PADV, K2D, OSPLIT>), grid, dim3(NW * 64), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    return SN_OK;
}

// Temporary device buffers of the host-array entry points (freed on scope exit).
struct TmpDev {
    std::vector<void *> p;
    ~TmpDev() { for (void *q : p) (void)hipFree(q); }
    template <typename T> T *get(size_t count) { void *q = nullptr; if (hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return nullptr; p.push_back(q); return static_cast<T *>(q); }
};

