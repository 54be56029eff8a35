// elementwise.h — the HBM-bound glue ops of the SurfaceNet graph on channels-last fp16 tensors.
//   maxpool2_kernel      Pool3DDNNLayer((2,2,2), stride=2)                 nets/SurfaceNet.py:37,46
//   upsample3_cat_kernel Bilinear_3DInterpolation x3 (zero-insert + fixed k^3 conv, closed form of SURVEY App. D)
//                        written straight into their channel groups of the ConcatLayer buffer
//                                                                          nets/layers.py:363-390, SurfaceNet.py:71
//   fuse_kernel          ChannelPool_weightedAverage over the view pairs   nets/layers.py:325-336
//   relw_*               the relative-weight MLP + grouped softmax         nets/SurfaceNet.py:84-100
#pragma once
// (Non-template kernels here are `static`: the header is included by several translation units of the library.)
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sn {

#include "mx_format.h"
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// One 8-channel group of an activation tensor <-> 8 fp32 values, for the three storage modes of conv3d_mfma.h:
//   SPLIT 0: fp16;  SPLIT 1: hi + lo fp16 planes;  SPLIT 2: fp16 hi plane + 16-byte slot of low-precision codes (mx_format.h).
template <int SPLIT>
__device__ __forceinline__ void sn_load8(const _Float16 *p, long long lo_off, float (&v)[8], int e8 = kMxActE8)
{
    const h8 q = *reinterpret_cast<const h8 *>(p);
    if constexpr (SPLIT == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)q[e];
    } else if constexpr (SPLIT == 1) {
        const h8 ql = *reinterpret_cast<const h8 *>(p + lo_off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)q[e] + (float)ql[e];
    } else {
        const uint4 s = *reinterpret_cast<const uint4 *>(p + lo_off);
        // codes [hi c0..3 | lo c0..3 | hi c4..7 | lo c4..7] of (value * 2^(127 - e8)); lo additionally * 2^kMxLoExp
        const mx_v32f d = sn_mx6_decode(mx_v6i{(int)s.x, (int)s.y, (int)s.z, 0, 0, 0});
        const float sc = sn_e8_to_float(e8) / kMxLoMul;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = (float)q[e] + d[4 + e] * sc; v[4 + e] = (float)q[4 + e] + d[12 + e] * sc; }
    }
}
template <int SPLIT>
__device__ __forceinline__ void sn_store8(_Float16 *p, long long lo_off, const float (&v)[8], int e8 = kMxActE8)
{
    h8 h;
    float lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = (_Float16)v[e];
        lo[e] = v[e] - (float)h[e];
    }
    *reinterpret_cast<h8 *>(p) = h;
    if constexpr (SPLIT == 1) {
        h8 l;
#pragma unroll
        for (int e = 0; e < 8; ++e) l[e] = (_Float16)lo[e];
        *reinterpret_cast<h8 *>(p + lo_off) = l;
    } else if constexpr (SPLIT == 2) {
        uint4 s;
        mx_v32h t = {};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t[e] = h[e]; t[4 + e] = (_Float16)(lo[e] * kMxLoMul);
            t[8 + e] = h[4 + e]; t[12 + e] = (_Float16)(lo[4 + e] * kMxLoMul);
        }
        const mx_v6i c = sn_mx6_cvt(t, e8);
        s.x = (unsigned)c[0]; s.y = (unsigned)c[1]; s.z = (unsigned)c[2]; s.w = 0u;
        *reinterpret_cast<uint4 *>(p + lo_off) = s;
    }
}

// in [B][C/8][D][D][D][8] -> out [B][C/8][D/2][D/2][D/2][8]; one thread = one output voxel of one 8-channel group.
// SPLIT: values are hi+lo pairs of fp16 planes (lo plane at +lo_off elements); the max is taken on hi+lo.
template <int SPLIT>
__global__ void __launch_bounds__(256) maxpool2_kernel(const _Float16 *in, _Float16 *out, int D, int C, long long total,
                                                       long long in_lo_off, long long out_lo_off)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c8n = C >> 3, Do = D >> 1;
    long long t = idx;
    const int z = (int)(t % Do); t /= Do;
    const int y = (int)(t % Do); t /= Do;
    const int x = (int)(t % Do); t /= Do;
    const int c8 = (int)(t % c8n);
    const long long b = t / c8n;
    // group-blocked layout [b][c8][x][y][z][8]
    const _Float16 *p = in + (((b * c8n + c8) * D + 2 * x) * D + 2 * y) * (long long)D * 8 + 2 * z * 8;
    float m[8], q[8];
    sn_load8<SPLIT>(p, in_lo_off, m);
#pragma unroll
    for (int o = 1; o < 8; ++o) {
        sn_load8<SPLIT>(p + ((long long)((o >> 2) * D + ((o >> 1) & 1)) * D + (o & 1)) * 8, in_lo_off, q);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = q[e] > m[e] ? q[e] : m[e];
    }
    sn_store8<SPLIT>(out + idx * 8, out_lo_off, m);
}

// Per-axis operator of the "bilinear" upsampler: output index o = F*m + ph reads
// in[m]*wa + in[m+1]*wb with in[m+1] = 0 past the end.
template <int F>
__device__ __forceinline__ void up_axis(int o, int n_in, int &m, float &wa, float &wb)
{
    m = o / F;
    const int ph = o - m * F;
    if (F == 2) {
        wa = ph == 0 ? 1.f : 0.5f;
        wb = ph == 0 ? 0.f : 0.5f;
    } else {
        const float third = 1.0f / 3.0f, two3 = 2.0f / 3.0f;
        wa = ph == 0 ? 1.f : (ph == 1 ? two3 : (ph == 2 ? third : 0.f));
        wb = ph == 0 ? 0.f : (ph == 1 ? 0.f : (ph == 2 ? third : two3));
    }
    if (m + 1 >= n_in) wb = 0.f;
}

// All three upsampled side outputs of one voxel in one pass: cat[...][16..63] = [up2(s2) | up4(s3) | up4(s4)], i.e. the 96
// contiguous bytes (per plane) that follow side_op1's 16 channels in the ConcatLayer buffer (nets/SurfaceNet.py:71).
// One thread = one output voxel of one of the 6 destination 8-channel groups (consecutive threads = consecutive z:
// fully coalesced 16-byte stores in the group-blocked layout [b][group][x][y][z][8]).
template <int SPLIT, int OSPLIT = SPLIT>
__global__ void __launch_bounds__(256) upsample3_cat_kernel(const _Float16 *s2, const _Float16 *s3, const _Float16 *s4, _Float16 *cat,
                                                            int Do, int cat_cs, long long total, long long lo2, long long lo3,
                                                            long long lo4, long long out_lo_off, int out_e8)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    long long t = idx;
    const int z = (int)(t % Do); t /= Do;
    const int y = (int)(t % Do); t /= Do;
    const int x = (int)(t % Do); t /= Do;
    const int g = (int)(t % 6);
    const long long b = t / 6;
    const int src = g >> 1, c8 = g & 1;                 // 0: s2 (x2), 1: s3 (x4), 2: s4 (x4)
    const _Float16 *in = src == 0 ? s2 : (src == 1 ? s3 : s4);
    const long long in_lo = src == 0 ? lo2 : (src == 1 ? lo3 : lo4);
    const int Di = src == 0 ? Do / 2 : Do / 4;
    int mx, my, mz;
    float ax, bx, ay, by, az, bz;
    if (src == 0) { up_axis<2>(x, Di, mx, ax, bx); up_axis<2>(y, Di, my, ay, by); up_axis<2>(z, Di, mz, az, bz); }
    else          { up_axis<4>(x, Di, mx, ax, bx); up_axis<4>(y, Di, my, ay, by); up_axis<4>(z, Di, mz, az, bz); }
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        const int dx = o >> 2, dy = (o >> 1) & 1, dz = o & 1;
        const float w = (dx ? bx : ax) * (dy ? by : ay) * (dz ? bz : az);
        if (w != 0.f) {
            const _Float16 *pq = in + (((((b * 2 + c8) * Di + mx + dx) * Di + my + dy) * Di + mz + dz) * 8LL);
            float q[8];
            sn_load8<SPLIT>(pq, in_lo, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += w * q[e];
        }
    }
    _Float16 *o = cat + ((((b * (cat_cs >> 3) + 2 + g) * Do + x) * Do + y) * Do + z) * 8LL;
    sn_store8<OSPLIT>(o, out_lo_off, acc, out_e8);   // OSPLIT: storage format the consumer (merge_conv_a) computes in
}

// The same, LDS-tiled (round 3): one workgroup = an 8 x 8 x Do block of output voxels of ONE destination group. The source voxels the
// block's stencils touch ((8/F+1)^2 x (Do/F+1): 425 for the x2 map, 81 for the x4 maps) are fetched ONCE into LDS as fp32 groups and every
// output voxel sums its <= 8 corners from there - in the per-voxel kernel above every thread fetches its own corners from L2 (~10 16-byte
// loads per thread, 13 TB/s of L2 traffic for 0.8 GB of output). Same corner order and fp32 arithmetic: bit-identical results.
// Requires Do % 8 == 0 (launch_up3 falls back to the per-voxel kernel otherwise).
template <int SPLIT, int OSPLIT = SPLIT>
__global__ void __launch_bounds__(256) upsample3_cat_tiled_kernel(const _Float16 *s2, const _Float16 *s3, const _Float16 *s4, _Float16 *cat,
                                                                  int Do, int cat_cs, long long lo2, long long lo3, long long lo4,
                                                                  long long out_lo_off, int out_e8)
{
    constexpr int TB = 8, SMAX = (TB / 2 + 1) * (TB / 2 + 1);            // source (x, y) footprint of a tile, x2 case: 5 x 5
    __shared__ float src[SMAX * 33 * 8];                                 // [sx][sy][sz <= Do/2 + 1 <= 33][8]  (Do <= 64)
    const int tiles = Do / TB;
    int t = blockIdx.x;
    const int ty = t % tiles; t /= tiles;
    const int tx = t % tiles; t /= tiles;
    const int g = t % 6;
    const long long b = t / 6;
    const int srcsel = g >> 1, c8 = g & 1;                               // 0: s2 (x2), 1: s3 (x4), 2: s4 (x4)
    const int F = srcsel == 0 ? 2 : 4;
    const _Float16 *in = srcsel == 0 ? s2 : (srcsel == 1 ? s3 : s4);
    const long long in_lo = srcsel == 0 ? lo2 : (srcsel == 1 ? lo3 : lo4);
    const int Di = Do / F, x0 = tx * TB, y0 = ty * TB, mx0 = x0 / F, my0 = y0 / F;
    const int nsx = TB / F + 1, nsz = Di + 1;                            // source extent per axis (one past the end: weight 0 there, slot zero-filled)
    for (int i = threadIdx.x; i < nsx * nsx * nsz; i += 256) {
        const int sz = i % nsz, sy = (i / nsz) % nsx, sx = i / (nsz * nsx);
        float q[8];
        const int gx = mx0 + sx, gy = my0 + sy;
        if (gx < Di && gy < Di && sz < Di) sn_load8<SPLIT>(in + (((((b * 2 + c8) * Di + gx) * Di + gy) * Di + sz) * 8LL), in_lo, q);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = 0.f;
        }
        float *d = src + ((sx * nsx + sy) * 33 + sz) * 8;
        *reinterpret_cast<float4 *>(d) = float4{q[0], q[1], q[2], q[3]};
        *reinterpret_cast<float4 *>(d + 4) = float4{q[4], q[5], q[6], q[7]};
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TB * TB * Do; idx += 256) {        // consecutive threads = consecutive z: coalesced 16-byte stores
        const int z = idx % Do, y = y0 + (idx / Do) % TB, x = x0 + idx / (Do * TB);
        int mx, my, mz;
        float ax, bx, ay, by, az, bz;
        if (srcsel == 0) { up_axis<2>(x, Di, mx, ax, bx); up_axis<2>(y, Di, my, ay, by); up_axis<2>(z, Di, mz, az, bz); }
        else             { up_axis<4>(x, Di, mx, ax, bx); up_axis<4>(y, Di, my, ay, by); up_axis<4>(z, Di, mz, az, bz); }
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int dx = o >> 2, dy = (o >> 1) & 1, dz = o & 1;
            const float w = (dx ? bx : ax) * (dy ? by : ay) * (dz ? bz : az);
            if (w != 0.f) {
                const float *q = src + (((mx - mx0 + dx) * nsx + (my - my0 + dy)) * 33 + mz + dz) * 8;
                const float4 q0 = *reinterpret_cast<const float4 *>(q), q1 = *reinterpret_cast<const float4 *>(q + 4);
                acc[0] += w * q0.x; acc[1] += w * q0.y; acc[2] += w * q0.z; acc[3] += w * q0.w;
                acc[4] += w * q1.x; acc[5] += w * q1.y; acc[6] += w * q1.z; acc[7] += w * q1.w;
            }
        }
        _Float16 *o = cat + ((((b * (cat_cs >> 3) + 2 + g) * Do + x) * Do + y) * Do + z) * 8LL;
        sn_store8<OSPLIT>(o, out_lo_off, acc, out_e8);
    }
}

// Calibration scan (sn_calibrate_dev): how many of a tensor's stored fp16 values exceed the range of 6-bit codes under each candidate premultiplier.
// hist[j], j = 0 .. kMxScanBins - 1: count of |v| > lim * 2^-(j - kMxScanBins / 2), lim = the code format's largest magnitude; hist[kMxScanBins] = count of
// non-zero values, hist[kMxScanBins + 1] = fp16 bits of the largest magnitude. `halfs` is a multiple of 8 (8-channel groups).
constexpr int kMxScanBins = 13;      // premultiplier exponents s = -6 .. +6
static __global__ void __launch_bounds__(256) mx_scan_kernel(const _Float16 *t, long long halfs, float lim, unsigned long long *hist)
{
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    unsigned cnt[kMxScanBins + 1], thr[kMxScanBins];
#pragma unroll
    for (int j = 0; j < kMxScanBins; ++j) {
        const _Float16 h = (_Float16)ldexpf(lim, -(j - kMxScanBins / 2));
        thr[j] = __builtin_bit_cast(unsigned short, h);       // (non-negative fp16 values order like their bit patterns)
        cnt[j] = 0;
    }
    cnt[kMxScanBins] = 0;
    unsigned mx = 0;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; i < halfs; i += (long long)gridDim.x * 256 * 8) {
        const u16x8 v = *reinterpret_cast<const u16x8 *>(t + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned b = v[e] & 0x7fffu;
            mx = b > mx ? b : mx;
            cnt[kMxScanBins] += b != 0u;
#pragma unroll
            for (int j = 0; j < kMxScanBins; ++j) cnt[j] += b > thr[j];
        }
    }
#pragma unroll
    for (int j = 0; j <= kMxScanBins; ++j) {
        unsigned c = cnt[j];
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(hist + j, (unsigned long long)c);
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned m2 = __shfl_xor(mx, o); mx = m2 > mx ? m2 : mx; }
    if ((threadIdx.x & 63) == 0) atomicMax(hist + kMxScanBins + 1, (unsigned long long)mx);
}

// unfused [n][n_vp][s3] f32, w [n][n_vp] -> fused [n][s3]; w == nullptr (n_vp == 1) copies.
static __global__ void __launch_bounds__(256) fuse_kernel(const float *unfused, const float *w, float *fused, int n_vp, int s3, long long total)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long cube = idx / s3;
    const int vox = (int)(idx - cube * s3);
    const float *u = unfused + cube * n_vp * (long long)s3 + vox;
    if (w == nullptr) { fused[idx] = u[0]; return; }
    float wsum = 0.f;
    for (int p = 0; p < n_vp; ++p) wsum += w[cube * n_vp + p];
    float acc = 0.f;
    for (int p = 0; p < n_vp; ++p) acc += u[(long long)p * s3] * (w[cube * n_vp + p] / wsum);
    fused[idx] = acc;
}

// Voxel-level colour fusion, generate_voxelLevelWeighted_coloredCubes (utils/utils.py:8-42; main_reconstruct.py:150-152),
// float32 op for op: vw = w*pred; vw /= sum_p vw; mc = ((x_a+mean) + (x_b+mean))/2; rgb = uint8(sum_p vw*mc).
// cvc (n*n_vp, 6, s3) is the MEAN-SUBTRACTED tensor the hot path produced (the caller's `X += mean` happens here).
static __global__ void __launch_bounds__(256) color_fuse_kernel(const float *cvc, const float *unfused, const float *w, unsigned char *rgb,
                                                         int n_vp, int s3, long long total, float m0, float m1, float m2, float m3,
                                                         float m4, float m5)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long long cube = idx / s3;
    const int vox = (int)(idx - cube * s3);
    const float mean[6] = {m0, m1, m2, m3, m4, m5};
    float tot = 0.f;
    for (int p = 0; p < n_vp; ++p) tot = tot + __fmul_rn(w[cube * n_vp + p], unfused[(cube * n_vp + p) * s3 + vox]);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int p = 0; p < n_vp; ++p) {
        const float vw = __fdiv_rn(__fmul_rn(w[cube * n_vp + p], unfused[(cube * n_vp + p) * s3 + vox]), tot);
        const float *x = cvc + ((cube * n_vp + p) * 6) * (long long)s3 + vox;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = __fadd_rn(x[(long long)c * s3], mean[c]), b = __fadd_rn(x[(long long)(c + 3) * s3], mean[c + 3]);
            const float mc = __fdiv_rn(__fadd_rn(a, b), 2.0f);
            acc[c] = __fadd_rn(acc[c], __fmul_rn(vw, mc));
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = acc[c];
        rgb[(cube * 3 + c) * s3 + vox] = (v == v) ? (unsigned char)(int)v : (unsigned char)0;   // NaN (all weights 0) -> 0
    }
}

// Relative-weight MLP: one block (128 threads) per feature row. W1 (258,100) row-major fp32.
static __global__ void __launch_bounds__(128) relw_mlp_kernel(const float *feat, const float *W1, const float *scale1, const float *shift1,
                                                       const float *w2, float b2, float *z, int d_in, int n_hidden)
{
    __shared__ float red[128];
    const int row = blockIdx.x, j = threadIdx.x;
    float hv = 0.f;
    if (j < n_hidden) {
        const float *f = feat + (size_t)row * d_in;
        float acc = 0.f;
        for (int k = 0; k < d_in; ++k) acc += f[k] * W1[(size_t)k * n_hidden + j];
        hv = 1.0f / (1.0f + expf(-(acc * scale1[j] + shift1[j]))) * w2[j];
    }
    red[j] = hv;
    __syncthreads();
    for (int st = 64; st > 0; st >>= 1) {
        if (j < st) red[j] += red[j + st];
        __syncthreads();
    }
    if (j == 0) z[row] = red[0] + b2;
}

// Same MLP with the feature row assembled on the fly (utils/viewPairSelection.py:69-74): row (cube, pair p = (i,j)) =
// [emb[cube,i] (128) | emb[cube,j] (128) | dissimilarity[cube,p] | angle[cube,p]]; same accumulation order as relw_mlp_kernel,
// so the two agree bit for bit. emb (n_cubes, n_views, 128); pairs (P,2); d, theta (n_cubes, P).
static __global__ void __launch_bounds__(128) relw_mlp_pairs_kernel(const float *emb, const int *pairs, const float *d, const float *theta, int n_views,
                                                             int P, const float *W1, const float *scale1, const float *shift1, const float *w2,
                                                             float b2, float *z, int n_hidden)
{
    __shared__ float red[128];
    __shared__ float f[258];
    const long long row = blockIdx.x;
    const int j = threadIdx.x;
    const long long cube = row / P;
    const int p = (int)(row - cube * P);
    const float *e1 = emb + ((size_t)cube * n_views + pairs[2 * p]) * 128, *e2 = emb + ((size_t)cube * n_views + pairs[2 * p + 1]) * 128;
    f[j] = e1[j];
    f[128 + j] = e2[j];
    if (j == 0) { f[256] = d[row]; f[257] = theta[row]; }
    __syncthreads();
    float hv = 0.f;
    if (j < n_hidden) {
        float acc = 0.f;
        for (int k = 0; k < 258; ++k) acc += f[k] * W1[(size_t)k * n_hidden + j];
        hv = 1.0f / (1.0f + expf(-(acc * scale1[j] + shift1[j]))) * w2[j];
    }
    red[j] = hv;
    __syncthreads();
    for (int st = 64; st > 0; st >>= 1) {
        if (j < st) red[j] += red[j + st];
        __syncthreads();
    }
    if (j == 0) z[row] = red[0] + b2;
}

static __global__ void relw_softmax_kernel(const float *z, float *out, int n, int n_vp)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    float mx = -3.0e38f;
    for (int p = 0; p < n_vp; ++p) mx = fmaxf(mx, z[(size_t)g * n_vp + p]);
    float sum = 0.f;
    for (int p = 0; p < n_vp; ++p) sum += expf(z[(size_t)g * n_vp + p] - mx);
    for (int p = 0; p < n_vp; ++p) out[(size_t)g * n_vp + p] = expf(z[(size_t)g * n_vp + p] - mx) / sum;
}

}  // namespace sn
