// cvc_warp.h — colored-voxel-cube construction on the GPU (HBM-bound gather/scatter kernel).
//
// Replaces CVC.gen_coloredCubes / __colorize_cube__ (utils/CVC.py:6-104) and the mean subtraction of
// CVC.preprocess_augmentation (utils/CVC.py:108-111). Arithmetic is the reference's, bit for bit:
//   X = double(i)*double(resol_f32) + double(min_f32)        two roundings   (CVC.py:16-18)
//   q = P(3x4 f64) . [X Y Z 1]: FMA chain over k=0..3          np.dot -> dgemm (CVC.py:37)
//   u = rint(q0/q2), v = rint(q1/q2)   IEEE divide, half-even  (CVC.py:38-39)
//   in scope iff 0 <= u < W and 0 <= v < H, else colour 0       (CVC.py:45-46)
// One thread = one voxel of one cube-view-pair (both views of the pair); lanes run along z, the
// fastest output axis, so the six planar fp32 stores (NCDHW, the reference's layout) and the 16-byte
// channels-last fp16 store (the layout conv1_1 consumes) are fully coalesced; image texels of
// z-neighbouring voxels fall in the same or adjacent pixels, so the u8 gathers are L2/L1 hits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elementwise.h"

namespace sn {

struct CvcArgs {
    const int64_t *pairs;      // (n*n_vp, 2)
    const float *xyz;          // (n, 3)
    const float *resol;        // (n,)
    const double *cams;        // (V, 3, 4)
    const uint8_t *img_base;   // all images, back to back
    const long long *img_off;  // (V,) byte offset of image v
    const int *img_h, *img_w;  // (V,)
    float *out_ncdhw;          // (n*n_vp, 6, s,s,s) or nullptr
    _Float16 *out_x0;          // (n*n_vp, s,s,s, 8) fp16 mean-subtracted, or nullptr
    long long x0_lo_off;       // second plane of x0 (element offset from out_x0), used when x0_mode != 0
    int x0_mode;               // 0: fp16 only; 1: fp16 lo plane (f16x3); 2: [fp8(hi) | fp8(lo*2^12)] slot (f16m8)
    float mean[6];
    int sub_mean_ncdhw;        // 1: planar output is value - mean (preprocess), 0: raw 0..255
    int n_vp, s, V;
};

__device__ __forceinline__ double sn_dot4(const double *p, double X, double Y, double Z)
{
    double t = __dmul_rn(p[0], X);
    t = __fma_rn(p[1], Y, t);
    t = __fma_rn(p[2], Z, t);
    t = __fma_rn(p[3], 1.0, t);
    return t;
}

static __global__ void __launch_bounds__(256) cvc_warp_kernel(CvcArgs a)
{
    const int s = a.s;
    const int s3 = s * s * s;
    const int vox = blockIdx.x * 256 + threadIdx.x;
    if (vox >= s3) return;
    const int sample = blockIdx.y;
    const int cube = sample / a.n_vp;
    const int k = vox % s, j = (vox / s) % s, i = vox / (s * s);
    const double r = (double)a.resol[cube];
    const double X = __dadd_rn(__dmul_rn((double)i, r), (double)a.xyz[3 * cube + 0]);
    const double Y = __dadd_rn(__dmul_rn((double)j, r), (double)a.xyz[3 * cube + 1]);
    const double Z = __dadd_rn(__dmul_rn((double)k, r), (double)a.xyz[3 * cube + 2]);

    float rgb[6];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        long long view = a.pairs[2 * (size_t)sample + side];
        if (view < 0) view += a.V;
        rgb[3 * side + 0] = rgb[3 * side + 1] = rgb[3 * side + 2] = 0.f;
        if (view >= 0 && view < a.V) {
            const double *P = a.cams + 12 * view;
            const double q0 = sn_dot4(P, X, Y, Z), q1 = sn_dot4(P + 4, X, Y, Z), q2 = sn_dot4(P + 8, X, Y, Z);
            const double u = rint(__ddiv_rn(q0, q2)), w = rint(__ddiv_rn(q1, q2));
            const int W = a.img_w[view], H = a.img_h[view];
            if (u >= 0.0 && u < (double)W && w >= 0.0 && w < (double)H) {
                const uint8_t *px = a.img_base + a.img_off[view] + ((size_t)(int)w * W + (int)u) * 3;
                rgb[3 * side + 0] = (float)px[0];
                rgb[3 * side + 1] = (float)px[1];
                rgb[3 * side + 2] = (float)px[2];
            }
        }
    }
    if (a.out_ncdhw) {
        float *o = a.out_ncdhw + (size_t)sample * 6 * s3 + vox;
#pragma unroll
        for (int c = 0; c < 6; ++c) o[(size_t)c * s3] = a.sub_mean_ncdhw ? rgb[c] - a.mean[c] : rgb[c];
    }
    if (a.out_x0) {
        float y[8];
#pragma unroll
        for (int c = 0; c < 6; ++c) y[c] = rgb[c] - a.mean[c];
        y[6] = y[7] = 0.f;
        _Float16 *o = a.out_x0 + ((size_t)sample * s3 + vox) * 8;
        if (a.x0_mode == 1) sn_store8<1>(o, a.x0_lo_off, y);
        else if (a.x0_mode == 2) sn_store8<2>(o, a.x0_lo_off, y, kMxX0E8);
        else sn_store8<0>(o, 0, y);
    }
}

// NCDHW fp32 (the reference's network input, already mean-subtracted) -> channels-last fp16 x0.
static __global__ void __launch_bounds__(256) ncdhw_to_x0_kernel(const float *X, _Float16 *x0, int s3, int nsamples, long long x0_lo_off, int x0_mode)
{
    const int vox = blockIdx.x * 256 + threadIdx.x;
    const int sample = blockIdx.y;
    if (vox >= s3 || sample >= nsamples) return;
    const float *src = X + (size_t)sample * 6 * s3 + vox;
    float y[8];
#pragma unroll
    for (int c = 0; c < 6; ++c) y[c] = src[(size_t)c * s3];
    y[6] = y[7] = 0.f;
    _Float16 *o = x0 + ((size_t)sample * s3 + vox) * 8;
    if (x0_mode == 1) sn_store8<1>(o, x0_lo_off, y);
    else if (x0_mode == 2) sn_store8<2>(o, x0_lo_off, y, kMxX0E8);
    else sn_store8<0>(o, 0, y);
}

}  // namespace sn
