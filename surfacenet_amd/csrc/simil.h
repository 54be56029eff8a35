// simil.h — glue kernels of the similarityNet / early-rejection stage. The 13 convolutions run on
// conv3d_f16_mfma<..., K2D = 1> (conv3d_mfma.h): a batch of N patches is the volume (x = patch index, y, z = pixel
// row, column), activations in the same 8-channel-group layout act[c/8][n][h][w][c%8].
//   patch_crop_kernel      image.cropImgPatches (utils/image.py:92-183; pyramidRate = 1 as earlyRejection.py:50 calls it:
//                          a 64x64 window of nearest-clamped pixels around the truncated cube-centre projection)
//                          + image.preprocess_patches (utils/image.py:9-36): RGB -> BGR, - mean_BGR, (h,w,c) -> (c,h,w)
//   nchw_to_p0_kernel      the same network input from host-preprocessed (n,3,64,64) float32 patches
//   (Pool2DLayer(2), nets/similarityNet.py:31-47, is fused into the store epilogue of the conv in front of it:
//    EPI_POOL2D in conv3d_mfma.h)
//   simil_features_kernel  FlattenLayer(pool5) ++ CropFeatureMapCenterLayer(pool1..4, r=1) -> L2NormLayer
//                                                                              nets/similarityNet.py:49-57, nets/layers.py:15-81
//   simil_dense_kernel     DenseLayer(5888 -> 128, linear)                     nets/similarityNet.py:57
//   pair_simil_kernel      DistanceLayer(Lp=2) + DenseLayer(1, sigmoid)       nets/similarityNet.py:71-77, nets/layers.py:130-138
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elementwise.h"

namespace sn {

constexpr int kPatch = 64;             // params.__imgPatch_hw_size
constexpr int kSimilFeat = 5888;       // 512*4 (pool5) + (64 + 128 + 256 + 512) * 4 (centre crops)
constexpr int kEmb = 128;              // params.__D_imgPatchEmbedding

// One thread = one pixel of one patch. centers: (2, n) float64 = (h, w) projections of the cube centres.
// patches_u8 (n,64,64,3) RGB and/or p0 (network input, 8-channel group 0: B-mean, G-mean, R-mean, 0...) are written.
template <int SPLIT>
__global__ void __launch_bounds__(256) patch_crop_kernel(const uint8_t *img, int H, int W, const double *center_h, const double *center_w,
                                                         int n, uint8_t *patches_u8, _Float16 *p0, long long lo_off, float mb, float mg, float mr)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)n * kPatch * kPatch) return;
    const int pw = (int)(idx % kPatch), ph = (int)((idx / kPatch) % kPatch), i = (int)(idx / (kPatch * kPatch));
    // (center * 1.0).astype(np.int) - patchSize/2, then clip to the image (image.py:160-169)
    const long long h0 = (long long)center_h[i] - kPatch / 2, w0 = (long long)center_w[i] - kPatch / 2;
    long long h = h0 + ph, w = w0 + pw;
    h = h < 0 ? 0 : (h > H - 1 ? H - 1 : h);
    w = w < 0 ? 0 : (w > W - 1 ? W - 1 : w);
    const uint8_t *px = img + ((size_t)h * W + (size_t)w) * 3;
    const uint8_t r = px[0], g = px[1], b = px[2];
    if (patches_u8) {
        uint8_t *o = patches_u8 + (size_t)idx * 3;
        o[0] = r; o[1] = g; o[2] = b;
    }
    if (p0) {
        const float v[8] = {(float)b - mb, (float)g - mg, (float)r - mr, 0.f, 0.f, 0.f, 0.f, 0.f};
        sn_store8<SPLIT>(p0 + (size_t)idx * 8, lo_off, v);
    }
}

template <int SPLIT>
__global__ void __launch_bounds__(256) nchw_to_p0_kernel(const float *X, int n, _Float16 *p0, long long lo_off)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int hw = kPatch * kPatch;
    if (idx >= (long long)n * hw) return;
    const int pix = (int)(idx % hw), i = (int)(idx / hw);
    const float *s = X + (size_t)i * 3 * hw + pix;
    const float v[8] = {s[0], s[hw], s[2 * hw], 0.f, 0.f, 0.f, 0.f, 0.f};
    sn_store8<SPLIT>(p0 + (size_t)idx * 8, lo_off, v);
}

struct SimilFeatArgs {
    const _Float16 *pool[5];     // pool1..pool5, [C/8][N][H][H][8]
    long long lo_off[5];
    float *feat;                 // (N, 5888) L2-normalised
    int n;
};

// One workgroup (256 threads) per patch: gather the 5888 features (fp32 = hi + lo), sum of squares, scale.
template <int SPLIT>
__global__ void __launch_bounds__(256) simil_features_kernel(SimilFeatArgs a)
{
    __shared__ float f[kSimilFeat];
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    // concat order (similarityNet.py:49-55): pool5 flatten, then centre crops of pool1, pool2, pool3, pool4; each (c, h, w)
    const int src[5] = {4, 0, 1, 2, 3};
    const int C[5] = {64, 128, 256, 512, 512}, Hs[5] = {32, 16, 8, 4, 2};
    int base = 0;
    float ss = 0.f;
    for (int k = 0; k < 5; ++k) {
        const int s = src[k], H = Hs[s], c8n = C[s] >> 3, h0 = H / 2 - 1;     // r = 1: rows/cols [H/2-1, H/2+1)
        // work item = (8-channel group, one of the 4 pixels)
        for (int t = tid; t < c8n * 4; t += 256) {
            const int c8 = t >> 2, hh = (t >> 1) & 1, ww = t & 1;
            float v[8];
            sn_load8<SPLIT>(a.pool[s] + ((((long long)c8 * a.n + i) * H + (h0 + hh)) * H + (h0 + ww)) * 8, a.lo_off[s], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f[base + ((c8 * 8 + e) << 2) + hh * 2 + ww] = v[e];
                ss += v[e] * v[e];
            }
        }
        base += C[s] * 4;
    }
    for (int o = 32; o; o >>= 1) ss += __shfl_xor(ss, o);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float inv = 1.0f / sqrtf(red[0] + red[1] + red[2] + red[3]);
    for (int t = tid; t < kSimilFeat; t += 256) a.feat[(size_t)i * kSimilFeat + t] = f[t] * inv;
}

// emb (N,128) = feat (N,5888) . W (5888,128) + b in fp32. One workgroup (256 threads) = 32 patches x 128 outputs x one of
// kDenseKS K-ranges (blockIdx.y), thread tile 4 patches x 4 outputs, K staged through LDS in steps of 32. Partial sums go to
// part[ks][n][128]; simil_dense_reduce_kernel adds them in fixed order (+ bias): deterministic, and independent of the batch
// size and of the patch's position in the batch.
constexpr int kDenseKS = 8, kDenseKR = kSimilFeat / kDenseKS;      // 736 = 23 steps of 32
__global__ void __launch_bounds__(256) simil_dense_kernel(const float *feat, const float *W, float *part, int n)
{
    constexpr int TM = 32, TK = 32;
    __shared__ float As[TK][TM + 1];
    __shared__ __attribute__((aligned(16))) float Ws[TK][kEmb];
    const int tid = threadIdx.x, i0 = blockIdx.x * TM;
    const int tj = (tid & 31) * 4, ti = (tid >> 5) * 4;          // outputs tj..tj+3, patches ti..ti+3
    float acc[4][4] = {};
    const int kbeg = blockIdx.y * kDenseKR;
    for (int k0 = kbeg; k0 < kbeg + kDenseKR; k0 += TK) {
        for (int t = tid; t < TM * TK; t += 256) {
            const int r = t / TK, k = t - r * TK;
            As[k][r] = (i0 + r < n) ? feat[(size_t)(i0 + r) * kSimilFeat + k0 + k] : 0.f;
        }
        for (int t = tid; t < TK * kEmb / 4; t += 256) {
            const int k = t / (kEmb / 4), c4 = t - k * (kEmb / 4);
            *reinterpret_cast<float4 *>(&Ws[k][c4 * 4]) = *reinterpret_cast<const float4 *>(W + (size_t)(k0 + k) * kEmb + c4 * 4);
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < TK; ++k) {
            const float4 w = *reinterpret_cast<const float4 *>(&Ws[k][tj]);
            const float a0 = As[k][ti], a1 = As[k][ti + 1], a2 = As[k][ti + 2], a3 = As[k][ti + 3];
            acc[0][0] = fmaf(a0, w.x, acc[0][0]); acc[0][1] = fmaf(a0, w.y, acc[0][1]); acc[0][2] = fmaf(a0, w.z, acc[0][2]); acc[0][3] = fmaf(a0, w.w, acc[0][3]);
            acc[1][0] = fmaf(a1, w.x, acc[1][0]); acc[1][1] = fmaf(a1, w.y, acc[1][1]); acc[1][2] = fmaf(a1, w.z, acc[1][2]); acc[1][3] = fmaf(a1, w.w, acc[1][3]);
            acc[2][0] = fmaf(a2, w.x, acc[2][0]); acc[2][1] = fmaf(a2, w.y, acc[2][1]); acc[2][2] = fmaf(a2, w.z, acc[2][2]); acc[2][3] = fmaf(a2, w.w, acc[2][3]);
            acc[3][0] = fmaf(a3, w.x, acc[3][0]); acc[3][1] = fmaf(a3, w.y, acc[3][1]); acc[3][2] = fmaf(a3, w.z, acc[3][2]); acc[3][3] = fmaf(a3, w.w, acc[3][3]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (i0 + ti + r < n)
            *reinterpret_cast<float4 *>(part + ((size_t)blockIdx.y * n + i0 + ti + r) * kEmb + tj) = float4{acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
}

__global__ void __launch_bounds__(256) simil_dense_reduce_kernel(const float *part, const float *b, float *emb, int n)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * kEmb) return;
    float acc = 0.f;
#pragma unroll
    for (int ks = 0; ks < kDenseKS; ++ks) acc += part[(size_t)ks * n * kEmb + idx];
    emb[idx] = acc + b[idx & (kEmb - 1)];
}

// emb_pairs (2*n, 128): rows 2i, 2i+1 form pair i -> simil (n,) = sigmoid(w * ||e1 - e2||_2 + b)
__global__ void __launch_bounds__(256) pair_simil_kernel(const float *emb_pairs, float *simil, int n, float w, float b)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const float *e1 = emb_pairs + (size_t)(2 * i) * kEmb, *e2 = e1 + kEmb;
    float ss = 0.f;
    for (int k = lane; k < kEmb; k += 64) { const float d = fabsf(e1[k] - e2[k]); ss += d * d; }
    for (int o = 32; o; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) simil[i] = 1.0f / (1.0f + expf(-(w * sqrtf(ss) + b)));
}

// All 2-combinations of views at once (earlyRejection.embeddingPairs2simil, utils/earlyRejection.py:59-90): emb (n_cubes, n_views,
// 128), pairs (P,2) -> simil (n_cubes, P). One wave per (cube, pair); same summation order as pair_simil_kernel, so the two
// entry points agree bit for bit.
__global__ void __launch_bounds__(256) pair_simil_all_kernel(const float *emb, const int *pairs, float *simil, long long total, int n_views, int P,
                                                             float w, float b)
{
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= total) return;
    const long long cube = i / P;
    const int p = (int)(i - cube * P);
    const float *e1 = emb + ((size_t)cube * n_views + pairs[2 * p]) * kEmb, *e2 = emb + ((size_t)cube * n_views + pairs[2 * p + 1]) * kEmb;
    float ss = 0.f;
    for (int k = lane; k < kEmb; k += 64) { const float d = fabsf(e1[k] - e2[k]); ss += d * d; }
    for (int o = 32; o; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) simil[i] = 1.0f / (1.0f + expf(-(w * sqrtf(ss) + b)));
}

}  // namespace sn
