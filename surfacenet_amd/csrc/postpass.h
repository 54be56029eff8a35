// postpass.h — the per-cube post-pass of the reconstruction loop on the GPU:
//   ray_pool_kernel        utils/rayPooling.py:143-260  rayPooling_1cube_numpy, all cubes x distinct views of a batch
//   d2s_count / d2s_scan / d2s_write   utils/sparseCubes.py:9-77  dense2sparse (threshold / vote filter, centre crop,
//                          ordered compaction into packed voxel lists)
// Integer / byte work, HBM- and atomics-bound: no MFMA. Results are the reference's, bit for bit.
//
// Ray pooling, restated (see oracle/post_oracle.py for the derivation from the reference's scatter + argmax):
//   voxel selected  iff fp16(pred) > fp16(thresh)                                        (rayPooling.py:214-215)
//   pixel (w,h) = rint(q0/q2), rint(q1/q2), depth bin d = rint(q2 / resol), q = P.[X Y Z 1] (camera.py:173-183, :228-229)
//   CELL  (w,h,d): keeps the selected voxel with the LARGEST flat index               (last write wins, :248)
//   PIXEL (w,h):   its cell with the largest stored prediction, ties -> smallest d     (argmax, :250)
//   that cell's voxel gets one vote per occurrence of the view in the cube's pair list  (:252-255)
//   all-zero pixel: argmax = column 0 = the view's minimum bin; its cell's voxel, or voxel 0 if that cell is empty.
// One workgroup = one (cube, pair-list entry); entries that repeat an earlier view of the same cube exit at once, the
// first occurrence votes with the view's multiplicity. Two open-addressing hash tables per workgroup, sized 2x the selected voxel count -
// in LDS when at most RP_LDS_M voxels are selected (round 4: the realistic case - ~1,700 of a 32^3 cube at the bench's threshold; every
// insert / max / lookup is then an LDS atomic instead of a round trip to L2), else in a global workspace (<= 2 MB/workgroup):
//   pixel table  key = (w,h) as 2 x int32            value = packed best (pred bits << 32 | ~relative depth)
//   cell table   key = (pixel slot, d)               value = max flat voxel index
// Using the pixel's slot in the cell key keeps both keys 64-bit for arbitrary int32 pixel coordinates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cvc_warp.h"

namespace sn {

constexpr int RP_NT = 1024;
constexpr int RP_LIST = 8192;        // selected voxels of a (cube, view) listed in LDS (2 x 32 KB)
constexpr int RP_LDS_M = 3400;       // ... up to this many (10 % of a 32^3 cube), the hash tables live in LDS as well: capacity RP_LDS_CAP, load factor <= 0.83
constexpr int RP_LDS_CAP = 4096;
constexpr int RP_LDS_Q = 14336, RP_LDS_TAB = 28672;
static_assert(RP_LDS_M * 4 <= RP_LDS_Q && RP_LDS_Q + RP_LDS_M * 4 <= RP_LDS_TAB && RP_LDS_M <= RP_LIST && RP_LDS_M * 6 <= RP_LDS_CAP * 5, "ray_pool_kernel LDS map");
// LDS map: [0, 32 K) list of selected voxels (RP_LIST entries; with LDS tables only the first RP_LDS_M are in use), then either the cell slots of all
// RP_LIST voxels [32 K, 64 K) or - LDS tables - the cell slots of RP_LDS_M voxels [14 K, 28 K) and the four tables [28 K, 28 K + 28 B x RP_LDS_CAP)
constexpr int RP_LDS_BYTES = RP_LDS_TAB + RP_LDS_CAP * 28 > 2 * RP_LIST * 4 ? RP_LDS_TAB + RP_LDS_CAP * 28 : 2 * RP_LIST * 4;
constexpr unsigned long long RP_EMPTY = ~0ull;
constexpr unsigned RP_NONE = 0xFFFFFFFFu;

struct RayPoolArgs {
    const int64_t *pairs;      // (n, n_vp, 2)
    const float *xyz;          // (n, 3)
    const float *resol;        // (n,)
    const double *cams;        // (V, 3, 4)
    const float *pred;         // (n, s,s,s) float32 probabilities (rounded to fp16 here, sparseCubes.py:136)
    uint8_t *votes;            // (n, s,s,s), zeroed by the caller
    unsigned long long *pix_key, *pix_best, *cell_key;   // [wg][cap_max]
    unsigned *cell_idx;        // [wg][cap_max]
    unsigned *cslot;           // [wg][s^3] cell slot of each selected voxel
    int *err;                  // device flag: 1 = pixel / depth outside the int32 range
    int n_vp, s, V, cap_max, use_thresh;
    float thresh;
};

__device__ __forceinline__ unsigned rp_hash(unsigned long long k, unsigned mask)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k & mask;
}

// insert-or-find; returns the slot. Table entries start as RP_EMPTY.
// The tables live in LDS or in global memory; the helpers are templates over the pointer type so that each instantiation addresses ONE address space
// (with generic pointers every access becomes a flat instruction behind a full s_waitcnt, and the lock-step passes below serialise again).
typedef __attribute__((address_space(3))) unsigned long long rp_lds_u64;
typedef __attribute__((address_space(3))) unsigned rp_lds_u32;
template <typename P>
__device__ __forceinline__ auto rp_ld(P p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename P>
__device__ __forceinline__ unsigned long long rp_cas(P p, unsigned long long expected, unsigned long long desired)
{
    __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expected;            // the value found (== the old `expected` on success)
}
template <typename P, typename T>
__device__ __forceinline__ void rp_max(P p, T v) { (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// insert-or-find; returns the slot. Table entries start as RP_EMPTY.
template <typename P>
__device__ __forceinline__ unsigned rp_insert(P keys, unsigned mask, unsigned long long k)
{
    unsigned h = rp_hash(k, mask);
    for (;;) {
        unsigned long long cur = rp_ld(keys + h);
        if (cur == k) return h;
        if (cur == RP_EMPTY) {
            unsigned long long prev = rp_cas(keys + h, RP_EMPTY, k);
            if (prev == RP_EMPTY || prev == k) return h;
        }
        h = (h + 1) & mask;
    }
}

template <typename P>
__device__ __forceinline__ bool rp_find(P keys, unsigned mask, unsigned long long k)
{
    unsigned h = rp_hash(k, mask);
    for (;;) {
        unsigned long long cur = rp_ld(keys + h);
        if (cur == k) return true;
        if (cur == RP_EMPTY) return false;
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ void rp_vote(uint8_t *votes, unsigned i, unsigned mult)
{
    atomicAdd(reinterpret_cast<unsigned *>(votes) + (i >> 2), mult << (8 * (i & 3)));
}

__global__ void __launch_bounds__(RP_NT) ray_pool_kernel(RayPoolArgs a)
{
    __shared__ int sh_i[4];          // m, dmin, vote-for-voxel-0 flag, unused
    const int tid = threadIdx.x;
    const int cube = blockIdx.y, entry = blockIdx.x, E = 2 * a.n_vp;
    const int s = a.s, V3 = s * s * s;

    // ---- which view is this, is it the first occurrence, how often does it occur (np.unique + inverse, :201,:255)
    const int64_t *pl = a.pairs + (size_t)cube * E;
    long long view = pl[entry];
    if (view < 0) view += a.V;
    unsigned mult = 0;
    for (int e = 0; e < E; ++e) {
        long long v = pl[e];
        if (v < 0) v += a.V;
        if (v == view) {
            if (e < entry) return;   // an earlier entry of this cube handles the view (uniform across the workgroup)
            ++mult;
        }
    }
    if (view < 0 || view >= a.V) return;   // rejected on the host for host arrays; device arrays are the caller's contract

    const float *pred = a.pred + (size_t)cube * V3;
    const _Float16 thr = (_Float16)a.thresh;
    const double r = (double)a.resol[cube];
    const double x0 = (double)a.xyz[3 * cube + 0], y0 = (double)a.xyz[3 * cube + 1], z0 = (double)a.xyz[3 * cube + 2];
    double P[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) P[i] = a.cams[12 * view + i];

    auto selected = [&](int i, _Float16 &p) { p = (_Float16)pred[i]; return !a.use_thresh || p > thr; };

    // ---- pass 1: count the selected voxels and list them (any order: every later step is a max / an add). With the list (up to
    // RP_LIST voxels, i.e. every realistic surface) the passes below touch only the selected voxels with all lanes busy; without it
    // a wave runs the fp64 projection whenever ANY of its 64 lanes holds a selected voxel - 32 times per lane at 1 % selected as at 10 %.
    __shared__ __attribute__((aligned(16))) unsigned char sh_mem[RP_LDS_BYTES];
    unsigned *const sh_list = reinterpret_cast<unsigned *>(sh_mem);
    // more than the 64 KiB of LDS a workgroup gets on CDNA1-3 - this library is built for gfx950 only (Makefile: 160 KiB per
    // CU, and the 1024-thread workgroup owns its CU anyway). Halve RP_LIST / RP_LDS_M before retargeting.
    static_assert(RP_LDS_BYTES + 64 <= 160 * 1024, "ray_pool_kernel's lists and tables exceed gfx950's LDS");
    if (tid == 0) { sh_i[0] = 0; sh_i[1] = 0x7fffffff; sh_i[2] = 0; }
    __syncthreads();
    // (eight 16-byte loads per thread in flight at once: one load per thread and iteration, each waited for before the next, made this scan of a
    // 128 KB cube ~100 us - 32 dependent memory round trips - i.e. most of the kernel at realistic selection rates)
    if ((V3 & 3) == 0 && (reinterpret_cast<size_t>(pred) & 15) == 0) {
        for (int base = 0; base < V3; base += 8 * 4 * RP_NT) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i4 = base + (k * RP_NT + tid) * 4;
                v[k] = i4 < V3 ? *reinterpret_cast<const float4 *>(pred + i4) : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i4 = base + (k * RP_NT + tid) * 4;
                if (i4 >= V3) continue;
                const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const _Float16 p = (_Float16)e[j];
                    if (!a.use_thresh || p > thr) {
                        const int pos = atomicAdd(&sh_i[0], 1);
                        if (pos < RP_LIST) sh_list[pos] = (unsigned)(i4 + j);
                    }
                }
            }
        }
    } else
    for (int i = tid; i < V3; i += RP_NT) {
        _Float16 p;
        if (selected(i, p)) {
            const int pos = atomicAdd(&sh_i[0], 1);
            if (pos < RP_LIST) sh_list[pos] = (unsigned)i;
        }
    }
    __syncthreads();
    const int m = sh_i[0];
    if (m == 0) return;
    const bool listed = m <= RP_LIST;
    const int n_it = listed ? m : V3;                       // iteration space of passes 3-5: list positions, or all voxels
    unsigned cap = 64;
    while (cap < 2u * (unsigned)m) cap <<= 1;
    if (m <= RP_LDS_M && cap > (unsigned)RP_LDS_CAP) cap = RP_LDS_CAP;       // (LDS tables: a fuller table instead of a global one)
    const unsigned mask = cap - 1;

    const size_t wg = (size_t)cube * E + entry;
    const bool lds_tables = m <= RP_LDS_M;                  // (uniform across the workgroup; implies `listed`)
    unsigned *cslot = a.cslot + wg * V3;
    unsigned *const sh_q = reinterpret_cast<unsigned *>(sh_mem + (lds_tables ? RP_LDS_Q : RP_LIST * 4));
    uint8_t *votes = a.votes + (size_t)cube * V3;

    // passes 2-5 over one set of tables (instantiated for LDS and for global tables)
    auto body = [&](auto pix_key, auto pix_best, auto cell_key, auto cell_idx) {
    // ---- pass 2: clear the part of the tables this workgroup uses
    for (unsigned i = tid; i < cap; i += RP_NT) { pix_key[i] = RP_EMPTY; pix_best[i] = 0; cell_key[i] = RP_EMPTY; cell_idx[i] = 0; }
    __syncthreads();

    // ---- pass 3: project, insert pixel and cell, keep the largest voxel index per cell
    // (measured and dropped, round 4: all of a thread's items advanced in lock-step, their k-th table accesses issued together - no faster with global tables,
    // slower with LDS tables; plain L1-cached probe loads with acquire fences between the passes - slower)
    int dmin_t = 0x7fffffff;
    for (int j = tid; j < n_it; j += RP_NT) {
        const int i = listed ? (int)sh_list[j] : j;
        _Float16 p;
        unsigned q = RP_NONE;
        if (listed || selected(i, p)) {
            const int kz = i % s, jy = (i / s) % s, ix = i / (s * s);
            const double X = __dadd_rn(__dmul_rn((double)ix, r), x0);
            const double Y = __dadd_rn(__dmul_rn((double)jy, r), y0);
            const double Z = __dadd_rn(__dmul_rn((double)kz, r), z0);
            const double q0 = sn_dot4(P, X, Y, Z), q1 = sn_dot4(P + 4, X, Y, Z), q2 = sn_dot4(P + 8, X, Y, Z);
            const double w = rint(__ddiv_rn(q0, q2)), h = rint(__ddiv_rn(q1, q2)), d = rint(__ddiv_rn(q2, r));
            // pixel key = (w, h) with the sign bits flipped: (-1,-1), a legal out-of-image pixel (the reference's ray pooling has no
            // in-image test), would otherwise equal the RP_EMPTY sentinel; the one pair that still does, (2^31-1, 2^31-1), is
            // rejected through the err path together with everything outside the int32 range
            const double lim = 2147483647.0, limp = 2147483646.0;
            if (!(w >= -lim && w <= limp && h >= -lim && h <= limp && d >= -lim && d <= lim)) {
                *a.err = 1;
            } else {
                const int wi = (int)w, hi = (int)h, di = (int)d;
                const unsigned ps = rp_insert(pix_key, mask, ((unsigned long long)((unsigned)wi ^ 0x80000000u) << 32) | ((unsigned)hi ^ 0x80000000u));
                q = rp_insert(cell_key, mask, ((unsigned long long)ps << 32) | (unsigned)di);
                rp_max(cell_idx + q, (unsigned)i);
                dmin_t = min(dmin_t, di);
            }
        }
        if (listed) sh_q[j] = q; else cslot[i] = q;
    }
    for (int o = 32; o; o >>= 1) dmin_t = min(dmin_t, __shfl_xor(dmin_t, o));
    if ((tid & 63) == 0) atomicMin(&sh_i[1], dmin_t);
    __syncthreads();
    const int dmin = sh_i[1];

    auto packed_of = [&](_Float16 p, int d) {
        const unsigned pb = (unsigned)__builtin_bit_cast(unsigned short, p);     // p > 0: fp16 bits are monotone
        return ((unsigned long long)pb << 32) | (0xFFFFFFFFu - (unsigned)(d - dmin));
    };

    // ---- pass 4: every cell's voxel bids for its pixel
    for (int j = tid; j < n_it; j += RP_NT) {
        const int i = listed ? (int)sh_list[j] : j;
        const unsigned q = listed ? sh_q[j] : cslot[i];
        if (q == RP_NONE || rp_ld(cell_idx + q) != (unsigned)i) continue;
        const _Float16 p = (_Float16)pred[i];
        if (!(p > (_Float16)0.f)) continue;                 // a stored 0 is indistinguishable from an empty cell
        const unsigned long long ck = rp_ld(cell_key + q);
        rp_max(pix_best + (unsigned)(ck >> 32), packed_of(p, (int)(unsigned)ck));
    }
    __syncthreads();

    // ---- pass 5: winners vote
    for (int j = tid; j < n_it; j += RP_NT) {
        const int i = listed ? (int)sh_list[j] : j;
        const unsigned q = listed ? sh_q[j] : cslot[i];
        if (q == RP_NONE || rp_ld(cell_idx + q) != (unsigned)i) continue;
        const _Float16 p = (_Float16)pred[i];
        const unsigned long long ck = rp_ld(cell_key + q);
        const unsigned ps = (unsigned)(ck >> 32);
        const int d = (int)(unsigned)ck;
        const unsigned long long best = rp_ld(pix_best + ps);
        int target = -1;
        if (p > (_Float16)0.f) {
            if (best == packed_of(p, d)) target = i;
        } else if (best == 0) {                              // every stored value of this pixel is 0: argmax -> column 0
            if (d == dmin) target = i;
            else if (!rp_find(cell_key, mask, ((unsigned long long)ps << 32) | (unsigned)dmin)) target = 0;
        }
        if (target > 0) rp_vote(votes, (unsigned)target, mult);
        else if (target == 0) sh_i[2] = 1;                  // several pixels may elect voxel 0: count it once
    }
    };
    if (lds_tables) {
        rp_lds_u64 *const t = (rp_lds_u64 *)(sh_mem + RP_LDS_TAB);            // (C-style casts: generic -> LDS address space)
        body(t, t + RP_LDS_CAP, t + 2 * RP_LDS_CAP, (rp_lds_u32 *)(t + 3 * RP_LDS_CAP));
    } else
        body(a.pix_key + wg * a.cap_max, a.pix_best + wg * a.cap_max, a.cell_key + wg * a.cap_max, a.cell_idx + wg * a.cap_max);
    __syncthreads();
    if (tid == 0 && sh_i[2]) rp_vote(votes, 0u, mult);
}

// ------------------------------------------------------------------------------------------------
// point projection (camera.perspectiveProj, utils/camera.py:123-184): every bound camera x every point.
// One thread = one (view, point): q = P.[x y z 1] as the FMA chain over k (what numpy's matmul -> dgemm computes), IEEE
// divides by q2, optional rint (numpy .round(): half-to-even). Outputs are float64 (V, n) row-major; lanes run along the
// points, so the three loads per lane are strided by 24 B and the stores are contiguous.
// ------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) project_points_kernel(const double *cams, const double *xyz, int n, int round_int,
                                                                    double *out_h, double *out_w, double *out_depth)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *P = cams + (size_t)blockIdx.y * 12;
    const double X = xyz[3 * i], Y = xyz[3 * i + 1], Z = xyz[3 * i + 2];
    const double q0 = sn_dot4(P, X, Y, Z), q1 = sn_dot4(P + 4, X, Y, Z), q2 = sn_dot4(P + 8, X, Y, Z);
    double w = __ddiv_rn(q0, q2), h = __ddiv_rn(q1, q2);
    if (round_int) { w = rint(w); h = rint(h); }
    const size_t o = (size_t)blockIdx.y * n + i;
    out_h[o] = h; out_w[o] = w;
    if (out_depth) out_depth[o] = q2;
}

// ------------------------------------------------------------------------------------------------
// dense2sparse
// ------------------------------------------------------------------------------------------------
struct SparseArgs {
    const float *pred;         // (n, s,s,s) float32
    const uint8_t *rgb;        // (n, 3, s,s,s) uint8 (sn_color_fuse output) or nullptr
    const uint8_t *votes;      // (n, s,s,s) or nullptr (ray pooling disabled)
    long long *offsets;        // (n+1,) exclusive prefix of the per-cube counts
    int *counts;               // (n,)
    uint8_t *ijk;              // (total, 3) voxel index inside the (cropped) cube
    uint16_t *pred16;          // (total,) fp16 bits
    uint8_t *rgb_out;          // (total, 3)
    uint8_t *votes_out;        // (total,)
    int s, lo, dc;             // cube edge, crop offset, cropped edge
    int by_votes, vote_thresh; // keep rule: votes >= vote_thresh (sparseCubes.py:60) or fp16(pred) > fp16(min_prob) (:62)
    float min_prob;
};

constexpr int D2S_NT = 1024;

__device__ __forceinline__ bool d2s_keep(const SparseArgs &a, int cube, int t, int &src)
{
    const int dc = a.dc, s = a.s;
    const int k = t % dc, j = (t / dc) % dc, i = t / (dc * dc);
    src = ((i + a.lo) * s + (j + a.lo)) * s + (k + a.lo);
    const size_t g = (size_t)cube * s * s * s + src;
    if (a.by_votes) return (int)a.votes[g] >= a.vote_thresh;
    return (_Float16)a.pred[g] > (_Float16)a.min_prob;
}

__global__ void __launch_bounds__(D2S_NT) d2s_count_kernel(SparseArgs a)
{
    __shared__ int red[D2S_NT / 64];
    const int cube = blockIdx.x, tid = threadIdx.x, T = a.dc * a.dc * a.dc;
    int cnt = 0, src;
    for (int t = tid; t < T; t += D2S_NT) cnt += d2s_keep(a, cube, t, src) ? 1 : 0;
    for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & 63) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) { int m = 0; for (int i = 0; i < D2S_NT / 64; ++i) m += red[i]; a.counts[cube] = m; }
}

__global__ void d2s_scan_kernel(const int *counts, long long *offsets, int n)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        long long acc = 0;
        for (int i = 0; i < n; ++i) { offsets[i] = acc; acc += counts[i]; }
        offsets[n] = acc;
    }
}

// Ordered compaction: voxels leave in ascending flat index of the cropped cube (np.where order, sparseCubes.py:60-62).
__global__ void __launch_bounds__(D2S_NT) d2s_write_kernel(SparseArgs a)
{
    __shared__ int wave_cnt[D2S_NT / 64];
    __shared__ int base_sh;
    const int cube = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, T = a.dc * a.dc * a.dc;
    const int s3 = a.s * a.s * a.s;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    const long long off = a.offsets[cube];
    for (int t0 = 0; t0 < T; t0 += D2S_NT) {
        const int t = t0 + tid;
        int src = 0;
        const bool keep = t < T && d2s_keep(a, cube, t, src);
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
        for (int i = 0; i < D2S_NT / 64; ++i) { const int c = wave_cnt[i]; if (i < wv) before += c; total += c; }
        const int base = base_sh;
        if (keep) {
            const long long o = off + base + before + __popcll(bal & ((1ull << lane) - 1ull));
            const int dc = a.dc;
            a.ijk[3 * o + 0] = (uint8_t)(t / (dc * dc));
            a.ijk[3 * o + 1] = (uint8_t)((t / dc) % dc);
            a.ijk[3 * o + 2] = (uint8_t)(t % dc);
            const size_t g = (size_t)cube * s3 + src;
            a.pred16[o] = __builtin_bit_cast(unsigned short, (_Float16)a.pred[g]);
            if (a.rgb) {
                const uint8_t *c = a.rgb + (size_t)cube * 3 * s3 + src;
                a.rgb_out[3 * o + 0] = c[0]; a.rgb_out[3 * o + 1] = c[s3]; a.rgb_out[3 * o + 2] = c[2 * (size_t)s3];
            }
            if (a.votes && a.votes_out) a.votes_out[o] = a.votes[g];
        }
        __syncthreads();
        if (tid == 0) base_sh = base + total;
        __syncthreads();
    }
}

}  // namespace sn
