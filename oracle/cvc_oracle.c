/*
 * oracle/cvc_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's colored-voxel-cube (CVC) construction, used only as the
 * checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Nothing under surfacenet_amd/ may link, import or call this file.
 *
 * Follows (reference file:line, all under /root/reference):
 *   utils/CVC.py:12-20   voxel-centre grid: idx*resol + min  (int64 * f32 scalar -> f64 arithmetic on
 *                        f32-rounded constants; multiply and add are two separately rounded f64 ops)
 *   utils/CVC.py:36-40   pts_3D = P(3x4,f64) . [X Y Z 1]^T ; u = q0/q2 ; v = q1/q2 ;
 *                        round-half-even -> int32 ; w = row 0, h = row 1
 *   utils/CVC.py:42-47   in-scope iff 0 <= w < W and 0 <= h < H (no depth test); gather img[h,w,:]
 *                        else 0; channel-planar (3,s,s,s), flat voxel index i*s*s + j*s + k
 *   utils/CVC.py:56-104  per cube: views = pairs.flatten(); output (n*N_vp, 6, s,s,s) float32
 *   utils/CVC.py:108-111 preprocess: X.astype(f32) - mean_rgb (broadcast over channel axis)
 *
 * Parity pin: tests/golden/cvc_*.npz were produced by the reference's own CVC.py executed in the
 * build container (oracle/gen_golden.py); tests/test_oracle_cvc.py checks this file against them.
 *
 * Dot-product order: np.dot(3x4, 4xN) lowers to BLAS dgemm; with SN_ORACLE_FMA (default) the
 * K=4 accumulation is the FMA chain t=P0*X; t=fma(P1,Y,t); t=fma(P2,Z,t); t=fma(P3,1,t), which is
 * what the dgemm micro-kernels of the build container's BLAS do (verified bit-for-bit in
 * tests/test_oracle_cvc.py::test_projection_chain_matches_numpy when that BLAS is present).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef SN_ORACLE_FMA
#define SN_ORACLE_FMA 1
#endif

static inline double dot4(const double *p, double X, double Y, double Z)
{
#if SN_ORACLE_FMA
    double t = p[0] * X;
    t = fma(p[1], Y, t);
    t = fma(p[2], Z, t);
    t = fma(p[3], 1.0, t);
    return t;
#else
    double t = p[0] * X;
    t = t + p[1] * Y;
    t = t + p[2] * Z;
    t = t + p[3];
    return t;
#endif
}

/* Projects one voxel centre; returns 1 and (*w,*h) if the rounded pixel is inside the image. */
static inline int project_px(const double *P, double X, double Y, double Z, int H, int W, int *w, int *h)
{
    double q0 = dot4(P + 0, X, Y, Z);
    double q1 = dot4(P + 4, X, Y, Z);
    double q2 = dot4(P + 8, X, Y, Z);
    double u = rint(q0 / q2); /* CVC.py:38-39; rint = round-half-even under the default FP mode */
    double v = rint(q1 / q2);
    /* int32 cast of non-finite / out-of-range values gives INT_MIN on the reference's x86 host,
     * i.e. "out of scope"; comparing in the double domain states the same rule portably. */
    if (!(u >= 0.0 && u < (double)W && v >= 0.0 && v < (double)H))
        return 0;
    *w = (int)u;
    *h = (int)v;
    return 1;
}

/*
 * out: (n*n_vp, 6, s,s,s) float32, C-contiguous.  mean6 == NULL -> raw 0..255 values
 * (gen_coloredCubes); mean6 != NULL -> value - mean6[c] in float32 (preprocess_augmentation with
 * augment_ON=False, crop_ON=False), so out-of-scope voxels become -mean.
 * imgs[v]: (H[v], W[v], 3) uint8 RGB.  P: (V,3,4) row-major float64.  view_pairs: (n, n_vp, 2) int64.
 * Returns 0, or -1 if a view id is outside [0,V) (the reference raises IndexError there).
 */
int sn_oracle_cvc(int n, int n_vp, int s, int V, const int64_t *view_pairs, const float *xyz,
                  const float *resol, const double *P, const uint8_t *const *imgs, const int *H,
                  const int *W, const float *mean6, float *out)
{
    const size_t s3 = (size_t)s * s * s;
    for (int c = 0; c < n; ++c) {
        const double r = (double)resol[c];
        const double x0 = (double)xyz[3 * c + 0], y0 = (double)xyz[3 * c + 1], z0 = (double)xyz[3 * c + 2];
        for (int p = 0; p < n_vp; ++p) {
            for (int side = 0; side < 2; ++side) {
                int64_t view = view_pairs[((size_t)c * n_vp + p) * 2 + side];
                if (view < 0) view += V; /* numpy negative indexing */
                if (view < 0 || view >= V) return -1;
                const double *Pv = P + 12 * view;
                const uint8_t *img = imgs[view];
                const int Hv = H[view], Wv = W[view];
                float *o = out + (((size_t)c * n_vp + p) * 6 + 3 * side) * s3;
                for (int i = 0; i < s; ++i) {
                    const double X = (double)i * r + x0; /* two roundings, no contraction */
                    for (int j = 0; j < s; ++j) {
                        const double Y = (double)j * r + y0;
                        for (int k = 0; k < s; ++k) {
                            const double Z = (double)k * r + z0;
                            const size_t vox = ((size_t)i * s + j) * s + k;
                            int w, h;
                            float rgb[3] = {0.f, 0.f, 0.f};
                            if (project_px(Pv, X, Y, Z, Hv, Wv, &w, &h)) {
                                const uint8_t *px = img + ((size_t)h * Wv + w) * 3;
                                rgb[0] = (float)px[0]; rgb[1] = (float)px[1]; rgb[2] = (float)px[2];
                            }
                            for (int ch = 0; ch < 3; ++ch)
                                o[ch * s3 + vox] = mean6 ? rgb[ch] - mean6[3 * side + ch] : rgb[ch];
                        }
                    }
                }
            }
        }
    }
    return 0;
}

/* camera.py:123-184 perspectiveProj restated for (N_Ms,3,4) x (N_pts,3): writes float (h,w) pairs
 * before rounding (hw_f, may be NULL) and rounded int64 (hw_i, may be NULL); layout (2, N_Ms, N_pts),
 * plane 0 = h, plane 1 = w. */
int sn_oracle_perspective_proj(int n_ms, int n_pts, const double *P, const double *xyz, double *hw_f, int64_t *hw_i)
{
    for (int m = 0; m < n_ms; ++m)
        for (int t = 0; t < n_pts; ++t) {
            const double *Pm = P + 12 * m;
            const double X = xyz[3 * t], Y = xyz[3 * t + 1], Z = xyz[3 * t + 2];
            double q0 = dot4(Pm, X, Y, Z), q1 = dot4(Pm + 4, X, Y, Z), q2 = dot4(Pm + 8, X, Y, Z);
            double u = q0 / q2, v = q1 / q2;
            size_t o = (size_t)m * n_pts + t, plane = (size_t)n_ms * n_pts;
            if (hw_f) { hw_f[o] = v; hw_f[plane + o] = u; }
            if (hw_i) { hw_i[o] = (int64_t)rint(v); hw_i[plane + o] = (int64_t)rint(u); }
        }
    return 0;
}

/* Exposes the projection numerators so the test can compare the summation chain with np.dot. */
void sn_oracle_dot34(int n_pts, const double *P, const double *pts4, double *out3)
{
    for (int t = 0; t < n_pts; ++t)
        for (int r = 0; r < 3; ++r)
            out3[(size_t)r * n_pts + t] = dot4(P + 4 * r, pts4[t], pts4[(size_t)n_pts + t], pts4[2 * (size_t)n_pts + t]);
}
