// fp6_probe.hip — facts about the 6-bit MX operand forms on gfx950 that conv3d_mfma.h's f16m8 step relies on.
//   (1) v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 (e2m3, format 2) / bf6 (e3m2, format 3) operands: a lane's 32 K-elements are the
//       192 bits of its first 6 operand VGPRs, element i at bits [6i, 6i+6); lane l = (row l&15, K block l>>4) as for fp8; every
//       lane supplies its own E8M0 scale (byte 0 of the scale VGPR with op_sel 0); A and B formats are independent.
//   (2) v_cvt_scalef32_2xpk16_{fp6,bf6}_f32 / v_cvt_scalef32_pk32_fp6_f16: element order of the result, direction of the scale,
//       rounding and saturation.
// Build: hipcc --offload-arch=gfx950 -O2 -o fp6_probe fp6_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));

template <int FA, int FB>
__global__ void kmfma(const unsigned *A, const unsigned *B, float *D, const int *sa, const int *sb)
{
    const int l = threadIdx.x;
    v8i a, b;
    for (int d = 0; d < 8; ++d) { a[d] = d < 6 ? (int)A[l * 6 + d] : 0x5a5a5a5a; b[d] = d < 6 ? (int)B[l * 6 + d] : 0x3c3c3c3c; }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FA, FB, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

__global__ void kcvt(const float *in, unsigned *out, float sc)
{
    v16f a, b; v32h h;
    for (int i = 0; i < 16; ++i) { a[i] = in[i]; b[i] = in[16 + i]; h[i] = (_Float16)in[i]; h[16 + i] = (_Float16)in[16 + i]; }
    v6i r0 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, sc);
    v6i r1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, sc);
    v6i r2 = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, sc);
    for (int i = 0; i < 6; ++i) { out[i] = r0[i]; out[6 + i] = r1[i]; out[12 + i] = r2[i]; }
}

static float dec6(int code, int fmt)       // fmt 2: e2m3 (bias 1), 3: e3m2 (bias 3)
{
    const int s = code >> 5, mb = fmt == 2 ? 3 : 2, E = (code & 31) >> mb, M = code & ((1 << mb) - 1), bias = fmt == 2 ? 1 : 3;
    const float v = E == 0 ? ldexpf((float)M, 1 - bias - mb) : ldexpf((float)((1 << mb) + M), E - bias - mb);
    return s ? -v : v;
}
static void put6(unsigned *w, int i, int code) { for (int b = 0; b < 6; ++b) if (code >> b & 1) w[(6 * i + b) >> 5] |= 1u << ((6 * i + b) & 31); }
static int get6(const unsigned *w, int i) { int c = 0; for (int b = 0; b < 6; ++b) c |= (int)(w[(6 * i + b) >> 5] >> ((6 * i + b) & 31) & 1) << b; return c; }

// (3) accumulation: D = C + sum with a LARGE C and small products — is the block sum added to the fp32 accumulator with one rounding?
__global__ void kacc(const unsigned *A, const unsigned *B, const float *Cin, float *D, int sa, int sb)
{
    const int l = threadIdx.x;
    v8i a, b;
    for (int d = 0; d < 8; ++d) { a[d] = d < 6 ? (int)A[l * 6 + d] : 0; b[d] = d < 6 ? (int)B[l * 6 + d] : 0; }
    v4f c;
    for (int r = 0; r < 4; ++r) c[r] = Cin[((l >> 4) * 4 + r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
static void run_acc()
{
    std::vector<unsigned> A(64 * 6, 0), B(64 * 6, 0);
    std::vector<float> Af(16 * 128), Bf(16 * 128), C(256);
    unsigned rng = 777;
    const int sa = 127 - 21, sb = 127;     // products are multiples of 2^-27: far below the ulp of C (2^-18 .. 2^-17)
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) {
            rng = rng * 1664525u + 1013904223u; const int ca = rng >> 26;
            rng = rng * 1664525u + 1013904223u; const int cb = rng >> 26;
            put6(&A[l * 6], i, ca); put6(&B[l * 6], i, cb);
            Af[(l & 15) * 128 + (l >> 4) * 32 + i] = dec6(ca, 2) * ldexpf(1.f, sa - 127);
            Bf[(l & 15) * 128 + (l >> 4) * 32 + i] = dec6(cb, 2) * ldexpf(1.f, sb - 127);
        }
    for (int i = 0; i < 256; ++i) { rng = rng * 1664525u + 1013904223u; C[i] = 40.f + (float)(rng >> 8) * (1.0f / 16777216.f) * 60.f; }
    unsigned *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, 64 * 24); hipMalloc(&dB, 64 * 24); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 64 * 24, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 24, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kacc, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, sa, sb);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double max_ulp = 0, mean_signed = 0, max_sum = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double sum = 0; for (int k = 0; k < 128; ++k) sum += (double)Af[i * 128 + k] * Bf[j * 128 + k];
        const double ref = (double)C[i * 16 + j] + sum, ulp = ldexp(1.0, ilogb(ref) - 23);
        const double e = ((double)D[i * 16 + j] - ref) / ulp;
        max_ulp = fmax(max_ulp, fabs(e)); mean_signed += e / 256; max_sum = fmax(max_sum, fabs(sum));
    }
    printf("accumulate: C in [40,100), |block sums| up to %g: max |D - (C + sum)| = %.3f ulp of the result, mean signed error %.3f ulp "
           "(one correctly rounded addition: <= 0.5 / ~0)\n", max_sum, max_ulp, mean_signed);
}

template <int FA, int FB>
static void run_mfma()
{
    std::vector<unsigned> A(64 * 6, 0), B(64 * 6, 0);
    std::vector<int> sa(64), sb(64);
    std::vector<float> Af(16 * 128), Bf(16 * 128);      // [row][k], [col][k]
    unsigned rng = 12345;
    for (int l = 0; l < 64; ++l) {
        sa[l] = 127 - (l % 5) | 0x7f00;                 // byte 0 is the scale; garbage in byte 1 must be ignored with op_sel 0
        sb[l] = 127 + (l % 3) - 1;
        for (int i = 0; i < 32; ++i) {
            rng = rng * 1664525u + 1013904223u; const int ca = rng >> 26;
            rng = rng * 1664525u + 1013904223u; const int cb = rng >> 26;
            put6(&A[l * 6], i, ca); put6(&B[l * 6], i, cb);
            Af[(l & 15) * 128 + (l >> 4) * 32 + i] = dec6(ca, FA) * ldexpf(1.f, (sa[l] & 255) - 127);
            Bf[(l & 15) * 128 + (l >> 4) * 32 + i] = dec6(cb, FB) * ldexpf(1.f, (sb[l] & 255) - 127);
        }
    }
    unsigned *dA, *dB; float *dD; int *dsa, *dsb;
    hipMalloc(&dA, 64 * 24); hipMalloc(&dB, 64 * 24); hipMalloc(&dD, 1024); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    hipMemcpy(dA, A.data(), 64 * 24, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 24, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((kmfma<FA, FB>), dim3(1), dim3(64), 0, 0, dA, dB, dD, dsa, dsb);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ref = 0; for (int k = 0; k < 128; ++k) ref += (double)Af[i * 128 + k] * Bf[j * 128 + k];
        maxerr = fmax(maxerr, fabs(ref - D[i * 16 + j])); maxref = fmax(maxref, fabs(ref));
    }
    printf("mfma formats A=%d B=%d, per-lane scales: max |D - ref| = %g (max |ref| %g)\n", FA, FB, maxerr, maxref);
}

int main()
{
    run_mfma<2, 2>(); run_mfma<3, 3>(); run_mfma<2, 3>(); run_mfma<3, 2>();
    run_acc();
    // conversions
    std::vector<float> in(32);
    const float tv[32] = {0.f, 0.0625f, 0.125f, 0.1875f, 0.25f, 0.3f, 0.5f, 0.75f, 1.f, 1.0625f, 1.1875f, 1.9f, 1.97f, 2.f, 2.125f, 2.375f,
                          3.f, 3.9f, 4.f, 4.25f, 4.75f, 7.f, 7.5f, 7.8f, 9.f, 30.f, -0.125f, -1.f, -3.3f, -7.5f, -100.f, 1e-3f};
    for (int i = 0; i < 32; ++i) in[i] = tv[i];
    float *din; unsigned *dout;
    hipMalloc(&din, 128); hipMalloc(&dout, 18 * 4);
    hipMemcpy(din, in.data(), 128, hipMemcpyHostToDevice);
    for (float sc : {1.f, 4.f, 0.25f}) {
        hipLaunchKernelGGL(kcvt, dim3(1), dim3(1), 0, 0, din, dout, sc);
        unsigned o[18];
        hipMemcpy(o, dout, 72, hipMemcpyDeviceToHost);
        printf("scale %g\n  in      :", sc); for (int i = 0; i < 32; ++i) printf(" %g", tv[i]);
        printf("\n  2xpk16 fp6_f32:"); for (int i = 0; i < 32; ++i) printf(" %g", dec6(get6(o, i), 2));
        printf("\n  pk32 fp6_f16  :"); for (int i = 0; i < 32; ++i) printf(" %g", dec6(get6(o + 6, i), 2));
        printf("\n  2xpk16 bf6_f32:"); for (int i = 0; i < 32; ++i) printf(" %g", dec6(get6(o + 12, i), 3));
        printf("\n");
    }
    return 0;
}
