// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, E8M0 block scales) operand layout on gfx950.
// D[i][j] = sum_k A[i][k] * B[k][j] * 2^(sa-127) * 2^(sb-127). Hypothesis (from the f16 16x16x32 layout): lane l holds
// A[i = l&15][k = (l>>4)*32 + 0..31] as 32 bytes (8 dwords, byte order = k order); same for B with j = l&15; D: j = l&15,
// i = (l>>4)*4 + r. The host fills A,B with small exactly-representable fp8 integers and compares.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t *A, const uint8_t *B, float *D, int sa, int sb)
{
    const int l = threadIdx.x;
    v8i a, b;
    const uint8_t *pa = A + (l & 15) * 128 + (l >> 4) * 32;   // A row-major [16][128]
    const uint8_t *pb = B + (l & 15) * 128 + (l >> 4) * 32;   // B stored as [j][k] (i.e. B^T row-major)
    for (int d = 0; d < 8; ++d) {
        a[d] = *(const int *)(pa + 4 * d);
        b[d] = *(const int *)(pb + 4 * d);
    }
    v4f c = {0, 0, 0, 0};
    // (a, b, c, cbsz = A format 0:fp8 e4m3, blgp = B format 0:fp8, opsel_a, scale_a, opsel_b, scale_b)
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

static uint8_t f8(float v)   // exact encoder for small values representable in e4m3 (bias 7)
{
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; v = fabsf(v);
    int e; float m = frexpf(v, &e);            // v = m * 2^e, m in [0.5,1)
    m *= 2; e -= 1;                            // m in [1,2)
    int mant = (int)roundf((m - 1.0f) * 8.0f);
    return s | (uint8_t)((e + 7) << 3) | (uint8_t)mant;
}

int main()
{
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    std::vector<float> Af(16 * 128), Bf(16 * 128);
    for (int i = 0; i < 16; ++i) for (int k2 = 0; k2 < 128; ++k2) {
        float va = (float)(((i * 7 + k2 * 3) % 9) - 4) * 0.5f, vb = (float)(((i * 5 + k2 * 11) % 7) - 3) * 0.25f;
        Af[i * 128 + k2] = va; Bf[i * 128 + k2] = vb; A[i * 128 + k2] = f8(va); B[i * 128 + k2] = f8(vb);
    }
    uint8_t *dA, *dB; float *dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int trial = 0; trial < 2; ++trial) {
        const int sa = trial == 0 ? 127 : 127 - 12, sb = 127;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb);
        std::vector<float> D(256);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        double maxerr = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double ref = 0; for (int k2 = 0; k2 < 128; ++k2) ref += (double)Af[i * 128 + k2] * Bf[j * 128 + k2];
            ref *= ldexp(1.0, sa - 127) * ldexp(1.0, sb - 127);
            maxerr = fmax(maxerr, fabs(ref - D[i * 16 + j]));
        }
        printf("trial %d (scale_a exp %d): max |D - ref| = %g ; D[0][0..3] = %g %g %g %g\n", trial, sa, maxerr, D[0], D[1], D[2], D[3]);
    }
    return 0;
}
