// tools/probe/mxmix_probe.hip - issue rate of v_mfma_scale_f32_16x16x128_f8f6f4 for MIXED operand formats: A = weights (cbsz), B = activations (blgp),
// each fp8 e4m3 (0), fp6 e2m3 (2) or fp4 (4). The question (round 5): does a 6-bit A operand with an 8-bit B operand run at the 6-bit rate (16 clocks) or at
// the 8-bit rate (32)? - 8-bit activation codes have the exponent range the static-scaled 6-bit ones lack (real pixels saturate them), 6-bit block-scaled
// weights do not need it. One wave per SIMD on every CU, 8 independent tied accumulators in AGPRs, clocks per instruction from the shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DEFK(NAME, TA, NA, TB, NB, FMT)                                                                                                             \
    __global__ void __launch_bounds__(256) NAME(float *out, const int *src, int iters, unsigned long long *clk)                                      \
    {                                                                                                                                               \
        const int lane = threadIdx.x;                                                                                                               \
        TA a[2]; TB b[2];                                                                                                                           \
        for (int j = 0; j < 2; ++j) {                                                                                                               \
            for (int e = 0; e < NA; ++e) a[j][e] = src[(lane * 8 + e + 97 * j) & 4095] & 0x3f3f3f3f;                                                \
            for (int e = 0; e < NB; ++e) b[j][e] = src[(lane * 8 + e + 1031 * j + 7) & 4095] & 0x3f3f3f3f;                                           \
        }                                                                                                                                           \
        f32x4 acc[8] = {};                                                                                                                          \
        int sc = 127;                                                                                                                               \
        asm volatile("" : "+v"(sc));                                                                                                                \
        const long long t0 = __builtin_readcyclecounter();                                                                                          \
        for (int it = 0; it < iters; ++it) {                                                                                                        \
            _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                                                           \
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] " FMT : "+a"(acc[q]) : "v"(a[q & 1]), "v"(b[(q >> 1) & 1]), "v"(sc)); \
        }                                                                                                                                           \
        const long long t1 = __builtin_readcyclecounter();                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = (unsigned long long)(t1 - t0);                                                            \
        float s = 0.f;                                                                                                                              \
        for (int q = 0; q < 8; ++q) s += acc[q][0];                                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                                             \
    }
DEFK(k88, v8i, 8, v8i, 8, "")
DEFK(k66, v6i, 6, v6i, 6, "cbsz:2 blgp:2")
DEFK(k68, v6i, 6, v8i, 8, "cbsz:2")
DEFK(k86, v8i, 8, v6i, 6, "blgp:2")
DEFK(k48, v4i, 4, v8i, 8, "cbsz:4")
DEFK(k46, v4i, 4, v6i, 6, "cbsz:4 blgp:2")

typedef void (*kern_t)(float *, const int *, int, unsigned long long *);
static void run(const char *name, kern_t kf, float *out, const int *src, unsigned long long *clk)
{
    const int iters = 200000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f; unsigned long long h = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kf, dim3(256), dim3(256), 0, 0, out, src, iters, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-34s %8.3f ms  %6.1f clocks per instruction and wave (%.2f GHz)\n", name, ms, (double)h / (8.0 * iters), h / (ms * 1e-3) / 1e9);
}

int main()
{
    float *out; int *src; unsigned long long *clk;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&src, 4096 * 4); (void)hipMalloc(&clk, 8);
    static int h[4096];
    unsigned x = 12345u;
    for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = (int)x; }
    (void)hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    run("A fp8 x B fp8", k88, out, src, clk);
    run("A fp6 x B fp6", k66, out, src, clk);
    run("A fp6 (weights) x B fp8 (act.)", k68, out, src, clk);
    run("A fp8 x B fp6", k86, out, src, clk);
    run("A fp4 x B fp8", k48, out, src, clk);
    run("A fp4 x B fp6", k46, out, src, clk);
    return 0;
}
