// tools/probe/power_probe.hip - what the chip SUSTAINS (time, not clocks) on long f16 MFMA streams: v_mfma_f32_16x16x32_f16 against v_mfma_f32_32x32x16_f16,
// operands in registers or re-read from LDS at the merge loop's rate (15 x 16-byte reads per 56 MFMAs). All 256 CUs, one or two waves per SIMD, ~4 ms
// per launch (DVFS settles within the first launches; the last of 5 is reported). Random fp16 operands (zero operands draw less power).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int LDSR>
__global__ void __launch_bounds__(512) k(float *out, const _Float16 *src, int iters, unsigned long long *clk)
{
    __shared__ half8 sh[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = *reinterpret_cast<const half8 *>(src + (size_t)i * 8);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    half8 a[4], b[4];
    for (int j = 0; j < 4; ++j) { a[j] = sh[(lane + 64 * j) & 4095]; b[j] = sh[(lane * 3 + 64 * j + 1000) & 4095]; }
    f32x4 acc4[14] = {};
    f32x16 acc16[4] = {};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (SHAPE == 16) {
#pragma unroll
            for (int q = 0; q < 56; ++q) {
                acc4[q % 14] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q & 3], b[(q >> 2) & 3], acc4[q % 14], 0, 0, 0);
                if constexpr (LDSR) { if (q < 15) { if (q & 1) a[(q >> 1) & 3] = sh[(lane + 64 * q + it) & 4095]; else b[(q >> 1) & 3] = sh[(lane * 3 + 64 * q + it) & 4095]; } }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 28; ++q) {      // 28 x 32x32x16 = the MACs of 56 x 16x16x32
                acc16[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q & 3], b[(q >> 2) & 3], acc16[q & 3], 0, 0, 0);
                if constexpr (LDSR) { if (q < 8) { if (q & 1) a[(q >> 1) & 3] = sh[(lane + 64 * q + it) & 4095]; else b[(q >> 1) & 3] = sh[(lane * 3 + 64 * q + it) & 4095]; } }   // half the operand bytes per MAC
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = (unsigned long long)(t1 - t0);
    float s = 0.f;
    for (int q = 0; q < 14; ++q) s += acc4[q][0];
    for (int q = 0; q < 4; ++q) s += acc16[q][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int LDSR>
void run(const char *name, int threads, float *out, const _Float16 *src, unsigned long long *clk)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f; unsigned long long h = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, LDSR>), dim3(256), dim3(threads), 0, 0, out, src, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double flops = 2.0 * 16 * 16 * 32 * 56.0 * iters * (threads / 64) * 256;
    printf("%-44s %d waves/SIMD  %7.3f ms  %7.1f TF  clocks %llu -> %.2f GHz, %.1f clocks per 16x16x32-equivalent and wave\n", name, threads / 256, ms, flops / (ms * 1e-3) / 1e12, h,
           h / (ms * 1e-3) / 1e9, (double)h / (56.0 * iters));
}

int main()
{
    float *out; _Float16 *src; unsigned long long *clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&src, 4096 * 16); hipMalloc(&clk, 8);
    _Float16 *hsrc = new _Float16[4096 * 8];
    unsigned st = 12345;
    for (int i = 0; i < 4096 * 8; ++i) { st = st * 1664525u + 1013904223u; hsrc[i] = (_Float16)(((int)(st >> 16) % 2001 - 1000) * 0.001f); }
    hipMemcpy(src, hsrc, 4096 * 16, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        run<16, 0>("16x16x32, operands in registers", threads, out, src, clk);
        run<32, 0>("32x32x16, operands in registers", threads, out, src, clk);
        run<16, 1>("16x16x32 + 15 LDS reads per 56 MFMAs", threads, out, src, clk);
        run<32, 1>("32x32x16 + 8 LDS reads per 28 MFMAs", threads, out, src, clk);
    }
    return 0;
}
