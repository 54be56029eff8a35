// tools/probe/mfma16_probe.hip - issue rate of the legacy v_mfma_f32_16x16x16_f16 against v_mfma_f32_16x16x32_f16 on gfx950 (is a K = 16 tail
// chunk cheaper than a zero-padded K = 32 one?). One wave per SIMD, 4 independent accumulators, 4096 MFMAs back to back.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int K>
__global__ void k(float *out, unsigned long long *clk)
{
    f32x4 acc[4] = {};
    half8 a8 = {}, b8 = {};
    half4 a4 = {}, b4 = {};
    for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(threadIdx.x * 0.01f + j); b8[j] = (_Float16)(1.f - j * 0.1f); }
    for (int j = 0; j < 4; ++j) { a4[j] = a8[j]; b4[j] = b8[j]; }
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 1024; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (K == 32) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[q], 0, 0, 0);
            else acc[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[q], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = (unsigned long long)(t1 - t0);
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
}
int main()
{
    float *out; unsigned long long *clk, h;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 8);
    hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 0, 0, out, clk); hipDeviceSynchronize(); hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("v_mfma_f32_16x16x32_f16: %.1f clocks per instruction\n", h / 4096.0);
    hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, out, clk); hipDeviceSynchronize(); hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("v_mfma_f32_16x16x16_f16: %.1f clocks per instruction\n", h / 4096.0);
    return 0;
}
