// tools/probe/uploader_probe.hip - what it would cost merge_conv_a to STOP reading the 48 upsampled channels of the concat buffer and to PRODUCE them
// instead, inside its halo staging (VERDICT r5 "stop materialising the concat tensor, or publish the costing as a measurement"; nets/SurfaceNet.py:48,57,68,71,
// nets/layers.py:376-390).
//
// merge_conv_a runs as 4-wave workgroups, one wave per SIMD (conv3d_mfma.h, one-wave-per-SIMD loop); a channel slab = one 8-channel group of the 10x10x10 halo tile
// of an 8x8x8 output tile: 1,000 voxels, 250 per wave, staged today by 8 LDS-DMA instructions per wave that sit behind MFMAs of the previous slab (16 + 16 KiB: the
// fp16 plane and the 6-bit code plane). The loader variant replaces the DMAs of 6 of the 8 slabs by: the closed-form interpolation stencil of Bilinear_3DInterpolation
// (<= 8 corners of the low-resolution side output, fp32 sources staged in LDS once per tile as the shipped upsample3_cat_tiled_kernel does), the hi / 6-bit-code
// split of the result (sn_store8<2>: the same conversion the shipped upsampler ends with) and two 16-byte LDS writes per voxel. This probe runs exactly that work -
// the product's own up_axis / sn_store8<2> from surfacenet_amd/csrc/elementwise.h, one 4-wave workgroup per CU, 160 KB of LDS so that a wave owns its SIMD - and
// reports shader clocks per slab and wave, for the x2 map (side 2) and the x4 maps (sides 3 / 4). Compare with a slab of the real loop: 3.375 weight pieces of
// ~3,450 clocks = ~11,600 clocks, every issue slot behind its 168 MFMAs per piece already taken.
//   hipcc -O3 --offload-arch=gfx950 -I surfacenet_amd/csrc -o tools/probe/uploader_probe tools/probe/uploader_probe.hip && tools/probe/uploader_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "elementwise.h"

using namespace sn;

template <int F>
__global__ void __launch_bounds__(256) loader_kernel(const float *seed, unsigned long long *clk, float *sink, int reps, int e8)
{
    // fp32 sources of one tile's stencils: (10 / F + 2)^3 voxels of 8 channels (x2: 7^3, x4: 5^3 - 4 would do for an aligned tile; 5 covers any origin)
    constexpr int NS = 10 / F + 2;
    __shared__ float src[NS * NS * NS * 8];
    __shared__ __attribute__((aligned(16))) char halo[2 * 16384 + 110 * 1024];      // the slab's two planes (+ padding up to the real kernel's LDS footprint: one workgroup per CU)
    for (int i = threadIdx.x; i < NS * NS * NS * 8; i += 256) src[i] = seed[i & 4095];
    __syncthreads();
    const int Di = 32 / F;
    const int x0 = 8, y0 = 16, z0 = 8;                                           // an interior tile of a 32^3 volume
    const int mx0 = (x0 - 1) / F, my0 = (y0 - 1) / F, mz0 = (z0 - 1) / F;
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        for (int i = threadIdx.x; i < 1000; i += 256) {                           // this wave's 250 halo voxels: four passes of 64 lanes
            const int hz = i % 10, hy = (i / 10) % 10, hx = i / 100;
            const int x = x0 - 1 + hx, y = y0 - 1 + hy, z = z0 - 1 + hz + (r & 1);
            int mx, my, mz;
            float ax, bx, ay, by, az, bz;
            up_axis<F>(x, Di, mx, ax, bx); up_axis<F>(y, Di, my, ay, by); up_axis<F>(z, Di, mz, az, bz);
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const int dx = o >> 2, dy = (o >> 1) & 1, dz = o & 1;
                const float w = (dx ? bx : ax) * (dy ? by : ay) * (dz ? bz : az);
                if (w != 0.f) {
                    const float *q = src + ((((mx - mx0 + dx) * NS + (my - my0 + dy)) * NS + (mz - mz0 + dz)) * 8);
                    const float4 q0 = *reinterpret_cast<const float4 *>(q), q1 = *reinterpret_cast<const float4 *>(q + 4);
                    acc[0] += w * q0.x; acc[1] += w * q0.y; acc[2] += w * q0.z; acc[3] += w * q0.w;
                    acc[4] += w * q1.x; acc[5] += w * q1.y; acc[6] += w * q1.z; acc[7] += w * q1.w;
                }
            }
            // hi plane at halo[16 i], code slots at halo[16384 + 16 i]: sn_store8<2> with the code plane 8192 halfs behind the hi plane
            sn_store8<2>(reinterpret_cast<_Float16 *>(halo + 16 * i), 8192, acc, e8);
        }
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) atomicAdd(clk + (threadIdx.x >> 6), (unsigned long long)(t1 - t0));
    if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = (float)halo[17] + (float)halo[16384 + 33];
}

template <int F>
static void run(const char *name, const float *seed, unsigned long long *clk, float *sink)
{
    const int reps = 2000, cus = 256;
    hipMemset(clk, 0, 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int w = 0; w < 2; ++w) {
        hipMemset(clk, 0, 4 * 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(loader_kernel<F>, dim3(cus), dim3(256), 0, 0, seed, clk, sink, reps, 127 - 2);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h[4];
    hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost);
    double per = 0;
    for (int w = 0; w < 4; ++w) per += (double)h[w] / cus / reps / 4.0;
    printf("%-28s %8.0f shader clocks per slab (250 voxels x 8 channels per wave, one wave per SIMD, all %d CUs busy)  = %.2f us per slab at the launch's %.2f GHz\n",
           name, per, cus, ms * 1e3 / reps, per / (ms * 1e3 / reps) / 1e3);
}

int main()
{
    float *seed, *sink; unsigned long long *clk;
    hipMalloc(&seed, 4096 * 4); hipMalloc(&sink, 64); hipMalloc(&clk, 64);
    float *h = new float[4096];
    unsigned st = 777;
    for (int i = 0; i < 4096; ++i) { st = st * 1664525u + 1013904223u; h[i] = (float)(st >> 8) / 16777216.0f; }      // sigmoid-like values in (0, 1)
    hipMemcpy(seed, h, 4096 * 4, hipMemcpyHostToDevice);
    run<2>("x2 map (side_op2, 2 groups)", seed, clk, sink);
    run<4>("x4 maps (side_op3 / 4, 4 groups)", seed, clk, sink);
    printf("reference: a slab of merge_conv_a's real loop = 3.375 weight pieces x ~3,450 clocks = ~11,600 clocks; the tile = 8 slabs + a ~15,000-clock store epilogue\n");
    return 0;
}
