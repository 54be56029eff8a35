// tools/probe/store_probe.hip - what does the store epilogue of a persistent conv kernel cost as a function of the plane stride?
// Mimics merge_conv_a's EPI_STORE address pattern: 256 workgroups x 8 waves walk 32 tiles of 8x8x8 voxels of a [B][G][32][32][32][8]
// fp16 tensor (+ a second plane set), each wave issuing 28 16-byte-per-lane stores per tile (2 group planes x 2 fragment rows x 8 z per
// instruction), with an idle gap (s_sleep) between the bursts standing in for the K loop. Prints shader clocks per burst (issue of the
// first store -> issue of the last + the wait until all have been acknowledged) for several plane strides.
//   hipcc --offload-arch=gfx950 -O3 -o store_probe store_probe.hip && ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void burst(char *out, size_t plane_stride, size_t samp_stride, size_t lo_off, int tiles, int gap, int nf,
                                             unsigned long long *clk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, v = lane & 15, kq = lane >> 4;
    long long t_issue = 0, t_ack = 0;
    for (int t = 0; t < tiles; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int b = tile >> 6, tx = (tile >> 4) & 3, ty = (tile >> 2) & 3, tz = tile & 3;
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(127);
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        for (int mp = 0; mp < 4; mp += 2) {
            const int m = mp + (kq & 1);                                   // fragment m of the wave: rows 2m, 2m+1 of x-slice `wave`
            const int x = tx * 8 + wave, y = ty * 8 + 2 * m + (v >> 3), z = tz * 8 + (v & 7);
            const size_t voff = ((size_t)((x * 32 + y) * 32 + z) + (size_t)(kq >> 1) * (plane_stride / 16)) * 16;
            for (int n = 0; n < nf; ++n) {
                char *o = out + (size_t)b * samp_stride + (size_t)(2 * n) * plane_stride + voff;
                const u32x4 d = {(unsigned)tile, (unsigned)n, (unsigned)lane, 0u};
                *reinterpret_cast<u32x4 *>(o) = d;
                *reinterpret_cast<u32x4 *>(o + lo_off) = d;
            }
        }
        const long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_readcyclecounter();
        t_issue += t1 - t0; t_ack += t2 - t1;
    }
    if (lane == 0) { atomicAdd(clk, (unsigned long long)t_issue); atomicAdd(clk + 1, (unsigned long long)t_ack); }
}

int main()
{
    const int B = 128, G = 14, nf = 7, tiles = 32;
    const size_t vol = 32768 * 16;                                  // bytes of one 8-channel group plane
    unsigned long long *clk;
    hipMalloc(&clk, 16);
    const size_t pads[] = {0, 256, 1024, 4096, 4096 + 256, 8192 + 512, 65536 + 4096 + 256};
    for (int gap : {0, 8}) {
        for (size_t pad : pads) {
            const size_t ps = vol + pad, ss = ps * G, lo = ss * B;
            char *buf;
            if (hipMalloc(&buf, 2 * lo) != hipSuccess) { printf("alloc failed\n"); return 1; }
            hipMemset(buf, 0, 2 * lo);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f;
            unsigned long long h[2] = {0, 0};
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(clk, 0, 16);
                hipEventRecord(e0);
                hipLaunchKernelGGL(burst, dim3(256), dim3(512), 0, 0, buf, ps, ss, lo, tiles, gap, nf, clk);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
            }
            const double per = 256.0 * 8 * tiles;
            printf("gap %d  plane pad %6zu B: kernel %.3f ms  issue %.0f clk/burst  ack-wait %.0f clk/burst  (%.1f GB written)\n", gap, pad, best,
                   h[0] / per, h[1] / per, 2.0 * nf * 2 * 64 * 16 * 8 * 256.0 * tiles / 1e9);
            hipFree(buf);
        }
    }
    return 0;
}
