// Probe: lane mapping of v_permlane16_swap_b32 on gfx950 (used by the widened store epilogue of conv3d_mfma.h).
// hipcc --offload-arch=gfx950 -O2 tools/probe/swap_probe.hip -o /tmp/swap_probe && /tmp/swap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main()
{
    unsigned *d, h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("r0:"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]); printf("\nr1:"); for (int i = 0; i < 64; ++i) printf(" %u", h[64 + i]); printf("\n");
    return 0;
}
