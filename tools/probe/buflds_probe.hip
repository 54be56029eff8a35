// buflds_probe.hip — what does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) write for lanes whose
// offset fails the descriptor's range check? (conv3d_mfma.h relies on: zeros.)  Build: hipcc --offload-arch=gfx950 -O2 -o buflds_probe buflds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char *in, int nbytes, unsigned *out, int soff)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[256 * 3];
    for (int i = threadIdx.x; i < 768; i += 64) lds[i] = 0xCDCDCDCDu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)in, (short)0, nbytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    // segment 0: lanes 0-31 in range, 32-47 offset = nbytes (first byte out of range), 48-63 huge / negative offsets
    unsigned off0 = lane < 32 ? lane * 16 : (lane < 48 ? (unsigned)nbytes + (lane - 32) * 16 : (lane < 56 ? 0x80000000u + lane * 16 : (unsigned)(-16 * (lane - 55))));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, off0, 0, 0, 0);
    // segment 1: voffset in range, soffset pushes the address past the end: is soffset range-checked?
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(lds + 256), 16, lane * 16, soff, 0, 0);
    // segment 2: partially out of range: offset = nbytes - 8 (only 8 of the 16 bytes are inside)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(lds + 512), 16, (unsigned)nbytes - 8, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += 64) out[i] = lds[i];
}
int main()
{
    const int n = 1024;      // descriptor covers the first 1024 bytes of a 4096-byte allocation filled with 0xAB
    char *d; unsigned *o;
    hipMalloc(&d, 4096); hipMemset(d, 0xAB, 4096); hipMalloc(&o, 768 * 4);
    k<<<1, 64>>>(d, n, o, 2048);
    std::vector<unsigned> h(768);
    hipMemcpy(h.data(), o, 768 * 4, hipMemcpyDeviceToHost);
    auto show = [&](const char *t, int seg, int lane) { printf("%s lane %2d: %08x %08x %08x %08x\n", t, lane, h[seg * 256 + lane * 4], h[seg * 256 + lane * 4 + 1], h[seg * 256 + lane * 4 + 2], h[seg * 256 + lane * 4 + 3]); };
    show("in range   ", 0, 0); show("in range   ", 0, 31); show("off=nbytes ", 0, 32); show("off>nbytes ", 0, 47); show("off=2^31+  ", 0, 48); show("off=-16    ", 0, 56); show("off=-128   ", 0, 63);
    show("soff past  ", 1, 0); show("soff past  ", 1, 63);
    show("straddling ", 2, 0);
    return 0;
}
