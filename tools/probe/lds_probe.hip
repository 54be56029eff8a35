// lds_probe.hip — how many clocks does one wave's LDS read take for the address patterns conv3d_mfma.h uses? (bank conflicts show up as
// extra clocks per instruction.) One wave per workgroup, N back-to-back independent reads of the same pattern, s_memtime around them.
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_probe lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v3i __attribute__((ext_vector_type(3)));

constexpr int N = 64;

template <int KIND>
__global__ void k(const unsigned *addr, long long *out, int *sink, int nwaves)
{
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<int *>(lds)[i] = i;
    __syncthreads();
    const unsigned a = addr[threadIdx.x & 63] + (threadIdx.x >> 6) * 4096;     // every wave its own 4 KB-shifted copy of the pattern
    int acc = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if constexpr (KIND == 128) { v4i d; asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a)); asm volatile("" ::"v"(d)); }
        if constexpr (KIND == 96) { v3i d; asm volatile("ds_read_b96 %0, %1" : "=v"(d) : "v"(a)); asm volatile("" ::"v"(d)); }
        if constexpr (KIND == 64) { long long d; asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(a)); asm volatile("" ::"v"(d)); }
        if constexpr (KIND == 32) { int d; asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(a)); asm volatile("" ::"v"(d)); }
        if ((i & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = acc + lds[a & 65535];
}

template <int KIND>
static double run(const std::vector<unsigned> &h, int nw = 1)
{
    unsigned *d; long long *o; int *s;
    hipMalloc(&d, 256); hipMalloc(&o, 8 * 16); hipMalloc(&s, 256 * 16);
    hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
    long long best = 1LL << 60;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64 * nw), 0, 0, d, o, s, nw);
        long long t[16]; hipMemcpy(t, o, 8 * nw, hipMemcpyDeviceToHost);
        long long mx = 0; for (int w = 0; w < nw; ++w) mx = t[w] > mx ? t[w] : mx;
        if (mx < best) best = mx;
    }
    hipFree(d); hipFree(o); hipFree(s);
    return (double)best / N;
}

int main()
{
    auto pat = [](auto f) { std::vector<unsigned> h(64); for (int l = 0; l < 64; ++l) h[l] = f(l); return h; };
    // weight fragment: lane * 16 contiguous
    printf("b128  lane*16 (weight fragment)                      : %.1f clk/read\n", run<128>(pat([](int l) { return (unsigned)l * 16; })));
    // activation fragment, VS = 16 B/voxel, HZ = 10: lanes (v = l&15: row v>>3, z = v&7; kq = l>>4: another tap, here +16*kq voxels)
    for (int gap : {1, 2, 4}) {
        auto h = pat([gap](int l) { int v = l & 15, kq = l >> 4; return (unsigned)(((v >> 3) * gap * 10 + (v & 7)) * 16 + kq * 1600); });
        printf("b128  X fragment VS=16, rows %d apart                  : %.1f clk/read\n", gap, run<128>(h));
    }
    for (int gap : {1, 2, 4}) {
        auto h = pat([gap](int l) { int v = l & 15, kq = l >> 4; return (unsigned)(((v >> 3) * gap * 10 + (v & 7)) * 32 + kq * 3200); });
        printf("b128  X fragment VS=32, rows %d apart                  : %.1f clk/read\n", gap, run<128>(h));
    }
    // the four tap quarters at unrelated offsets (one z step, one y row, one x plane)
    {
        auto h = pat([](int l) { int v = l & 15, kq = l >> 4; const int off[4] = {0, 16, 160, 1600}; return (unsigned)(((v >> 3) * 4 * 10 + (v & 7)) * 16 + off[kq]); });
        printf("b128  X fragment VS=16, gap 4, taps (0, +z, +y, +x)      : %.1f clk/read\n", run<128>(h));
    }
    {
        auto h = pat([](int l) { int v = l & 15, kq = l >> 4; return (unsigned)(((v >> 3) * 4 * 10 + (v & 7)) * 16 + kq * 1600); });
        printf("b96   MX slot VS=16, gap 4                              : %.1f clk/read\n", run<96>(h));
    }
    printf("b96   lane*16                                           : %.1f clk/read\n", run<96>(pat([](int l) { return (unsigned)l * 16; })));
    printf("b64   lane*16 (MX weight dwords 4..5, scales)           : %.1f clk/read\n", run<64>(pat([](int l) { return (unsigned)l * 16; })));
    printf("b64   lane*8                                            : %.1f clk/read\n", run<64>(pat([](int l) { return (unsigned)l * 8; })));
    printf("b32   lane*4                                            : %.1f clk/read\n", run<32>(pat([](int l) { return (unsigned)l * 4; })));
    printf("b32   lane*16                                           : %.1f clk/read\n", run<32>(pat([](int l) { return (unsigned)l * 16; })));
    printf("b128  all lanes the same address (broadcast)            : %.1f clk/read\n", run<128>(pat([](int) { return 0u; })));
    // throughput of the LDS itself: the same patterns issued by 1 / 2 / 4 / 8 waves of one workgroup at once (clocks per read as one wave sees them)
    for (int nw : {1, 2, 4, 8}) {
        auto w16 = pat([](int l) { return (unsigned)l * 16; });
        auto xg4 = pat([](int l) { int v = l & 15, kq = l >> 4; return (unsigned)(((v >> 3) * 4 * 10 + (v & 7)) * 16 + kq * 800); });
        printf("%d waves: b128 lane*16 %.1f | b128 X gap4 %.1f | b96 lane*16 %.1f | b64 lane*16 %.1f | b64 lane*8 %.1f | b32 lane*4 %.1f clk/read\n", nw,
               run<128>(w16, nw), run<128>(xg4, nw), run<96>(w16, nw), run<64>(w16, nw), run<64>(pat([](int l) { return (unsigned)l * 8; }), nw),
               run<32>(pat([](int l) { return (unsigned)l * 4; }), nw));
    }
    // bank structure, 8 waves: a 16-lane group = two 128-byte rows, the second one d bytes behind the first; the four lane quarters q * dq apart
    for (int d : {128, 192, 256, 320, 384, 512, 640, 768, 1024, 1152, 1280})
        printf("8 waves: rows %4d B apart, quarters 2048 B apart : %.1f clk/read\n", d,
               run<128>(pat([d](int l) { int v = l & 15, kq = l >> 4; return (unsigned)((v >> 3) * d + (v & 7) * 16 + kq * 2048); }), 8));
    for (int dq : {0, 16, 32, 64, 128, 160, 256, 512, 800, 1024, 1600, 2048})
        printf("8 waves: rows 128 B apart (256 B contiguous), quarters %4d B apart : %.1f clk/read\n", dq,
               run<128>(pat([dq](int l) { int v = l & 15, kq = l >> 4; return (unsigned)(v * 16 + kq * dq); }), 8));
    for (int dq : {16, 160, 1600})
        printf("8 waves: rows 640 B apart, quarters %4d B apart : %.1f clk/read\n", dq,
               run<128>(pat([dq](int l) { int v = l & 15, kq = l >> 4; return (unsigned)((v >> 3) * 640 + (v & 7) * 16 + kq * dq); }), 8));
    // which lane quarters share an LDS pass? quarter k alone is moved by 160 B (a y step of the 10-voxel halo row), the others read the same addresses
    for (int k = 1; k < 4; ++k)
        for (int dq : {128, 160, 1600})
            printf("8 waves: rows 640 B apart, only quarter %d moved by %4d B : %.1f clk/read\n", k, dq,
                   run<128>(pat([k, dq](int l) { int v = l & 15, kq = l >> 4; return (unsigned)((v >> 3) * 640 + (v & 7) * 16 + (kq == k ? dq : 0)); }), 8));
    for (int k = 1; k < 4; ++k)
        printf("8 waves: rows 640 B apart, quarters 0 and %d at 0, the other two at 160 / 320 B : %.1f clk/read\n", k,
               run<128>(pat([k](int l) { int v = l & 15, kq = l >> 4; int o = (kq == 0 || kq == k) ? 0 : 160; return (unsigned)((v >> 3) * 640 + (v & 7) * 16 + o); }), 8));
    // the kernels' own activation-fragment reads: every K-chunk of a channel slab (4 consecutive K groups = the 4 lane quarters), fragment 0,
    // 8 waves; group g = tap * CS8 + channel group (conv3d_mfma.h write_koff), tap = ((dx * 3) + dy) * 3 + dz (2-D: dy * 3 + dz)
    struct Cfg { const char *name; int vs, cs8, hy, hz, dil, ntap, gap; };
    const Cfg cfgs[] = {
        {"merge / conv1 (16 B, 1 group, rows 4 apart)", 16, 1, 10, 10, 1, 27, 4}, {"merge / conv1, rows 1 apart (EPI_SIDEPOOL map)", 16, 1, 10, 10, 1, 27, 1},
        {"conv2/3 f16x3 (32 B, 2 groups, rows 4 apart)", 32, 2, 10, 10, 1, 27, 4}, {"conv2/3, rows 2 apart", 32, 2, 10, 10, 1, 27, 2}, {"conv2/3, rows 1 apart", 32, 2, 10, 10, 1, 27, 1},
        {"conv4 (16 B, dilation 2, 12-voxel rows, rows 4 apart)", 16, 1, 12, 12, 2, 27, 4}, {"conv4, rows 2 apart", 16, 1, 12, 12, 2, 27, 2}, {"conv4, rows 1 apart", 16, 1, 12, 12, 2, 27, 1},
        {"2-D f16x3 (32 B, 2 groups, rows 2 apart)", 32, 2, 10, 10, 1, 9, 2}, {"2-D f16x3, rows 1 apart", 32, 2, 10, 10, 1, 9, 1},
    };
    for (const Cfg &c : cfgs) {
        const int G = c.ntap * c.cs8, nch = (G + 3) / 4;
        double sum = 0, worst = 0;
        for (int ch = 0; ch < nch; ++ch) {
            auto h = pat([&](int l) {
                const int v = l & 15, kq = l >> 4;
                int g = 4 * ch + kq; if (g >= G) g = G - 1;
                const int tap = g / c.cs8, cg = g % c.cs8;
                const int dz = tap % 3, dy = (tap / 3) % 3, dx = c.ntap == 27 ? tap / 9 : 0;
                const int off = (((dx * c.dil) * c.hy + dy * c.dil) * c.hz + dz * c.dil) * c.vs + cg * 16;
                return (unsigned)((((v >> 3) * c.gap) * c.hz + (v & 7)) * c.vs + off);
            });
            const double t = run<128>(h, 8);
            sum += t; worst = t > worst ? t : worst;
        }
        printf("8 waves, %-56s: mean %.1f  worst chunk %.1f clk/read over %d chunks\n", c.name, sum / nch, worst, nch);
    }
    // which tap pairs may share an LDS pass (16-byte voxels, rows 4 apart)? quarters (0,1) and (2,3) each d bytes apart
    for (int d : {16, 32, 144, 160, 176, 320, 1280, 1440, 1600, 1616, 1760, 1920, 2880, 3040, 3200, 3360, 3520})
        printf("8 waves: tap pair %4d B apart : %.1f clk/read\n", d,
               run<128>(pat([d](int l) { int v = l & 15, kq = l >> 4; return (unsigned)((v >> 3) * 640 + (v & 7) * 16 + (kq & 1) * d + (kq >> 1) * 16384); }), 8));
    // PMAP (EPI_SIDEPOOL) fragment: rows adjacent
    {
        auto h = pat([](int l) { int v = l & 15, kq = l >> 4; return (unsigned)(((v >> 3) * 10 + (v & 7)) * 16 + kq * 1600); });
        printf("b128  X fragment VS=16, adjacent rows (EPI_SIDEPOOL map) : %.1f clk/read\n", run<128>(h));
    }
    return 0;
}
