// tools/probe/dma_probe.hip - what does ISSUING an LDS-DMA instruction (global_load_lds_dwordx4 / buffer_load_dwordx4 ... lds) cost the
// issuing wave, alone and next to waves that run MFMA bursts or LDS reads? (the conv kernels' load segments are bounded by it)
// 256 workgroups x 8 waves; waves 0-3 (one per SIMD) issue N DMAs of 1 KiB from an L2-resident region per round and time the issue
// (first -> behind the last) and the drain (s_waitcnt vmcnt(0)); waves 4-7 idle / run MFMA chains / run ds_read_b128 chains.
//   hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, int KIND>      // KIND 0: global_load_lds, 1: buffer_load lds
__global__ __launch_bounds__(512) void probe(const char *src, int rounds, int partner, unsigned long long *clk, float *sink)
{
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 4) {
        long long t_issue = 0, t_drain = 0;
        const char *g = src + (size_t)(blockIdx.x & 15) * 65536 + lane * 16;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(src + (size_t)(blockIdx.x & 15) * 65536), 0, 65536, 0x00020000);
        for (int r = 0; r < rounds; ++r) {
            __builtin_amdgcn_s_barrier();
            const long long t0 = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int slot = (i * 4 + wave + r) & 31;
                char *dst = lds + slot * 1024;
                if constexpr (KIND == 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + slot * 1024),
                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)dst, 16, slot * 1024 + lane * 16, 0, 0, 0);
            }
            const long long t1 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t2 = __builtin_readcyclecounter();
            t_issue += t1 - t0; t_drain += t2 - t1;
            __builtin_amdgcn_s_barrier();
        }
        if (lane == 0) { atomicAdd(clk, (unsigned long long)t_issue); atomicAdd(clk + 1, (unsigned long long)t_drain); }
    } else {
        f32x4 acc[4] = {};
        half8 a = {}, b = {};
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 0.01f + j); b[j] = (_Float16)(1.f - j * 0.1f); }
        float s = 0.f;
        for (int r = 0; r < rounds; ++r) {
            __builtin_amdgcn_s_barrier();
            if (partner == 1) {
#pragma unroll
                for (int k = 0; k < 28; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k & 3], 0, 0, 0);
            } else if (partner == 2) {
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    const half8 v = *reinterpret_cast<const half8 *>(lds + 32768 + ((k * 64 + lane) & 1023) * 16);
                    s += (float)v[0];
                }
            }
            __builtin_amdgcn_s_barrier();
        }
        if (sink && lane == 0 && wave == 4) sink[blockIdx.x] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] + s;
    }
}

// loaders: K ds_read_b128 (distinct addresses, conflict-free: lane-linear 16 B) then s_waitcnt lgkmcnt(0); partners as above
template <int K>
__global__ __launch_bounds__(512) void probe_lds(int rounds, int partner, unsigned long long *clk, float *sink)
{
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float *>(lds)[i] = (float)i;
    __syncthreads();
    if (wave < 4) {
        long long t_all = 0;
        float s = 0.f;
        for (int r = 0; r < rounds; ++r) {
            __builtin_amdgcn_s_barrier();
            const long long t0 = __builtin_readcyclecounter();
            half8 v[K];
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = *reinterpret_cast<const half8 *>(lds + ((i * 4 + wave + r) & 31) * 1024 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < K; ++i) asm volatile("" ::"v"(v[i]));
            const long long t1 = __builtin_readcyclecounter();
            t_all += t1 - t0;
#pragma unroll
            for (int i = 0; i < K; ++i) s += (float)v[i][0];
            __builtin_amdgcn_s_barrier();
        }
        if (lane == 0) { atomicAdd(clk, (unsigned long long)t_all); if (sink && s == 12345.f) sink[0] = s; }
    } else {
        f32x4 acc[4] = {};
        half8 a = {}, b = {};
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 0.01f + j); b[j] = (_Float16)(1.f - j * 0.1f); }
        for (int r = 0; r < rounds; ++r) {
            __builtin_amdgcn_s_barrier();
            if (partner == 1) {
#pragma unroll
                for (int k = 0; k < 28; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k & 3], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
        }
        if (sink && lane == 0 && wave == 4) sink[blockIdx.x] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
    }
}
// loaders run a conv-kernel-like load segment (address VALU, K LDS reads, D LDS-DMAs with their M0 set-up, scalar bookkeeping) while
// their SIMD partners run a burst of NM MFMAs: how long does the SEGMENT take, and how long the whole slot (barrier to barrier)?
template <int K, int D, int NM>
__global__ __launch_bounds__(512) void probe_seg(const char *src, int rounds, unsigned long long *clk, float *sink, int salt)
{
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float *>(lds)[i] = (float)i;
    __syncthreads();
    long long t_seg = 0, t_slot = 0;
    f32x4 acc[28] = {};
    half8 a = {}, b = {};
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 0.01f + j); b[j] = (_Float16)(1.f - j * 0.1f); }
    float s = 0.f;
    const char *g = src + (size_t)(blockIdx.x & 15) * 65536 + lane * 16;
    // roles alternate every slot, as in the kernel: group 0 = waves 0-3, group 1 = waves 4-7
    const int grp = wave >> 2;
    for (int r = 0; r < rounds; ++r) {
        const long long t0 = __builtin_readcyclecounter();
        if (((r ^ grp) & 1) == 0) {
            // ---- load segment
            int base = ((r * 7 + salt) & 15) * 1024;
            asm volatile("" : "+s"(base));
            half8 v[K > 0 ? K : 1];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                int ad = base + ((i * 4 + (wave & 3)) & 31) * 1024 + lane * 16;
                asm volatile("" : "+v"(ad));                 // (one VALU add per read, like the tap-offset + fragment-base sums)
                v[i] = *reinterpret_cast<const half8 *>(lds + (ad & 32767));
            }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int slot = 32 + ((i * 4 + (wave & 3) + r) & 15);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + slot * 1024),
                                                 (__attribute__((address_space(3))) void *)(lds + slot * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < K; ++i) asm volatile("" ::"v"(v[i]));
            const long long t1 = __builtin_readcyclecounter();
            t_seg += t1 - t0;
        } else {
#pragma unroll
            for (int k = 0; k < NM; ++k) acc[k % 28] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k % 28], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();
        const long long t2 = __builtin_readcyclecounter();
        if (((r ^ grp) & 1) == 0) t_slot += t2 - t0;
        if ((r & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) { atomicAdd(clk, (unsigned long long)t_seg); atomicAdd(clk + 1, (unsigned long long)t_slot); }
    float t = s;
    for (int k = 0; k < 28; ++k) t += acc[k][0];
    if (sink && t == 12345.f) sink[0] = t;
}
template <int K, int D, int NM>
static void run_seg(const char *src, unsigned long long *clk, float *sink)
{
    const int rounds = 4000;
    unsigned long long h[2];
    hipMemset(clk, 0, 16);
    hipLaunchKernelGGL((probe_seg<K, D, NM>), dim3(256), dim3(512), 65536, 0, src, rounds, clk, sink, 3);
    hipDeviceSynchronize();
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double per = 256.0 * 8 * rounds / 2;
    printf("segment: %2d reads + %d DMAs next to %2d MFMAs: load %4.0f clk, slot (barrier to barrier) %4.0f clk  [MFMA issue alone = %d]\n", K, D, NM, h[0] / per, h[1] / per, NM * 16);
}

template <int K>
static void run_lds(unsigned long long *clk, float *sink, int partner)
{
    const int rounds = 2000;
    unsigned long long h[2];
    hipMemset(clk, 0, 16);
    hipLaunchKernelGGL((probe_lds<K>), dim3(256), dim3(512), 65536, 0, rounds, partner, clk, sink);
    hipDeviceSynchronize();
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double per = 256.0 * 4 * rounds;
    printf("ds_read_b128 K=%-2d partner=%-5s: %.0f clk until all landed (%.0f per read)\n", K, partner == 0 ? "idle" : "mfma", h[0] / per, h[0] / per / K);
}

template <int N, int KIND>
static void run(const char *src, unsigned long long *clk, float *sink, int partner)
{
    const int rounds = 2000;
    unsigned long long h[2];
    hipMemset(clk, 0, 16);
    hipLaunchKernelGGL((probe<N, KIND>), dim3(256), dim3(512), 65536, 0, src, rounds, partner, clk, sink);
    hipDeviceSynchronize();
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double per = 256.0 * 4 * rounds;
    printf("%-11s N=%d partner=%-5s: issue %.0f clk (%.0f per instruction), drain %.0f clk\n", KIND ? "buffer_load" : "global_load", N,
           partner == 0 ? "idle" : (partner == 1 ? "mfma" : "lds"), h[0] / per, h[0] / per / N, h[1] / per);
}

int main()
{
    char *src; unsigned long long *clk; float *sink;
    hipMalloc(&src, 16 * 65536); hipMemset(src, 1, 16 * 65536);
    hipMalloc(&clk, 16); hipMalloc(&sink, 256 * 4);
    for (int partner = 0; partner < 3; ++partner) {
        run<1, 0>(src, clk, sink, partner); run<2, 0>(src, clk, sink, partner); run<4, 0>(src, clk, sink, partner);
        run<1, 1>(src, clk, sink, partner); run<2, 1>(src, clk, sink, partner); run<4, 1>(src, clk, sink, partner);
    }
    for (int partner = 0; partner < 2; ++partner) { run_lds<1>(clk, sink, partner); run_lds<4>(clk, sink, partner); run_lds<8>(clk, sink, partner); run_lds<11>(clk, sink, partner); run_lds<22>(clk, sink, partner); }
    run_seg<0, 0, 28>(src, clk, sink); run_seg<0, 0, 56>(src, clk, sink);
    run_seg<4, 0, 28>(src, clk, sink); run_seg<4, 3, 28>(src, clk, sink); run_seg<11, 1, 28>(src, clk, sink); run_seg<11, 1, 56>(src, clk, sink); run_seg<11, 1, 14>(src, clk, sink);
    run_seg<22, 1, 28>(src, clk, sink); run_seg<4, 3, 0>(src, clk, sink); run_seg<11, 1, 0>(src, clk, sink);
    return 0;
}
