// conv3d_mfma.h — implicit-GEMM 3-D convolution on the CDNA4 matrix cores (gfx950 only).
//
// Replaces, for the MI355X path, what the reference obtains from cuDNN through Lasagne:
//   Conv3DDNNLayer 3x3x3 / 1x1x1 'same' cross-correlation   (nets/SurfaceNet.py:33-74)
//   DilatedConv3DLayer via GpuDnnConv3dGradW, dilation 2     (nets/layers.py:200-253)
//   batch_norm(...) at deterministic=True folded to y = act(conv*scale + shift)
//   merge_conv3 (1x1x1, 100 -> 1, BN, sigmoid) fused as a second epilogue (nets/SurfaceNet.py:74)
//
// Data layout (ours, not the reference's NCDHW): activations are fp16 in 8-channel groups,
//   act[b][c/8][x][y][z][c%8] (c padded to a multiple of 8): one voxel's group is one 16-byte vector = exactly one
//   lane's share of a v_mfma_f32_16x16x32_f16 B operand, and one channel slab of a halo tile is a set of contiguous
//   z-rows in HBM (the LDS-DMA that stages it reads whole cache lines instead of 16-byte slivers).
// Precision modes (template SPLIT):
//   SPLIT=0  "f16":   operands rounded to fp16, fp32 accumulate. L_inf vs the fp64 oracle ~2e-3 on
//                     BN-calibrated nets -> does NOT meet the 1e-3 parity bar; offered as the fast mode.
//   SPLIT=1  "f16x3": every operand is an unevaluated sum hi+lo of two fp16 numbers (22 significant
//                     bits); w*x = wh*xh + wl*xh + wh*xl on three MFMAs (the dropped wl*xl term is
//                     2^-22 relative), fp32 accumulate -> fp32-class results at 1/3 of the f16 MFMA rate,
//                     still 5.3x the f32-input MFMA rate of gfx950 (no xf32/TF32 on this chip).
//                     Activations live in HBM as two fp16 planes (hi, lo).
//   SPLIT=2  "f16m8": the main term wh*xh on the f16 MFMA; the two correction terms wl*xh + wh*xl (2^-11 of it) on ONE
//                     v_mfma_scale_f32_16x16x128_f8f6f4 per 64 k on 6-bit operands (fp6 e2m3, mx_format.h: lo parts pre-scaled by 2^11, a static
//                     per-tensor premultiplier on the activation side, one E8M0 scale per 32-element weight block), which issues at twice the fp8 rate:
//                     1.5 MFMA units per product instead of 3 (the two merge layers of the default mode; every layer of the all-MX mode). Second
//                     activation plane = 16-byte slots per voxel and group holding 16 six-bit codes [hi c0..3 | lo c0..3 | hi c4..7 | lo c4..7].
//   SPLIT=3  "f16m8e" (round 5; the dilated layers conv4_x of the default mode): the same with fp8 e4m3 codes - 2 MFMA units per product, and the EXPONENT RANGE
//                     the 6-bit codes lack: a static 6-bit premultiplier cannot hold the data-dependent outliers of the conv4 chain (profiles/r5/README.md).
//                     Second activation plane = [fp8(hi * 2^s) x8 | fp8(lo * 2^12 * 2^s) x8] per group; weights as plain fp8 codes (no block scales).
// GEMM view: D[cout][voxel] += W[cout][k] * X[k][voxel], k = (tap, cin) in 8-channel groups.
//   A operand = weights, pre-packed on the host in fragment order (lane l: cout = l&15, k = (l>>4)*8 + j);
//   B operand = activations of a (TX+2R)x(TY+2R)x(TZ+2R) halo tile, staged per channel slab in LDS and re-read
//               for every one of the 27 taps through a per-slab tap-offset table;
//   D         = lane l, reg r: voxel = l&15, cout = (l>>4)*4 + r  -> 4 consecutive channels per lane = half of a voxel's 16-byte group; the store
//               epilogues finish two voxel fragments together and exchange the halves (v_permlane16_swap): ONE 16-byte store per lane and plane.
// Execution structure (v2):
//   * PERSISTENT workgroups: grid = (#CUs x workgroups/CU, cout splits); each workgroup walks tiles
//     blockIdx.x, +gridDim.x, ... so nothing is ever staged behind a cold start except its very first slab.
//   * Everything that enters LDS arrives by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip): the weight
//     stream in pieces of PCH K-chunks (double-buffered, continuous across slabs and tiles) and the halo tile
//     of the NEXT (tile, slab) into the second halo buffer while the current one is being multiplied. Voxels
//     outside the volume ('same' padding / PadLayer) are fetched from a zero page.
//   * One barrier per weight piece; the K loop inside a piece is software-pipelined by hand (inline-asm
//     ds_read_b128 with counted s_waitcnt: hipcc would sink every read to its first use and drain lgkmcnt(0)).
//   * Workgroup = NW waves = (NW*XS) x 8 x 8 output voxels x (NF*16) output channels; wave w owns x-slices
//     [w*XS, (w+1)*XS) and all NF channel fragments, so the 1x1x1 reduction of merge_conv3 stays inside a wave.
#pragma once
#include <hip/hip_runtime.h>
#include "mx_format.h"
#include <stdint.h>
#include <type_traits>

// Fixed choices whose alternatives were measured and lost (A/B logs: profiles/r2 .. r4/README.md; the switches themselves were removed in round 5):
// XCD-aware tile walk on; a fragment's two z-rows lie 4 y-rows apart in the halo tile of the 3x3x3 layers and of the 2-D nets (conflict-free LDS
// passes: bank conflicts 0.21-0.40 -> 0.04-0.17, lds_probe), 2 apart under dilation 2, adjacent for the 1x1x1 side convolutions; the MX step's lane
// covers BOTH correction terms of 2 channel groups (two 16-byte slot reads); ping-pong loops: branch-free weight-DMA issue, per-tile resync of the
// two wave groups, the next slab's halo DMAs in the slab's first load slot, one K-chunk per segment.
constexpr int kRowGap3x3 = 4, kRowGap2D = 4, kRowGapDil2 = 2;
#ifndef SN_TIMING
#define SN_TIMING 0       // diagnostic build (tools/wave_timing.py): per-wave shader-clock totals added into a.status[2..] (results stay valid). One-wave-per-SIMD
                          // loop: 1 {burst A, burst B}, 2 {vmcnt wait, barrier}, 3 {burst M, whole piece} per piece, 4 per slab {slab head, piece loop}; 10: per
                          // TILE {epilogue, K loop} (every loop)
#endif
// one-wave-per-SIMD loop: sched_barrier(0) behind every (MFMA, filler) pair pins the written order
#define PW_SB __builtin_amdgcn_sched_barrier(0)

namespace sn {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxSlab = 128;

struct ConvArgs {
    const _Float16 *in;   // [B][in_cs/8][D][D][D][8]  (hi plane; lo plane at +in_lo_off elements when SPLIT)
    _Float16 *out;        // [B][out_cs/8][D][D][D][8], channels [out_coff, out_coff+out_cp)
    float *out_f32;       // EPI_FINAL: [B][D][D][D]
    const _Float16 *wpack;
    const float *scale;   // [nsplit*NF*16]
    const float *shift;
    const float *w3;      // EPI_FINAL: [NF*16] fp32 weights of the fused 1x1x1 conv
    const void *zero_page;    // >= 16 zero bytes: source of out-of-volume / padding voxels (generic halo path of the 2-D nets)
    // EPI_SIDEPOOL: packed A fragments of the 1x1x1 side conv [(NF+1)/2 K-chunks][hi | lo][64 lanes][8 halfs], its folded BN, and the two
    // destinations: side output -> channels [side_coff, side_coff+16) of a [B][side_cs/8][D]^3[8] tensor (format OSPLIT),
    // pooled output -> [B][pool_cs/8][D/2]^3[8] (format SPLIT)
    const _Float16 *side_w;
    const float *side_scale, *side_shift;
    _Float16 *side_out, *pool_out;
    long long side_lo_off, pool_lo_off;
    int side_cs, side_coff, pool_cs, side_act;
    unsigned *status;         // numeric status word of the context (or nullptr): status_bit is OR-ed in when an output value is not
    unsigned status_bit;      // finite or exceeds the fp16 range its hi plane is stored in (|y| > 65504)
    unsigned mx_sat_bits;     // != 0: the output tensor carries a 6-bit code plane; fp16 bits of the largest value its static premultiplier represents
                              // (7.5 * 2^-s). A larger stored value saturates its code (only that element's correction term degrades): status[1] |= status_bit,
                              // a WARNING the host can read (sn_numeric_status), never an error
    float scale3, shift3;
    long long wsplit_stride;  // halfs between channel splits in wpack
    long long in_lo_off, out_lo_off;
    long long out_code_off;   // OSPLIT 4 (hi + lo + fp8 code slots: a tensor with readers of both kinds): the slot plane, in halfs from `out`
    int in_cs, out_cs, out_coff, out_cp;
    int D, DX, tiles_x, tiles_y, tiles_z, total_tiles;   // volume = DX x D x D voxels (3-D nets: DX = D; 2-D nets: x = image index)
    int act;              // 0 relu, 1 sigmoid
    int mx_in_e8, mx_out_e8, mx_side_e8;   // 6-bit MX forms (mx_format.h): E8M0 exponent 127 - s of the static premultiplier of the code planes
                                           // of the input tensor / of out and pool_out / of side_out
    int nslab;
    int c8_last;          // 8-channel groups of the layer's LAST channel slab; every other slab holds the kernel's CS8 (pack_conv_host cuts them that way).
                          // (Round 4: the table of per-slab sizes this replaces lived in the kernel-argument segment, and "slab_c8[slab]" with a run-time
                          // index was a global_load_ubyte + s_waitcnt vmcnt(0) at EVERY slab boundary - a wait for all LDS-DMAs in flight plus a memory
                          // round trip, ~500 clocks twice per slab with the matrix pipe idle.)
    int bridge;           // f16x3 kernels on the ping-pong loop: a slab's last K-chunk is filled up with the next slab's first units (write_koff_part)
};

enum { EPI_STORE = 0, EPI_FINAL = 1, EPI_POOL2D = 2, EPI_SIDEPOOL = 3 };   // POOL2D (2-D nets): store epilogue fused with the 2x2 max-pool that follows;
// SIDEPOOL (3-D nets): the layer's own output is never stored - its two consumers, the 1x1x1 side convolution (+BN+sigmoid, one more
// MFMA chain on the in-register outputs) and the 2x2x2 max-pool, run in the epilogue and store THEIR outputs (nets/SurfaceNet.py:37-38,46-47)

template <int V> struct IntC { static constexpr int value = V; };
__device__ __forceinline__ float sn_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
// Numeric status of the values an epilogue stores: the layer fails (status bit, SN_ERR_RANGE) when an ACCUMULATOR is not finite (it overflowed
// or met inf - inf upstream; ReLU would map a NaN / -inf pre-activation to a clean 0, so the test has to look in front of the activation) or
// when a stored value leaves the fp16 range (its hi half is +inf).
// Both conditions are tracked with one packed VALU op per PAIR of values instead of two compares and a scalar OR per
// value: c = fma(x, 0, c) stays (+-)0 while every x is finite and turns into a sticky NaN at the first +-inf / NaN (inf * 0 = NaN); run over the
// ACCUMULATORS it finds a non-finite pre-activation (scale and shift are finite), run in packed fp16 over the stored hi halves it finds a value
// that left the fp16 range (hi = +inf). `asm volatile`: as plain compares the checks were sunk behind the tile's last store, with the 112
// pre-activations of merge_conv_a kept live (a quarter of them in scratch) and an s_waitcnt vmcnt(0) - i.e. a wait for every store in flight -
// per reload: 21,000 clocks of epilogue per tile where the stores alone take 4,900 (tools/probe/store_probe.hip; SN_TIMING 10, round 3)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sn_track_acc(f32x2_t &c, const f32x4 &a)
{
    const f32x2_t lo = __builtin_shufflevector(a, a, 0, 1), hi = __builtin_shufflevector(a, a, 2, 3);
    asm volatile("v_pk_fma_f32 %0, %1, 0, %0 op_sel_hi:[1,0,1]" : "+v"(c) : "v"(lo));
    asm volatile("v_pk_fma_f32 %0, %1, 0, %0 op_sel_hi:[1,0,1]" : "+v"(c) : "v"(hi));
}
// (round 4: the stored-value tracker is a packed MAX, same cost - every store epilogue stores non-negative values (ReLU / sigmoid / pooled ReLU), so the
// largest stored half is +inf exactly when a value left the fp16 range, and the same register also says whether a value exceeded the range of the
// 6-bit code plane its tensor carries: the saturation warning of ConvArgs::mx_sat_bits)
// (v_pk_max_f16 drops a quiet-NaN operand: a stored NaN half whose ACCUMULATOR was finite would pass. That takes a NaN folded scale / shift - e.g.
// sigmoid(NaN) in a 1x1x1 side convolution - and pack_conv_host rejects every non-finite folded constant at load time (SN_ERR_ARG naming the layer and
// channel; tests/test_gpu_numerics.py::test_non_finite_batchnorm_constants_are_rejected_at_load), so no NaN can enter behind sn_track_acc. ADVICE r4.)
__device__ __forceinline__ void sn_track_h2(unsigned &c, unsigned h2) { asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(c) : "v"(h2)); }
__device__ __forceinline__ unsigned sn_tracked_max_bits(unsigned h) { const unsigned lo = h & 0x7fffu, hi = (h >> 16) & 0x7fffu; return lo > hi ? lo : hi; }   // fp16 bits of the largest stored value
__device__ __forceinline__ bool sn_tracked_bad(const f32x2_t &c, unsigned h) { return !(c.x == 0.f && c.y == 0.f) || sn_tracked_max_bits(h) >= 0x7c00u; }

// hi/lo split of an fp32 value into two fp16 (hi = rn(y), lo = rn(y - hi)); |y - hi - lo| <= 2^-22 |y|
__device__ __forceinline__ void sn_split(float y, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)y;
    lo = (_Float16)(y - (float)hi);
}

// ---- inline-asm LDS reads with counted waits --------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int OFF>
__device__ __forceinline__ void lds_read128(half8 &d, unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset is 16-bit");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read32(int &d, unsigned addr)
{
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void lds_read64(long long &d, unsigned addr)
{
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read128i(v4i &d, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
typedef int v3i __attribute__((ext_vector_type(3)));
template <int OFF>
__device__ __forceinline__ void lds_read96i(v3i &d, unsigned addr)
{
    asm volatile("ds_read_b96 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
// one 48-bit code unit (4 channels: hi x4, lo x4; mx_format.h) -> bytes [6*half, 6*half+6) of a slot (half-slot writers of the pooling epilogues)
__device__ __forceinline__ void sn_mx6_store_unit(char *slot, int half, const _Float16 (&h)[4], const float (&lo)[4], int e8)
{
    unsigned w[2][2];
    sn_mx6_units(h, lo, h, lo, e8, w);
    unsigned short *d = reinterpret_cast<unsigned short *>(slot) + 3 * half;
    d[0] = (unsigned short)w[0][0]; d[1] = (unsigned short)(w[0][0] >> 16); d[2] = (unsigned short)w[0][1];
}
// E reads (tap offset, next chunk's activation fragments, MX operands) issued behind group g of a chunk when ET reads are spread PER per group
constexpr int sn_e_after(int g, int ET, int PER) { return g < 0 ? 0 : (g * PER >= ET ? 0 : ((g + 1) * PER <= ET ? PER : ET - g * PER)); }
// fp8 e4m3 pack of four fp32 values (round-to-nearest-even, saturating at +-448)
__device__ __forceinline__ int sn_pack_fp8x4(float a, float b, float c, float d)
{
    auto cl = [](float x) { return fminf(fmaxf(x, -448.f), 448.f); };
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(cl(a), cl(b), 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(cl(c), cl(d), r, true);
}
template <int N>
__device__ __forceinline__ void lgkm_wait()
{
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N < 15 ? N : 15));
    __builtin_amdgcn_sched_barrier(0);   // keep the MFMAs that consume the data below the wait (guide rule 18)
}
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p;
}
template <int AUX = 0>
__device__ __forceinline__ void dma16(const void *gsrc, void *lds_dst_wave_base)
{
    // 64 lanes x 16 B: lane l's bytes land at lds_dst_wave_base + 16*l (destination is lane-linear by construction)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_dst_wave_base, 16, 0, AUX);
}

// The same in the instruction's scalar-base form, M0 set by hand: lane l's 16 bytes from sbase + voff(l) land at lds_dst + 16 l. One SALU op (M0), no
// per-lane 64-bit address arithmetic per DMA (the one-wave-per-SIMD loop issues its DMAs between MFMAs, where every instruction counts).
__device__ __forceinline__ void dma16_s(const void *sbase, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}
// The same through a buffer descriptor: lanes whose byte offset fails the descriptor's range check (offset >= num_records, which
// includes "negative" offsets) deposit ZEROS in LDS (tools/probe/buflds_probe.hip) - the hardware does the 'same' padding.
__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, void *lds_dst_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_dst_wave_base, 16, (int)voff, 0, 0, 0);
}
// MFMAs of the one-wave-per-SIMD loop as inline asm with the accumulator TIED and in the AGPR file ("+a"). Through the builtins hipcc turns two
// thirds of that loop's 168 MFMAs per piece into their three-address form (result in a fresh register tuple), lets the 224 accumulators drift through
// the AGPR file and repairs the drift with ~150 v_accvgpr_read / _write per piece, accumulators parked in VGPRs in between. The compiler's hazard
// recogniser does not look into inline asm: the callers keep every VALU write of an operand >= 2 instructions away and put s_nops between the
// last MFMA and the epilogue's first accumulator read.
#define SN_STR_(x) #x
#define SN_STR(x) SN_STR_(x)
__device__ __forceinline__ void pw_mfma_f16(f32x4 &c, const half8 &a, const half8 &b)
{
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// K = 128 MX-scaled step on 6-bit operands (192 bits per lane); OPSEL picks the byte of `sa` that holds the A-side block scale
template <int OPSEL>
__device__ __forceinline__ void pw_mfma_mx6(f32x4 &c, const mx_v6i &a, const mx_v6i &b, int sa, int sb)
{
    static_assert(OPSEL >= 0 && OPSEL < 4, "scale byte");
#define SN_MX6_ASM(SEL) "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 " SEL " cbsz:" SN_STR(SN_MX_FMT) " blgp:" SN_STR(SN_MX_FMT)
    if constexpr (OPSEL == 0) asm volatile(SN_MX6_ASM("op_sel_hi:[0,0,0]") : "+a"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    else if constexpr (OPSEL == 1) asm volatile(SN_MX6_ASM("op_sel:[1,0,0] op_sel_hi:[0,0,0]") : "+a"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    else if constexpr (OPSEL == 2) asm volatile(SN_MX6_ASM("op_sel_hi:[1,0,0]") : "+a"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    else asm volatile(SN_MX6_ASM("op_sel:[1,0,0] op_sel_hi:[1,0,0]") : "+a"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
#undef SN_MX6_ASM
}
// The same on fp8 e4m3 operands (256 bits per lane); sa, sb: E8M0 scales of the weight / activation side in byte 0
__device__ __forceinline__ void pw_mfma_mx8(f32x4 &c, const v8i &a, const v8i &b, int sa, int sb)
{
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
// Halo-voxel validity bits kept in the top bits of a lane's precomputed halo offset (see stage_halo_buf): a set bit that applies to
// the tile at hand stays in the offset and pushes it out of the descriptor's range.
constexpr unsigned HB_ALWAYS = 1u << 31, HB_XLO = 1u << 30, HB_XHI = 1u << 29, HB_YLO = 1u << 28, HB_YHI = 1u << 27, HB_ZLO = 1u << 26,
                   HB_ZHI = 1u << 25, HB_OFFMASK = (1u << 25) - 1;

template <int KS, int DIL, int MF, int NF, int EPI, int SPLIT, int CS8, int PCH_, int NW_, int PADV_, int K2D = 0>
struct ConvCfg {
    static constexpr int R = (KS / 2) * DIL;
    static constexpr int RX = K2D ? 0 : R;                 // K2D: KSxKS taps over (y,z) only; x (the image index) has no halo
    static constexpr int XS = MF / 4;                      // x-slices per wave
    static constexpr int NW = NW_;                         // waves per workgroup
    static constexpr int NT = NW * 64;
    static constexpr bool F4 = (K2D == 2);                 // 2-D maps of 4x4 pixels: one MFMA voxel fragment = one whole 4x4 image
    static constexpr int TX = F4 ? NW * MF : NW * XS, TY = F4 ? 4 : 8, TZ = F4 ? 4 : 8;
    static constexpr int HX = TX + 2 * RX, HY = TY + 2 * R, HZ = TZ + 2 * R;
    static constexpr int HVOX = HX * HY * HZ;
    static constexpr int CS8MAX = CS8;                     // 8-channel groups per slab
    static constexpr int SLOTS = CS8 + PADV_;              // 16-byte slots per halo voxel (PADV: one pad slot)
    static constexpr int VS = SLOTS * 16;                  // LDS bytes per halo voxel
    static constexpr int PCH = PCH_;                       // K-chunks per weight piece (one barrier per piece)
    static constexpr int NPL = SPLIT ? 2 : 1;              // activation planes (hi [, lo | fp8 pair])
    static constexpr int NPLM = SPLIT == 1 ? 2 : 1;        // planes the f16 main loop reads (f16x3: hi and lo)
    static constexpr int FRAG = 1024 * NPL;                // weight bytes per (K-chunk, cout fragment)
    static constexpr int MFRAG = 1024 * NPLM;              // bytes of one f16 weight fragment set (hi [+ lo])
    static_assert(SPLIT < 2 || PCH_ == 2, "f16m8: a weight piece is 2 K-chunks + one MX step");
    static constexpr int MXF = SPLIT == 3 ? 0 : SN_MX_FMT; // code format of the MX step: 0 fp8 e4m3 (SPLIT 3), else the 6-bit form of mx_format.h
    static constexpr int WBUF = PCH * NF * FRAG;
    static constexpr int NTAP = (K2D ? 1 : KS) * KS * KS;
    static constexpr int KOFF_N = NTAP * CS8MAX + 24;      // + look-ahead padding (2 chunks; f16m8: one 8-group piece; bridged slabs: up to 7 units of the next slab)
    // one wave per SIMD (conv3d_f16_mfma, PWM loop): 4 waves x (8 voxel x NF cout) fragments, accumulators in AGPRs; the tap tables of ALL slabs
    // (x both halo buffers) are written once per launch instead of once per slab
    static constexpr bool PWM = (SPLIT == 3 || (SPLIT == 2 && SN_MX_FMT != 0)) && K2D == 0 && NW_ == 4 && KS == 3 && MF == 8 && PCH_ == 2;
    // tap tables of the one-wave-per-SIMD loop, written once per launch: a slab's table depends only on the halo buffer it sits in, on its first unit (bridge
    // pieces: 27 units per slab against 8 per piece - a function of slab mod 8) and on whether it is the tile's last: 16 tables per buffer, any number of slabs
    static constexpr int PW_TABS = 16;
    static constexpr int pw_tab(int kb, int slab, bool last) { return kb * PW_TABS + (last ? 8 : 0) + (slab & 7); }
    // LDS distance of voxel fragment m from fragment 0 of the same lane under the row-gap-4 map (frag_xyz: hx = wave * XS + (m >> 2), hy = (m & 3) + 4 (v >> 3)):
    // a compile-time constant, so the PWM loop addresses all fragments as one per-lane register + the read's immediate offset
    static constexpr int pw_xoff(int m) { return (((m >> 2) * HY + (m & 3)) * HZ) * VS; }
    // the same under the row-gap-2 map (hy = (m & 1) + 4 ((m >> 1) & 1) + 2 (v >> 3)); XGAP = the map this kernel's 3-D form uses (frag_xyz)
    static constexpr int XGAP = DIL == 2 ? kRowGapDil2 : kRowGap3x3;
    static constexpr int xoff_of(int m) { return XGAP == 4 ? pw_xoff(m) : (((m >> 2) * HY + (m & 1) + 4 * ((m >> 1) & 1)) * HZ) * VS; }
    // f16x3 3x3(x3) kernels on the ping-pong loop (PTAB): a slab's tap table depends only on the halo buffer it sits in, on its first unit (a function of
    // slab mod 4 with bridge chunks: 27 or 18 units per slab, 4 per chunk) and on whether it is the tile's last (b = 0, possibly fewer groups) - 8 tables per buffer, written once per launch instead of once per slab in a load slot
    static constexpr bool PTAB = SPLIT < 2 && NW_ == 8 && KS == 3 && PCH_ >= 2 && (K2D == 0 || SPLIT != 0) && !PWM;      // (= the kernels of the f16 / f16x3 ping-pong loop, 3-D and 2-D)
    static constexpr int KTAB_N = PWM ? PW_TABS * KOFF_N : (PTAB ? 8 * KOFF_N : KOFF_N);   // ints per halo buffer
    static constexpr int NSEG0 = (HVOX * VS + 1023) / 1024;
    static constexpr int NSEG = PWM ? (NSEG0 + NW_ - 1) / NW_ * NW_ : NSEG0;   // 1 KiB DMA segments per plane (one-wave-per-SIMD loop: the same number for every wave)
    static constexpr int XPLANE = NSEG * 1024;
    static constexpr int XBUF = XPLANE * NPL;
#ifndef SN_PW_WB3
#define SN_PW_WB3 0       // 1: one-wave-per-SIMD loop with a RING OF THREE weight-piece buffers where the LDS has room (the merge layers: 159 KB) - a piece's DMAs are issued two
                          // pieces ahead and the per-piece wait lets the newest ones fly. Built and measured in round 6 (profiles/r6/ab_r6_wb3.log): bit-identical, merge_conv_b
                          // 2.751 -> 2.793 ms, merge_conv_a 1.863 -> 1.861 - the per-piece vmcnt + barrier wait is the waves' skew at the barrier, not DMA latency. Off.
#endif
    static constexpr bool CST_LDS = (EPI == EPI_STORE) && NF >= 7;   // wide store epilogues: keep scale/shift in LDS so the compiler's vmcnt(0) before their use cannot serialise the stores (measured: merge_conv_a -4 %, narrower layers +3..6 % -> off there)
    static constexpr int EPI_CONST = CST_LDS ? NF * 16 * 4 * 2 : 0;   // scale, shift of this cout split, fp32
    static constexpr int NWB = (SN_PW_WB3 && PWM && 2 * XBUF + 3 * WBUF + 2 * KTAB_N * 4 + EPI_CONST <= 160 * 1024) ? 3 : 2;      // weight-piece buffers
    static constexpr int wb_next(int b) { return NWB == 3 ? (b == 2 ? 0 : b + 1) : (b ^ 1); }
    static constexpr int LDS_BYTES = 2 * XBUF + NWB * WBUF + 2 * KTAB_N * 4 + EPI_CONST;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
    // two 4-wave workgroups per CU only when 256 registers per lane are plausibly enough (accumulators = MF*NF*4)
    static constexpr int WG_PER_CU = (LDS_BYTES <= 80 * 1024 && NW == 4 && MF * NF <= 32) ? 2 : 1;
    // two waves per SIMD (8-wave workgroup or two 4-wave workgroups per CU): keep <= 256 registers per lane
    static constexpr int MIN_WAVES_PER_SIMD = (NW == 8 || WG_PER_CU == 2) ? 2 : 1;
};

// Does this configuration run the f16 / f16x3 ping-pong loop with bridge chunks (kernel: PPX, BRIDGE_OK)? - the launcher refuses a layer packed
// with bridge chunks on any other kernel
template <int KS, int SPLIT, int NW, int PCH, int NF, int K2D, int MF = 4>
constexpr bool sn_conv_has_bridge()
{
    return (SPLIT == 1 && NW == 8 && KS == 3 && PCH >= 2) ||
           (SPLIT == 2 && K2D == 0 && NW == 8 && KS == 3 && SN_MX_FMT != 0) ||
           ((SPLIT == 3 || (SPLIT == 2 && SN_MX_FMT != 0)) && K2D == 0 && NW == 4 && MF == 8 && KS == 3 && PCH == 2);
}

// OSPLIT: storage format of the OUTPUT tensor (defaults to SPLIT): lets an f16x3 layer feed an f16m8 layer.
template <int KS, int DIL, int MF, int NF, int EPI, int SPLIT, int CS8, int PCH_, int NW_, int PADV_, int K2D = 0, int OSPLIT_ = -1>
__global__ void __launch_bounds__(NW_ * 64, (ConvCfg<KS, DIL, MF, NF, EPI, SPLIT, CS8, PCH_, NW_, PADV_, K2D>::MIN_WAVES_PER_SIMD))
conv3d_f16_mfma(ConvArgs a)
{
    constexpr int OSPLIT = OSPLIT_ < 0 ? SPLIT : OSPLIT_;
    using C = ConvCfg<KS, DIL, MF, NF, EPI, SPLIT, CS8, PCH_, NW_, PADV_, K2D>;
    constexpr int NPL = C::NPL;
    __shared__ __attribute__((aligned(16))) char lds[C::LDS_BYTES];
    char *const xbuf = lds;                                   // [2][NPL][XPLANE]
    char *const wbuf = lds + 2 * C::XBUF;                     // [NWB][WBUF]
    int *const kbuf = reinterpret_cast<int *>(lds + 2 * C::XBUF + C::NWB * C::WBUF);   // [2][KOFF_N]  (PWM: [2][PW_TABS][KOFF_N])
    // epilogue constants live in LDS: a global load in the epilogue would make hipcc wait vmcnt(0), i.e. for every store
    // issued before it (measured: 21 us per tile of serialised store->load round trips in merge_conv_a)
    float *const cst = reinterpret_cast<float *>(lds + 2 * C::XBUF + C::NWB * C::WBUF + 2 * C::KTAB_N * 4);   // [2][NF*16]

    // the wave id IS wave-uniform, but anything derived from threadIdx is divergent to hipcc: without the readfirstlane every
    // loop and LDS-DMA destination indexed by it becomes an EXEC-masked (waterfall) loop (guide T20)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int v = lane & 15, kq = lane >> 4;
    int mx_sb = a.mx_in_e8;                                  // scale operand of the MX step's activation side (6-bit forms): loaded once, kept in a VGPR
    if constexpr (SPLIT == 3 || (SPLIT == 2 && SN_MX_FMT != 0)) asm volatile("" : "+v"(mx_sb));
    const int D = a.D, DX = a.DX;
    const size_t VOL = (size_t)DX * D * D;
    const int tstride = gridDim.x;
    auto slab_c8_of = [&](int sl) -> int { return sl + 1 == a.nslab ? a.c8_last : C::CS8MAX; };      // channel groups of slab sl (scalar arithmetic only)
    // XCD-aware walk (speed only): workgroup i is observed to run on XCD i % 8; within each round of gridDim.x tiles XCD x
    // takes the contiguous run [x*G/8, (x+1)*G/8) so that neighbouring tiles (shared halos) meet in one L2.
    int tile = ((gridDim.x & 7) == 0) ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    if (tile >= a.total_tiles) return;

    const char *const wsrc0 = reinterpret_cast<const char *>(a.wpack + (size_t)blockIdx.y * a.wsplit_stride);

    // ---- helpers ---------------------------------------------------------------------------------------------
    auto tile_origin = [&](int t, int &b, int &x0, int &y0, int &z0) {
        const int tz = t % a.tiles_z; t /= a.tiles_z;
        const int ty = t % a.tiles_y; t /= a.tiles_y;
        const int tx = t % a.tiles_x;
        b = t / a.tiles_x;
        x0 = tx * C::TX; y0 = ty * C::TY; z0 = tz * C::TZ;
    };
    // LDS-DMA of the halo tile of (tile t, channel groups [c0, c0+c8n)) into halo buffer xb. The NSEG*NPL one-KiB
    // segments are dealt round-robin to the waves; this call issues this wave's instructions [k0, k0+kn) and
    // returns how many it really issued (the tail waves own one segment less).
    constexpr int HT = (C::NSEG * NPL + C::NW - 1) / C::NW;   // halo DMA instructions per wave and slab
    constexpr int FULLP = ((C::NTAP * C::CS8MAX + 3) / 4 + C::PCH - 1) / C::PCH;      // weight pieces of a full channel slab
    // halo DMA instalment per piece (generic path); the buffer path issues its whole (cheap) set behind the first weight piece of a slab
    constexpr int HQ = (K2D == 0 || FULLP <= 1) ? HT : (HT + FULLP - 2) / (FULLP > 1 ? FULLP - 1 : 1);
    auto stage_halo = [&](int t, int c0, int c8n, int xb, int k0, int kn) -> int {
        int b, x0, y0, z0;
        tile_origin(t, b, x0, y0, z0);
        const _Float16 *in_b = a.in + (size_t)b * VOL * a.in_cs;
        int issued = 0;
        for (int k = k0; k < k0 + kn; ++k) {
            const int li = k * C::NW + wave;
            if (li >= C::NSEG * NPL) break;
            const int pl = li / C::NSEG, seg = li - pl * C::NSEG;
            const int slot = seg * 64 + lane;
            const int hv = slot / C::SLOTS, part = slot - hv * C::SLOTS;
            const int hz = hv % C::HZ, hy = (hv / C::HZ) % C::HY, hx = hv / (C::HZ * C::HY);
            const int gx = x0 - C::RX + hx, gy = y0 - C::R + hy, gz = z0 - C::R + hz;
            const bool ok = part < c8n && hv < C::HVOX && (unsigned)gx < (unsigned)DX && (unsigned)gy < (unsigned)D &&
                            (unsigned)gz < (unsigned)D;
            const _Float16 *p = in_b + ((size_t)(c0 + part) * VOL + ((size_t)(gx * D + gy) * D + gz)) * 8 + (pl ? a.in_lo_off : 0);
            dma16(ok ? (const void *)p : a.zero_page, xbuf + xb * C::XBUF + pl * C::XPLANE + seg * 1024);
            ++issued;
        }
        return issued;
    };
    // ---- buffer-addressed halo staging ----------------------------------------------------------------------------
    // The generic stage_halo above spends ~35 VALU + ~30 SALU instructions and an EXEC-masked branch per 1 KiB DMA on turning
    // (segment, lane) into a clamped global address. Here every lane keeps, per DMA slot k of its wave, ONE precomputed word:
    // the byte offset of its halo voxel relative to the tile's halo origin inside the channel slab, plus validity bits for the
    // volume faces (and "never valid" for the tail of the last segment). Per DMA: (word & keep-mask of the tile) + tile offset ->
    // voffset of a buffer_load..lds whose descriptor covers exactly the slab's c8n group planes; a voxel outside the volume keeps a
    // high bit, fails the range check, and the hardware writes zeros ('same' padding / PadLayer).
    //   3-D nets: six face bits + "never" in bits 25..31, offsets < 2^25 relative to the sample's slab.
    //   2-D nets (K2D; x = image index, no halo along x): four face bits (y, z) in bits 28..31, "never" = offset 0x0FFFFFF0; the
    //   descriptor starts at the TILE's first image, so offsets (up to one group plane of the whole chunk) stay < 2^28. Images past
    //   the end of a partial last tile read whatever follows inside the descriptor (or zeros): their outputs are never stored.
    // Preconditions (checked by launch_conv): only the first / last tile along an axis has out-of-volume halo voxels, and the
    // slab fits the offset field. The one-plane f16 mode of the 2-D nets (4-group slabs: up to 400 MB) keeps the generic path.
    constexpr bool BUFH = (K2D == 0) || (SPLIT != 0);
    constexpr bool PPM = SPLIT == 2 && K2D == 0 && NW_ == 8 && KS == 3 && SN_MX_FMT != 0;   // ping-pong K loop, f16m8 kernels (slab loop)
    constexpr bool PPX = SPLIT < 2 && NW_ == 8 && KS == 3 && PCH_ >= 2 && BUFH;          // ... f16 / f16x3 kernels
    constexpr bool PP = PPM || PPX;
    constexpr bool PWM = C::PWM;                   // one wave per SIMD (slab loop)
    constexpr bool UNI = PP || PWM;                // loops in which hipcc's divergence analysis loses wave-uniform values (stage_halo_buf)
    constexpr unsigned FB_YLO = K2D ? (1u << 31) : HB_YLO, FB_YHI = K2D ? (1u << 30) : HB_YHI, FB_ZLO = K2D ? (1u << 29) : HB_ZLO,
                       FB_ZHI = K2D ? (1u << 28) : HB_ZHI, FB_NEVER = K2D ? 0x0FFFFFF0u : HB_ALWAYS, FB_OFFMASK = K2D ? 0x0FFFFFFFu : HB_OFFMASK;
    unsigned hword[HT];
    if constexpr (BUFH) {
        const int xl = (a.tiles_x - 1) * C::TX, yl = (a.tiles_y - 1) * C::TY, zl = (a.tiles_z - 1) * C::TZ;
        static_for<0, HT>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const int li = k * C::NW + wave;
            const int seg = li % C::NSEG;
            const int slot = seg * 64 + lane;
            const int hv = slot / C::SLOTS, part = slot - hv * C::SLOTS;
            const int hz = hv % C::HZ, hy = (hv / C::HZ) % C::HY, hx = hv / (C::HZ * C::HY);
            unsigned o = ((unsigned)part * (unsigned)VOL + (unsigned)((hx * D + hy) * D + hz)) * 16u;
            if (hv >= C::HVOX || li >= C::NSEG * NPL) o = FB_NEVER;
            else {
                if constexpr (K2D == 0) {
                    if (hx < C::RX) o |= HB_XLO;
                    if (xl - C::RX + hx >= DX) o |= HB_XHI;
                }
                if (hy < C::R) o |= FB_YLO;
                if (yl - C::R + hy >= D) o |= FB_YHI;
                if (hz < C::R) o |= FB_ZLO;
                if (zl - C::R + hz >= D) o |= FB_ZHI;
            }
            hword[k] = o;
        });
    }
    // tile-dependent scalars of the buffer path: keep-mask and byte offset of the halo origin (may be negative)
    auto tile_halo_consts = [&](int x0, int y0, int z0, unsigned &keep, int &toff) {
        unsigned inv = K2D ? 0u : HB_ALWAYS;
        if constexpr (K2D == 0) {
            if (x0 == 0) inv |= HB_XLO;
            if (x0 == (a.tiles_x - 1) * C::TX) inv |= HB_XHI;
        }
        if (y0 == 0) inv |= FB_YLO;
        if (y0 == (a.tiles_y - 1) * C::TY) inv |= FB_YHI;
        if (z0 == 0) inv |= FB_ZLO;
        if (z0 == (a.tiles_z - 1) * C::TZ) inv |= FB_ZHI;
        keep = FB_OFFMASK | inv;
        toff = K2D ? ((y0 - C::R) * D + (z0 - C::R)) * 16 : (((x0 - C::RX) * D + (y0 - C::R)) * D + (z0 - C::R)) * 16;
    };
    // issues ALL of this wave's halo DMAs of (sample b [2-D: first image x0 of the tile], channel groups [c0, c0+c8n)) into halo buffer xb
    auto stage_halo_buf = [&](int b, unsigned keep, int toff, int c0, int c8n, int xb, int kb = 0, int ke = 1 << 20) -> int {      // [kb, ke): this call's instalment of the wave's HT slots
        // opaque to the optimiser: otherwise it hoists (hword[k] & keep) + toff and the descriptors of BOTH candidate tiles out of the
        // K loop as loop invariants (8 VGPRs + 16 SGPRs live across it) and the accumulators spill
        if constexpr (UNI) {      // (wave-uniform values that hipcc's divergence analysis loses inside the ping-pong piece loop)
            b = __builtin_amdgcn_readfirstlane(b); keep = __builtin_amdgcn_readfirstlane(keep);
            toff = __builtin_amdgcn_readfirstlane(toff); c0 = __builtin_amdgcn_readfirstlane(c0);
        }
        asm volatile("" : "+s"(b), "+s"(keep), "+s"(toff), "+s"(c0));
        const char *base0;
        int nrec;
        if constexpr (K2D == 0) {
            base0 = reinterpret_cast<const char *>(a.in) + 2 * ((size_t)b * VOL * a.in_cs + (size_t)c0 * VOL * 8);
            nrec = c8n * (int)VOL * 16;
        } else {      // b = x0: the descriptor begins at the tile's first image inside group plane c0 and ends with group plane c0+c8n-1
            base0 = reinterpret_cast<const char *>(a.in) + 2 * ((size_t)c0 * VOL * 8 + (size_t)b * D * D * 8);
            nrec = (c8n * (int)VOL - b * D * D) * 16;
        }
        if constexpr (UNI) {      // ... and of the descriptor itself: a buffer_load with a "divergent" resource becomes a waterfall loop
            const unsigned long long bq = (unsigned long long)(size_t)base0;
            base0 = (const char *)(size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bq >> 32)) << 32) |
                                           (unsigned)__builtin_amdgcn_readfirstlane((int)bq));
            nrec = __builtin_amdgcn_readfirstlane(nrec);
        }
        int issued = 0;
        static_for<0, HT>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const int li = k * C::NW + wave;
            // (only the last slot of a wave can fall behind the last segment: HT = ceil(NSEG*NPL / NW))
            if ((k + 1 < HT || C::NSEG * NPL == HT * C::NW || li < C::NSEG * NPL) && k >= kb && k < ke) {
                const char *base = (NPL > 1 && li >= C::NSEG) ? base0 + 2 * a.in_lo_off : base0;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, nrec, 0x00020000);
                dma16_buf(rs, (hword[k] & keep) + (unsigned)toff, xbuf + xb * C::XBUF + li * 1024);
                ++issued;
            }
        });
        return issued;
    };
    // s_waitcnt vmcnt(n) with a run-time (wave-uniform) n; raw s_barrier (a __syncthreads() would make hipcc drain
    // vmcnt(0) because LDS-DMAs are pending, defeating the counted wait)
    auto wg_barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };
    // BRIDGE chunks (f16x3 kernels on the ping-pong loop; all slabs of a layer hold the same number of channel groups): a slab's NTAP * c8n
    // (tap, group) units are not a multiple of the 4 a K-chunk holds - 27 = 6.75 chunks in the 3-D nets, 18 = 4.5 in the similarityNet - and
    // used to be padded with zero weights slab by slab. Now the last chunk of a slab is filled up with the FIRST units of the next slab of the
    // tile (b of them), which then starts at its unit o = b: 27 chunks per four 3-D slabs instead of 28, 9 per two 2-D slabs instead of 10.
    // The next slab's halo tile sits in the OTHER halo buffer and has landed by then (its DMAs are issued in the slab's first piece; the
    // segment in front of the bridge chunk waits vmcnt(0) before its barrier): the bridge entries of the tap table simply carry that buffer's
    // distance. pack_conv_host lays the weights out in the same unit order and decides whether a layer qualifies (a.bridge).
    // The f16m8 kernels work in PIECES of 8 units (two f16 chunks + one MX step over the same 8): 27 units = 3.375 pieces were run as 4, the
    // 4th with one chunk, three units and a full-size weight DMA. Bridged, merge_conv_a's 8 slabs are 27 pieces instead of 32 and merge_conv_b's
    // 13 are 44 instead of 52; the bridge piece is a slab's third or fourth, behind the vmcnt(0) of the second piece's MX load slot.
    constexpr bool BRIDGE_OK = (PPX && SPLIT == 1) || PPM || PWM;
    constexpr int UM = SPLIT >= 2 ? 8 : 4;                   // units per chunk / per piece: what a slab's unit count is rounded up to
    constexpr int BSTEP = (UM - (C::NTAP * C::CS8MAX) % UM) % UM;      // bridged layers (all slabs hold CS8MAX groups): a slab starts this many units later (mod UM) than its predecessor
    const bool bridge = BRIDGE_OK && a.bridge != 0;
    // units of slab `slab` in its chunks: GU - o of its own (o: taken by the slab before) + b of the next slab's
    auto slab_units = [&](int c8n, int slab, int &o, int &b) {
        const int GU = C::NTAP * c8n;
        o = 0; b = 0;
        if (bridge) {
            o = (slab * ((UM - (GU & (UM - 1))) & (UM - 1))) & (UM - 1);
            b = (slab + 1 == a.nslab) ? 0 : ((UM - ((GU - o) & (UM - 1))) & (UM - 1));
        }
        return GU - o + b;
    };
    auto chunks_of = [&](int c8n, int slab) { int o, b; return (slab_units(c8n, slab, o, b) + 3) >> 2; };
    // K-chunks the weight STREAM holds for a slab (f16m8 pads every slab to whole pieces)
    auto wchunks_of = [&](int c8n, int slab) { int o, b; return SPLIT >= 2 ? (((slab_units(c8n, slab, o, b) + 7) >> 3) << 1) : chunks_of(c8n, slab); };
    const int wchunk0 = wchunks_of(slab_c8_of(0), 0);        // ... of a tile's first slab
    // tap table of slab `slab` (c8n groups) into table buffer kb (= the halo buffer that holds the slab): entry g = LDS byte offset of
    // (tap, group) unit g's 16-byte slot relative to a voxel's own slot
    auto write_koff_part = [&](int c8n, int kb, int slab, int t0, int nt) {
        int uo, ub;
        const int own = slab_units(c8n, slab, uo, ub) - ub, nchunk = (own + ub + 3) >> 2;
        int *k = kbuf + (PWM ? C::pw_tab(kb, slab, slab + 1 == a.nslab) : (C::PTAB ? kb * 8 + ((slab + 1 == a.nslab) ? 4 : 0) + (slab & 3) : kb)) * C::KOFF_N;
        for (int g = t0; g < (nchunk + 4) * 4; g += nt) {
            int o = 0;
            if (g < own + ub) {
                const int u = g < own ? g + uo : g - own;                       // unit of this slab | bridge: of the next one, in the other halo buffer
                const int far = g < own ? 0 : (1 - 2 * kb) * C::XBUF;
                const int tap = C::CS8MAX == 1 ? u : u / c8n, c8 = C::CS8MAX == 1 ? 0 : u - tap * c8n;      // (one-group slabs: no run-time division in the load slot that carries the table write)
                const int dz = tap % KS, dy = (tap / KS) % KS, dx = tap / (KS * KS);
                o = ((dx * DIL * C::HY + dy * DIL) * C::HZ + dz * DIL) * C::VS + c8 * 16 + far;
            }
            k[g] = o;
        }
    };
    auto write_koff = [&](int c8n, int kb, int slab) { write_koff_part(c8n, kb, slab, tid, C::NT); };
    // LDS-DMA of `nch` K-chunks of packed weights starting at byte offset `off` of this split's stream
    auto stage_w = [&](size_t off, int nch, int wbi) {
        const int cnt = nch * NF * NPL;
        const char *src = wsrc0 + off;
        char *dst = wbuf + wbi * C::WBUF;
        for (int i = wave; i < cnt; i += C::NW) dma16(src + (size_t)i * 1024 + lane * 16, dst + i * 1024);
    };

    // Position of voxel fragment m's lane inside the workgroup's output tile. Default: wave w owns x-slices [w*XS, (w+1)*XS), a fragment is
    // 2 y-rows x 8 z. PMAP (EPI_SIDEPOOL): wave w owns x in {2(w&3), 2(w&3)+1} x 4 y-rows, so that every 2x2x2 pooling cell lies inside ONE
    // wave (x partner = fragment m+2, y partner = lane^8, z partner = lane^1).
    constexpr bool PMAP = (EPI == EPI_SIDEPOOL);
    static_assert(!PMAP || ((MF == 4 || MF == 8) && NW_ == 8 && K2D == 0), "EPI_SIDEPOOL: 8 waves x 4 (8) fragments over an 8x8x8 (16x8x8) tile");
    auto frag_xyz = [&](int m, int &hx, int &hy, int &hz) {
        if constexpr (C::F4) { hx = wave * MF + m; hy = v >> 2; hz = v & 3; }
        else if constexpr (PMAP) { hx = (MF / 2) * (wave & 3) + (m >> 1); hy = 2 * (wave >> 2) + (m & 1) + 4 * (v >> 3); hz = v & 7; }   // (MF = 8: four x-slices per wave)
        else if constexpr (K2D == 0 && KS == 3 && C::XGAP == 4) { hx = wave * C::XS + (m >> 2); hy = (m & 3) + 4 * (v >> 3); hz = v & 7; }
        else if constexpr (K2D == 0 && KS == 3 && C::XGAP == 2) { hx = wave * C::XS + (m >> 2); hy = (m & 1) + 4 * ((m >> 1) & 1) + 2 * (v >> 3); hz = v & 7; }
        else if constexpr (K2D == 1 && kRowGap2D == 4 && C::VS == 32) { hx = wave * C::XS + (m >> 2); hy = (m & 3) + 4 * (v >> 3); hz = v & 7; }
        else { hx = wave * C::XS + (m >> 2); hy = 2 * (m & 3) + (v >> 3); hz = v & 7; }
    };
    int xbase[MF];
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        int hx, hy, hz;
        frag_xyz(m, hx, hy, hz);
        xbase[m] = ((hx * C::HY + hy) * C::HZ + hz) * C::VS;
    }
    const unsigned xbuf_a = lds_addr(xbuf), wbuf_a = lds_addr(wbuf) + lane * 16, kbuf_a = lds_addr(kbuf) + kq * 4;

    if constexpr (C::CST_LDS)
        for (int i = tid; i < NF * 16; i += C::NT) {
            cst[i] = a.scale[blockIdx.y * NF * 16 + i];
            cst[NF * 16 + i] = a.shift[blockIdx.y * NF * 16 + i];
        }
    // EPI_SIDEPOOL: the side conv's A fragments stay in registers for the whole kernel (K order = the order in which a lane's accumulator
    // registers hold the channels: k = 8*kq + j <-> channel 16*(2q + (j >= 4)) + 4*kq + (j & 3) of K-chunk q; packed accordingly on the host)
    constexpr int NQ = (NF + 1) / 2;
    half8 sw[EPI == EPI_SIDEPOOL ? NQ : 1][2];
    if constexpr (EPI == EPI_SIDEPOOL) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            sw[q][0] = *reinterpret_cast<const half8 *>(a.side_w + ((size_t)(q * 2 + 0) * 64 + lane) * 8);
            sw[q][1] = *reinterpret_cast<const half8 *>(a.side_w + ((size_t)(q * 2 + 1) * 64 + lane) * 8);
        }
    }
    // ---- prologue: first halo tile, its tap table, first weight piece ----------------------------------------
    {
        const int c8n = slab_c8_of(0);
        if constexpr (BUFH) {
            int b, x0, y0, z0, toff;
            unsigned keep;
            tile_origin(tile, b, x0, y0, z0);
            tile_halo_consts(x0, y0, z0, keep, toff);
            stage_halo_buf(K2D ? x0 : b, keep, toff, 0, c8n, 0);
        } else stage_halo(tile, 0, c8n, 0, 0, HT);
        if constexpr (C::PTAB) {      // the 8 tables per halo buffer (see ConvCfg::PTAB): slabs 0..3 as representatives of their residue, the last slab once per residue
            for (int kb = 0; kb < 2; ++kb)
                for (int sl = 0; sl < a.nslab; ++sl)
                    if (sl < 4 || sl + 1 == a.nslab) write_koff(slab_c8_of(sl), kb, sl);
        } else
        if constexpr (PWM) {      // the 16 tables per halo buffer (ConvCfg::pw_tab; the bridge entries depend on the buffer): slabs 0..7 as representatives, the last slab
            for (int kb = 0; kb < 2; ++kb)
                for (int sl = 0; sl < a.nslab; ++sl)
                    if (sl < 8 || sl + 1 == a.nslab) write_koff(slab_c8_of(sl), kb, sl);
        } else
        write_koff(c8n, 0, 0);
        const int nch = wchunks_of(c8n, 0);
        stage_w(0, nch < C::PCH ? nch : C::PCH, 0);
        if constexpr (C::NWB == 3) stage_w((size_t)C::PCH * NF * C::FRAG, C::PCH, 1);      // ring of three: the layer's second piece as well (every slab of such a layer holds >= 2 pieces)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        wg_barrier();
    }
    int xb = 0;     // halo / tap-table buffer holding the current slab
    int wbi = 0;    // weight buffer holding the current piece
    long long t_vm = 0, t_bar = 0, n_piece = 0, t_rel = 0;      // SN_TIMING accumulators
    const long long t_kernel0 = SN_TIMING ? __builtin_readcyclecounter() : 0;
    bool bad = false;   // a stored value left the fp16 range / is NaN (checked on the fp32 value in the epilogue)
    f32x2_t trk_acc = {0.f, 0.f};   // ... the store epilogues' form of the same test (sn_track_acc / sn_track_h2)
    unsigned trk_h = 0;

    for (; tile < a.total_tiles; tile += tstride) {
        int b, x0, y0, z0;
        tile_origin(tile, b, x0, y0, z0);
        // halo constants of this tile (later slabs) and of this workgroup's next tile (its first slab is staged during the last slab here)
        int cur_toff = 0, nxt_toff = 0, nxt_b = 0;
        unsigned cur_keep = 0, nxt_keep = 0;
        if constexpr (BUFH) {
            tile_halo_consts(x0, y0, z0, cur_keep, cur_toff);
            if (tile + tstride < a.total_tiles) {
                int nx0, ny0, nz0;
                tile_origin(tile + tstride, nxt_b, nx0, ny0, nz0);
                tile_halo_consts(nx0, ny0, nz0, nxt_keep, nxt_toff);
                if constexpr (K2D != 0) nxt_b = nx0;                  // 2-D nets: the descriptor is tile-relative along x
            }
        }

        f32x4 acc[MF][NF];
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

        int c0 = 0;
        size_t woff = 0;   // byte offset of the current slab in the weight stream
        // The one-slot offset between the two wave groups is set up per TILE (group 1 waits one barrier here, group 0 one barrier
        // behind its last segment), so that both groups run the epilogue at the same time. With a free-running offset the tile boundary costs
        // two slots of (epilogue + load) each - a group's epilogue sits in its load slot while the partner's short MFMA burst ends and waits -
        // which merge_conv_a's store epilogue cannot afford (+4 % against the non-ping-pong kernel before this, A/B r3e).
        if constexpr (PP) { if (wave >= C::NW / 2) wg_barrier(); }
        const long long t_tile0 = SN_TIMING == 10 ? __builtin_readcyclecounter() : 0;     // 10: per tile {epilogue, K loop}
        // PWM: what a piece hands to its successor - the operand fragments of the successor's first f16 chunk (read during the MX burst), the tap
        // offsets of its second chunk and of its MX step. A tile's first piece loads them cold, right here.
        half8 pw_xf[PWM ? MF : 1], pw_wf[PWM ? NF : 1];
        const unsigned pw_wv = (unsigned)(wave * 1024 + lane * 16);      // this lane's bytes within this wave's first KiB of a weight piece (weight DMAs)
        // Per-slab scalars carried from slab to slab instead of re-derived (slab_units with its branches, three to four times per slab): the slab's first unit
        // (all loops), and - PWM, where the slab boundary is the one place with no MFMA in flight: ~120 scalar instructions there were 480 clocks per slab,
        // 3 % of the kernel - the base of the halo tile staged during the slab (one 8-channel group plane further per slab; behind the tile's last slab the
        // next tile's first group)
        int sl_o = 0;             // first (tap, group) unit of the current slab's chunk / piece sequence (bridge chunks: the units before it went to the slab before)
        const char *pw_hnext = nullptr, *pw_hptr = nullptr;
        if constexpr (PWM) {
            const bool hn = tile + tstride < a.total_tiles;
            pw_hptr = reinterpret_cast<const char *>(a.in) + 2 * ((size_t)b * VOL * a.in_cs) + VOL * 16;                 // slab 0 stages slab 1
            pw_hnext = reinterpret_cast<const char *>(a.in) + 2 * ((size_t)(hn ? nxt_b : b) * VOL * a.in_cs);             // (no next tile: this tile's first slab once more, the buffer is idle)
        }
        int pw_koB = 0;
        long long pw_k2 = 0;
        if constexpr (PWM) {
            const unsigned tab0 = kbuf_a + (unsigned)(C::pw_tab(xb, 0, a.nslab == 1) * (C::KOFF_N * 4));
            int koA;
            lds_read32<0>(koA, tab0);
            lds_read32<16>(pw_koB, tab0);
            lds_read64<0>(pw_k2, tab0 + kq * 4);
            lgkm_wait<0>();
            const unsigned xa = xbuf_a + xb * C::XBUF + (unsigned)xbase[0] + (unsigned)koA, wp0 = wbuf_a + wbi * C::WBUF;
            static_for<0, MF>([&](auto mc) { constexpr int m = decltype(mc)::value; lds_read128<C::xoff_of(m)>(pw_xf[m], xa); });
            static_for<0, NF>([&](auto nc) { constexpr int n = decltype(nc)::value; lds_read128<n * 1024>(pw_wf[n], wp0); });
            lgkm_wait<0>();
        }
        // (launch_conv: nslab >= 1. Said out loud because the zero-trip path around the slab loop, never taken, otherwise meets the loop's exit in front of the
        // epilogue with the accumulators in a different register assignment: ~220 AGPR-to-AGPR copies per tile on the path that is taken)
        if constexpr (PWM) __builtin_assume(a.nslab >= 1);       // (the eight-wave kernels: conv4_x / conv1_3 +1 % with it - left as they were)
        for (int slab = 0; slab < a.nslab; ++slab) {
            const bool last_slab = slab + 1 == a.nslab;
            const int c8n = last_slab ? a.c8_last : C::CS8MAX;
            // units of this slab's chunks / pieces: GU - o of its own (o: taken by the slab before) + b of the next slab's (slab_units, branch-free on the carried o)
            const int su_o = sl_o, su_b = (UM - ((C::NTAP * c8n - su_o) & (UM - 1))) & ((bridge && !last_slab) ? UM - 1 : 0);
            const int units = C::NTAP * c8n - su_o + su_b;
            const int nchunk = (units + 3) >> 2, wchunk = SPLIT >= 2 ? ((units + 7) >> 3) << 1 : nchunk;
            const int npiece = (nchunk + C::PCH - 1) / C::PCH;
            // what comes after this slab: next slab of this tile, or slab 0 of this workgroup's next tile
            const int ntile = last_slab ? tile + tstride : tile;
            const bool have_next = ntile < a.total_tiles;
            const int nslab_i = last_slab ? 0 : slab + 1;
            const bool nlast = nslab_i + 1 == a.nslab;
            const int nc8n = nlast ? a.c8_last : C::CS8MAX;
            // ... and the K-chunks its weight stream holds (the size of its first weight piece)
            const int n_o = (bridge && !last_slab) ? (su_o + BSTEP) & (UM - 1) : 0;
            const int n_units = C::NTAP * nc8n - n_o + ((UM - ((C::NTAP * nc8n - n_o) & (UM - 1))) & ((bridge && !nlast) ? UM - 1 : 0));
            const int n_wchunk = SPLIT >= 2 ? ((n_units + 7) >> 3) << 1 : (n_units + 3) >> 2;
            const int nc0 = last_slab ? 0 : c0 + c8n;
            const size_t nwoff = last_slab ? 0 : woff + (size_t)wchunk * NF * C::FRAG;
            if constexpr (!PP && !PWM) { if (have_next) write_koff(nc8n, xb ^ 1, nslab_i); }
            // the next halo tile is fetched in npiece-1 instalments, each issued right after a weight piece so that a
            // counted vmcnt can wait for the weights while the newest halo DMAs stay in flight
            // Instalment size HQ is a compile-time constant (sized for a full slab) so that the wait in front of the barrier is a
            // fixed s_waitcnt vmcnt(HQ) or vmcnt(0), one scalar branch; the last instalment of a short slab takes whatever is left.
            int hdone = 0;

            if constexpr (PWM) {
                // ---- ONE WAVE PER SIMD (round 4): 4 waves x (8 voxel x NF cout) fragments, accumulators in AGPRs ------------------------------
                // The ping-pong loops below put two waves on a SIMD and let one load while the other computes; every hand-over is a workgroup
                // barrier (6 per weight piece, each ~100-190 clocks of idle matrix pipe), and a wave's (4 + NF) operand reads serve 4 NF MFMAs.
                // Here a wave owns the SIMD and 512 registers: 8 x NF accumulator fragments (224 AGPRs), (8 + NF) reads per 8 NF MFMAs - a third fewer
                // LDS reads per MFMA, half the weight-fragment reads per workgroup - and NOTHING is issued outside an MFMA burst: a piece is three
                // bursts of 8 NF MFMAs (f16 chunk 2p, f16 chunk 2p+1, the MX step), and behind MFMA i of a burst sits at most one other
                // instruction - an operand read of the NEXT burst (double-buffered registers), an LDS-DMA of the next weight piece / next halo
                // tile, a register shuffle of the 6-bit operands (MI355X_MICROARCH.md: a 16-clock MFMA hides ~2 single-issue instructions of its
                // own wave). sched_barrier(0) behind every pair pins the order. ONE barrier per piece, between the second f16 burst and the MX
                // burst: by then every wave has read all it needs from the piece's weight buffer (so the piece after next may be fetched into
                // it) and its share of the next piece's DMAs has landed (the MX burst reads the next piece's first fragments).
                // Same K order, same MFMAs per accumulator as the ping-pong loop: bit-identical results.
                static_assert(SPLIT >= 2 && C::PCH == 2 && BUFH && EPI != EPI_SIDEPOOL && C::NSEG % C::NW == 0 && C::XPLANE + C::xoff_of(MF - 1) < 65536,
                              "one-wave-per-SIMD loop: f16m8 3x3x3 kernels");
                long long pws1 = 0;                    // SN_TIMING 4: per slab {everything in front of the piece loop since the previous slab's last piece, the piece loop}
                constexpr int mxo = 2 * NF * 1024;
                constexpr int WCNT = C::PCH * NF * NPL, WPW = (WCNT + C::NW - 1) / C::NW;      // 1 KiB DMAs per weight piece / per wave
                constexpr int NM = MF * NF;
                // burst A: behind its first 16 MFMAs (operand reads) one DMA slot every DSP MFMAs - the next weight piece's WPW and (a slab's first two pieces) HT / 2 halo DMAs
                constexpr int DSP = (NM - 16) / (WPW + HT / 2) >= 3 ? 3 : 2;
                static_assert(MF + NF <= 16 && 16 + DSP * (WPW + HT / 2 - 1) < NM && 4 + 2 * NF + 15 < NM, "filler schedule exceeds the burst");
                const unsigned tab_a = kbuf_a + (unsigned)(C::pw_tab(xb, slab, last_slab) * (C::KOFF_N * 4));
                const unsigned ntab_a = kbuf_a + (unsigned)(C::pw_tab(xb ^ 1, nslab_i, nlast) * (C::KOFF_N * 4));
                const unsigned xs_a = xbuf_a + xb * C::XBUF + (unsigned)xbase[0];
                const unsigned nxs_a = xbuf_a + (xb ^ 1) * C::XBUF + (unsigned)xbase[0];
                // the halo tile staged during this slab's first two pieces: the next slab's / the next tile's first slab (carried pointers, see the tile loop)
                const bool hnext = last_slab && have_next;
                int htoff = hnext ? nxt_toff : cur_toff;
                unsigned hkeep = hnext ? nxt_keep : cur_keep;
                htoff = __builtin_amdgcn_readfirstlane(htoff); hkeep = __builtin_amdgcn_readfirstlane(hkeep);
                const char *const hbase = last_slab ? pw_hnext : pw_hptr;
                auto halo_rsrc = [&](int plane) {
                    const char *base = hbase + (plane ? 2 * a.in_lo_off : 0);
                    const unsigned long long bq = (unsigned long long)(size_t)base;
                    base = (const char *)(size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bq >> 32)) << 32) |
                                                  (unsigned)__builtin_amdgcn_readfirstlane((int)bq));
                    return __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, __builtin_amdgcn_readfirstlane((int)VOL * 16), 0x00020000);
                };
                // halo descriptors of the slab staged during this one: the f16 plane goes out with the slab's first piece, the code plane with its second
                static_assert(HT % 2 == 0 && C::NSEG * NPL == HT * C::NW && (HT / 2) * C::NW == C::NSEG && WCNT % C::NW == 0, "PWM DMA schedule");
                constexpr int HH = HT / 2;
                const __amdgpu_buffer_rsrc_t rs_hi = halo_rsrc(0), rs_lo = halo_rsrc(1);
                const unsigned hdst = lds_addr(xbuf) + (unsigned)((xb ^ 1) * C::XBUF + wave * 1024);      // + k * NW KiB: this wave's k-th halo segment
                int p = 0;
                if constexpr (SN_TIMING == 4) { pws1 = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (t_rel != 0) t_vm += pws1 - t_rel; }
                do {             // (launch_conv: every slab of a PWM layer has at least two pieces)
                    // ONE loop body for all pieces (a second, specialised copy for the slab's first piece - written out, or peeled off by the optimiser when
                    // it can see "p == 0" - made hipcc spill 75 registers and shuffle accumulators between the register files around every MFMA)
                    int p_opaque = p;
                    asm volatile("" : "+s"(p_opaque));
                    const bool first = p_opaque == 0;
                    const unsigned wp = wbuf_a + wbi * C::WBUF, wpn = wbuf_a + C::wb_next(wbi) * C::WBUF;
                    const bool more = p + 1 < npiece;
                    // the piece after this one: the slab's next, else the first piece of the next slab / tile (tap table and halo tile of the OTHER buffer)
                    const unsigned nk_a = more ? tab_a + (unsigned)(8 * (p + 1)) * 4 : ntab_a;
                    const unsigned nx_a = more ? xs_a : nxs_a;
                    // the weight piece fetched during this one: two buffers - the next piece; ring of three - the piece after next (the slab's, else piece 0 / 1 of the
                    // next slab / tile: every slab holds at least two), into the buffer the PREVIOUS piece was read from (everybody left it at that piece's barrier)
                    size_t w_off;
                    if constexpr (C::NWB == 3) w_off = p + 2 < npiece ? woff + (size_t)(2 * p + 4) * NF * C::FRAG : (have_next ? nwoff : 0) + (size_t)(2 * (p + 2 - npiece)) * NF * C::FRAG;
                    else w_off = more ? woff + (size_t)(2 * p + 2) * NF * C::FRAG : (have_next ? nwoff : 0);
                    const char *const wsrc = wsrc0 + w_off;
                    const unsigned wdst_a = lds_addr(wbuf) + (unsigned)((C::NWB == 3 ? C::wb_next(C::wb_next(wbi)) : (wbi ^ 1)) * C::WBUF + wave * 1024);
                    const bool halo_now = p_opaque < 2;           // (pieces 0 and 1 of the slab carry the next halo tile's DMAs)
                    __amdgpu_buffer_rsrc_t rs_p = first ? rs_hi : rs_lo;
                    unsigned hdst_p = hdst + (first ? 0u : (unsigned)(HH * C::NW * 1024));
                    asm volatile("" : "+s"(rs_p), "+s"(hdst_p));
                    half8 xfB[MF], wfB[NF];
                    int koA_n = 0;
                    // SN_TIMING (diagnostic builds; PWM loop): 1 {burst A, burst B}, 2 {vmcnt wait, barrier}, 3 {burst M, whole piece} per piece
                    long long pwt[6] = {0, 0, 0, 0, 0, 0};
#define PW_T(i) do { if constexpr (SN_TIMING >= 1 && SN_TIMING <= 3) { pwt[i] = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } } while (0)
                    PW_T(0);
                    // ---- burst A: f16 chunk 2p; behind it: chunk 2p+1's operands, the next piece's tap offsets, the next weight piece's DMAs
                    const unsigned kosB = xs_a + (unsigned)pw_koB;
                    static_for<0, NM>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, n = i / MF, m = i % MF;
                        pw_mfma_f16(acc[m][n], pw_wf[n], pw_xf[m]);
                        if constexpr (i == 0) lds_read128<NF * 1024>(wfB[0], wp);
                        else if constexpr (i <= MF) lds_read128<C::xoff_of(i >= 1 && i <= MF ? i - 1 : 0)>(xfB[i >= 1 && i <= MF ? i - 1 : 0], kosB);
                        else if constexpr (i < MF + NF) lds_read128<(NF + (i < MF + NF ? i - MF : 0)) * 1024>(wfB[i < MF + NF ? i - MF : 0], wp);
                        else if constexpr (i >= 16 && (i - 16) % DSP == 0 && (i - 16) / DSP < WPW + HH) {
                            constexpr int q = (i - 16) / DSP;                // DMA slot q of the burst: weights and halo segments alternate while both last
                            constexpr int MN = WPW < HH ? WPW : HH;
                            constexpr bool is_h = q < 2 * MN ? (q % 2 == 1) : (HH > WPW);
                            if constexpr (!is_h) {
                                constexpr int k = q < 2 * MN ? q / 2 : q - HH;      // this wave's k-th KiB of the next weight piece
                                dma16_s(wsrc, pw_wv + k * (C::NW * 1024), wdst_a + k * (C::NW * 1024));
                            } else {
                                constexpr int j = q < 2 * MN ? q / 2 : q - WPW;
                                if (halo_now) {
                                    const unsigned hw = first ? hword[j] : hword[HH + j];
                                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(hdst_p + (unsigned)(j * C::NW * 1024)),
                                                 "v"((hw & hkeep) + (unsigned)htoff), "s"(rs_p) : "memory");
                                }
                            }
                        }
                        PW_SB;
                    });
                    lgkm_wait<0>();
                    PW_T(1);
                    // ---- burst B: f16 chunk 2p+1 (a slab's stream is padded to whole pieces: an absent chunk multiplies zero weights); behind it: the
                    // MX step's 6-bit weight fragments with their scales, and the code slots of its first two voxel fragments (the others follow
                    // inside the MX burst, which walks the voxel fragments in its OUTER loop: 3 instead of 8 fragments' slots live at a time)
                    const unsigned ks0 = xs_a + (unsigned)(int)pw_k2, ks1 = xs_a + (unsigned)(int)(pw_k2 >> 32);
                    v4i x8h[MF][2], wa4[NF];
                    long long wb2[NF], wsc = 0;
                    constexpr bool FP8 = C::MXF == 0;                    // fp8 e4m3 codes: whole 16-byte slots / 2 x 16 weight bytes per lane, no block scales
                    v4i wa4b[FP8 ? NF : 1];                              // fp8: the second 16 bytes of a lane's weight operand (6-bit forms: wb2, 8 bytes)
                    typedef std::conditional_t<FP8, v8i, mx_v6i> mx_op;
                    mx_op x8[MF];
                    // 6-bit forms: two 12-byte code slots (read whole, 16 bytes each) -> one 192-bit operand: the second slot moves down by one register;
                    // fp8: the two slots ARE the 256-bit operand
                    auto form_x8 = [&](auto mc) {
                        constexpr int mm = decltype(mc)::value;
                        asm volatile("" : "+v"(x8h[mm][0]), "+v"(x8h[mm][1]));
                        if constexpr (FP8) x8[mm] = __builtin_shufflevector(x8h[mm][0], x8h[mm][1], 0, 1, 2, 3, 4, 5, 6, 7);
                        else x8[mm] = __builtin_shufflevector(x8h[mm][0], x8h[mm][1], 0, 1, 2, 4, 5, 6);
                        asm volatile("" : "+v"(x8[mm]));
                    };
                    constexpr int RB = 4 + 2 * NF;                        // reads 1 .. RB behind MFMAs 1 .. RB
                    static_for<0, NM>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, n = i / MF, m = i % MF;
                        pw_mfma_f16(acc[m][n], wfB[n], xfB[m]);
                        if constexpr (i == 0) { if constexpr (!FP8) lds_read64<mxo + 1024 + 8>(wsc, wp); }
                        else if constexpr (i <= 4) {
                            constexpr int mm = (i - 1) / 2, sl = (i - 1) & 1;
                            lds_read128i<C::XPLANE + C::xoff_of(mm)>(x8h[mm][sl], sl ? ks1 : ks0);
                        } else if constexpr (i <= RB) {
                            constexpr int j = i - 5, nn = j / 2;
                            if constexpr ((j & 1) && FP8) lds_read128i<mxo + nn * 2048 + 1024>(wa4b[nn], wp);
                            else if constexpr (j & 1) lds_read64<mxo + nn * 2048 + 1024>(wb2[nn], wp);
                            else lds_read128i<mxo + nn * 2048>(wa4[nn], wp);
                        } else if constexpr (i == RB + 1) lds_read32<0>(koA_n, nk_a);         // the next piece's tap offsets (ks0 / ks1 hold this piece's)
                        else if constexpr (i == RB + 2) lds_read32<16>(pw_koB, nk_a);
                        else if constexpr (i == RB + 3) lds_read64<0>(pw_k2, nk_a + kq * 4);
                        else if constexpr (i == RB + 12) {
                            lgkm_wait<0>();
                        } else if constexpr (i == RB + 13) form_x8(IntC<0>{});
                        else if constexpr (i == RB + 15) form_x8(IntC<1>{});
                        PW_SB;
                    });
                    // the one barrier of the piece: this wave's part of the next weight piece (issued a burst ago) and, from a slab's second piece on,
                    // of the next halo tile has landed; nobody reads this piece's weight buffer any more
                    PW_T(2);
                    if constexpr (C::NWB == 3) {
                        // the NEXT piece's weights were requested a whole piece ago; what this piece requested (WPW weight DMAs for the piece after next, and the HH halo
                        // DMAs of a slab's first two pieces, interleaved with them) may stay in flight - the memory pipe returns in order. (A slab of only two pieces
                        // must see its second halo plane before the next slab starts: no allowance there.)
                        if (!halo_now) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(WPW) : "memory");
                        else if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(WPW + HH) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    PW_T(3);
                    wg_barrier();
                    PW_T(4);
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- burst M: the MX step, voxel fragment m in the outer loop (NF MFMAs each). Behind group m: the code slots of fragment m + 2,
                    // two operand reads of the NEXT piece's first f16 chunk, the operand of fragment m + 1 formed (its slots were requested a
                    // group ago: counted wait - LDS returns in order), and (a slab's first piece) one DMA of the next halo tile
                    typedef int v2i_ __attribute__((ext_vector_type(2)));
                    mx_op wa[NF];
#pragma unroll
                    for (int n = 0; n < NF; ++n) {
                        if constexpr (FP8) wa[n] = __builtin_shufflevector(wa4[n], wa4b[n], 0, 1, 2, 3, 4, 5, 6, 7);
                        else {
                        const v2i_ b2 = __builtin_bit_cast(v2i_, wb2[n]);
                        const v4i b4 = __builtin_shufflevector(b2, b2, 0, 1, -1, -1);
                        wa[n] = __builtin_shufflevector(wa4[n], b4, 0, 1, 2, 3, 4, 5);
                        }
                        asm volatile("" : "+v"(wa[n]));
                    }
                    int wsc_lo = FP8 ? 127 - 12 : (int)wsc, wsc_hi = (int)(wsc >> 32);      // (fp8: the lo parts of the plain weight codes carry 2^12: one E8M0 scale 2^-12 for all)
                    asm volatile("s_nop 3" : "+v"(wsc_lo), "+v"(wsc_hi));      // (VALU moves that formed the operands above -> first MFMA: the hazard is not visible to hipcc)
                    const unsigned kosA = nx_a + (unsigned)koA_n;
                    // filler slots behind MFMA n of voxel-fragment group m: n = 0, 1 the code slots of fragment m + 2; then - NF >= 7 - n = 2, 3 two next-A reads and
                    // n = 4 the operand of fragment m + 1, or - NF = 5, 6 - n = 2 the operand of fragment m + 1 (two MFMAs ahead of its first use: the hazard
                    // between a VALU write and an inline-asm MFMA is ours to keep) and n = 3, 4 the next-A reads
                    static_assert(NF >= 5, "MX burst: 5 filler slots per voxel fragment");
                    constexpr int NA_N0 = NF >= 7 ? 2 : 3, FORM_N = NF >= 7 ? 4 : 2;
                    auto valid_na = [](int g) constexpr { return g < 0 ? 0 : (2 * g < MF + NF ? 1 : 0) + (2 * g + 1 < MF + NF ? 1 : 0); };
                    // next-A read r = 0 .. MF + NF - 1: weight fragment 0, the MF activation fragments, weight fragments 1 .. NF - 1
                    auto next_a = [&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        if constexpr (r == 0) lds_read128<0>(pw_wf[0], wpn);            // (straight into the carried registers: chunk 2p is long done with them)
                        else if constexpr (r <= MF) lds_read128<C::xoff_of(r - 1)>(pw_xf[r - 1], kosA);
                        else if constexpr (r < MF + NF) lds_read128<(r - MF) * 1024>(pw_wf[r - MF], wpn);
                    };
                    static_for<0, NM>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, m = i / NF, n = i % NF;
                        if constexpr (FP8) pw_mfma_mx8(acc[m][n], wa[n], x8[m], wsc_lo, mx_sb);
                        else pw_mfma_mx6<(n & 3)>(acc[m][n], wa[n], x8[m], n < 4 ? wsc_lo : wsc_hi, mx_sb);
                        if constexpr (n == 0) { if constexpr (m + 2 < MF) lds_read128i<C::XPLANE + C::xoff_of(m + 2 < MF ? m + 2 : 0)>(x8h[m + 2 < MF ? m + 2 : 0][0], ks0); }
                        else if constexpr (n == 1) { if constexpr (m + 2 < MF) lds_read128i<C::XPLANE + C::xoff_of(m + 2 < MF ? m + 2 : 0)>(x8h[m + 2 < MF ? m + 2 : 0][1], ks1); }
                        else if constexpr (n == NA_N0) next_a(IntC<2 * m>{});
                        else if constexpr (n == NA_N0 + 1) next_a(IntC<2 * m + 1>{});
                        else if constexpr (n == FORM_N) {
                            if constexpr (m + 1 < MF) {
                                // slots of fragment m + 1: read in burst B (m = 0) or behind group m - 1; younger reads: that group's next-A reads, this group's
                                // two slot reads and - where they come first - its next-A reads
                                if constexpr (m >= 1) lgkm_wait<valid_na(m - 1) + (m + 2 < MF ? 2 : 0) + (FORM_N > NA_N0 ? valid_na(m) : 0)>();
                                form_x8(IntC<(m + 1 < MF ? m + 1 : 0)>{});
                            }
                        }
                        PW_SB;
                    });
                    lgkm_wait<0>();
                    PW_T(5);
#undef PW_T
                    if constexpr (SN_TIMING == 1) { t_vm += pwt[1] - pwt[0]; t_bar += pwt[2] - pwt[1]; ++n_piece; }
                    if constexpr (SN_TIMING == 2) { t_vm += pwt[3] - pwt[2]; t_bar += pwt[4] - pwt[3]; ++n_piece; }
                    if constexpr (SN_TIMING == 3) { t_vm += pwt[5] - pwt[4]; t_bar += pwt[5] - pwt[0]; ++n_piece; }
                    wbi = C::wb_next(wbi);
                } while (++p < npiece);
                if constexpr (SN_TIMING == 4) { t_rel = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_bar += t_rel - pws1; ++n_piece; if (last_slab) t_rel = 0; }
            } else
            if constexpr (PPX) {
                // ---- PING-PONG K loop, f16 / f16x3 kernels (round 3) ---------------------------------------------------
                // Same structure as the f16m8 loop below: a segment = one K-chunk; its NPLM * (MF + NF) operand fragments are read into registers in
                // the wave's LOAD slot (with the tap offset of the next segment and the wave's DMA duties), its (SPLIT ? 3 : 1) * MF * NF MFMAs run as
                // one uninterrupted burst in its COMPUTE slot; wave group 1 runs one barrier behind group 0, so each SIMD's matrix pipe always belongs
                // to exactly one wave. DMA duties: the next weight piece and (a slab's first piece) the next slab's halo tile with the first segment of
                // a piece, the wait for the weights with the piece's last segment (the halo, younger, may stay in flight: counted wait).
                // Measured and dropped (profiles/r3, r4 README): halo DMAs dealt out over the load slots or issued inside the burst, two chunks per
                // segment, a scheduling barrier between operand reads and DMA duties.
                static_assert(C::PCH >= 2 && BUFH && SPLIT < 2 && C::PTAB, "ping-pong loop (f16 / f16x3): at least two chunks per weight piece");
                const unsigned koff_a = kbuf_a + (unsigned)((xb * 8 + (last_slab ? 4 : 0) + (slab & 3)) * (C::KOFF_N * 4));
                const unsigned xslab = xbuf_a + xb * C::XBUF;
                constexpr int NPLM = C::NPLM;
                int ko, ko_n = 0;
                const int bridge_b = su_b;                          // bridge chunks (write_koff_part): units this slab's last chunk takes from the next slab
                lds_read32<0>(ko, koff_a);
                lgkm_wait<0>();
                int p = 0;
                do {
                    const int ch0 = p * C::PCH;
                    const unsigned wp = wbuf_a + wbi * C::WBUF;
                    const int nseg = (nchunk - ch0) < C::PCH ? (nchunk - ch0) : C::PCH;      // chunks (= segments) of this piece (>= 1)
                    int hnow = 0;
                    static_for<0, C::PCH>([&](auto scc) {
                        constexpr int sc = decltype(scc)::value;
                        if (sc < nseg) {
                            half8 xf[NPLM][MF], wf[NPLM][NF];
                            {
                                const unsigned kos = xslab + (unsigned)ko;
                                // XIMM: under the row-gap maps of the 3-D kernels the distance of fragment m from fragment 0 is a compile-time constant
                                // (ConvCfg::xoff_of): one address register + immediate offsets instead of an add per read in the load slot
                                constexpr bool XIMM = K2D == 0 && !PMAP && KS == 3 && C::XPLANE + C::xoff_of(MF - 1) < 65536;
                                const unsigned kos0 = (unsigned)xbase[0] + kos;
                                static_for<0, MF>([&](auto mc) {
                                    constexpr int m = decltype(mc)::value;
                                    if constexpr (XIMM) {
                                        lds_read128<C::xoff_of(m)>(xf[0][m], kos0);
                                        if constexpr (SPLIT == 1) lds_read128<C::XPLANE + C::xoff_of(m)>(xf[1][m], kos0);
                                    } else {
                                        lds_read128<0>(xf[0][m], (unsigned)xbase[m] + kos);
                                        if constexpr (SPLIT == 1) {
                                            if constexpr (C::XPLANE < 65536) lds_read128<(C::XPLANE < 65536 ? C::XPLANE : 0)>(xf[1][m], (unsigned)xbase[m] + kos);
                                            else lds_read128<0>(xf[1][m], (unsigned)xbase[m] + kos + C::XPLANE);
                                        }
                                    }
                                });
                                static_for<0, NF>([&](auto nc) {
                                    constexpr int n = decltype(nc)::value;
                                    lds_read128<(sc * NF + n) * C::MFRAG>(wf[0][n], wp);
                                    if constexpr (SPLIT == 1) lds_read128<(sc * NF + n) * C::MFRAG + 1024>(wf[1][n], wp);
                                });
                                lds_read32<0>(ko_n, koff_a + (unsigned)(ch0 + sc + 1) * 16);      // tap offset of the next segment's chunk
                            }
                            if constexpr (sc == 0) {
                                // the piece after this one: next piece of the slab (possibly short), else the first piece of the next slab / tile.
                                // Branch-free issue: always the compile-time maximum per wave; an index beyond the piece repeats its last item (same
                                // bytes, same place), and behind the layer's last piece its first one is fetched into the idle buffer
                                const bool real = (p + 1 < npiece) || have_next;
                                const size_t w_off = (p + 1 < npiece) ? woff + (size_t)(ch0 + C::PCH) * NF * C::FRAG : nwoff;
                                int w_nch = C::PCH;
                                if (p + 1 < npiece) { const int rem = wchunk - (ch0 + C::PCH); if (rem < C::PCH) w_nch = rem; }
                                else { if (n_wchunk < C::PCH) w_nch = n_wchunk; }
                                const char *src = wsrc0 + (real ? w_off : 0);
                                char *dst = wbuf + (wbi ^ 1) * C::WBUF;
                                const int cnt = (real ? w_nch : (wchunk0 < C::PCH ? wchunk0 : C::PCH)) * NF * NPL;
                                constexpr int WPWX = (C::PCH * NF * NPL + C::NW - 1) / C::NW;
                                static_for<0, WPWX>([&](auto kc) {
                                    int i = decltype(kc)::value * C::NW + wave;
                                    i = i < cnt ? i : cnt - 1;
                                    dma16(src + (size_t)i * 1024 + lane * 16, dst + i * 1024);
                                });
                                // the next slab's halo tile, behind the weight DMAs
                                if (p == 0 && have_next)
                                    hnow = stage_halo_buf(last_slab ? nxt_b : (K2D ? x0 : b), last_slab ? nxt_keep : cur_keep, last_slab ? nxt_toff : cur_toff, nc0, nc8n, xb ^ 1);
                            }
                            // (bridge chunk next: it reads the NEXT slab's halo tile - everything this wave has in flight must have landed before the barrier)
                            const bool pre_bridge = bridge_b > 0 && ch0 + sc == nchunk - 2;
                            if (sc == nseg - 1) {
                                // the next weight piece has landed; halo DMAs issued in this piece's first load segment may still fly unless the slab ends here
                                if (!pre_bridge && p + 1 < npiece && hnow == HT) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(HT) : "memory");
                                else if (!pre_bridge && p + 1 < npiece && HT > 1 && hnow == HT - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(HT > 1 ? HT - 1 : 0) : "memory");
                                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            } else if (pre_bridge) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            lgkm_wait<0>();
                            ko = ko_n;
                            wg_barrier();
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int n = 0; n < NF; ++n) {
                                if constexpr (SPLIT == 1) {
#pragma unroll
                                    for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][n], xf[0][m], acc[m][n], 0, 0, 0);
#pragma unroll
                                    for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][n], xf[1][m], acc[m][n], 0, 0, 0);
                                }
#pragma unroll
                                for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][n], xf[0][m], acc[m][n], 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            wg_barrier();
                        }
                    });
                    wbi ^= 1;
                } while (++p < npiece);
            } else
            if constexpr (PPM) {
                // ---- PING-PONG K loop, f16m8 kernels launched as eight-wave workgroups (round 3; since round 4 only the all-MX mode's 3x3x3 layers) --
                // The two waves of a SIMD (wave w of group 0 = waves 0..3 and wave w + 4 of group 1) never compete for the matrix pipe: a
                // SEGMENT (one K-chunk of f16 MFMAs, or one MX step) is LOADED - every operand fragment of the segment read from LDS into
                // registers, plus this wave's share of the DMA issue - and then COMPUTED as one uninterrupted burst of MF*NF MFMAs whose operands are
                // all in registers; a workgroup barrier separates the two, and group 1 runs exactly one barrier behind group 0, so on every SIMD
                // one wave computes while its partner loads. Nothing is software-pipelined inside a wave: the latency of the LDS reads, the
                // address arithmetic, the DMA issue, the register shuffles of the 6-bit operands all sit in the load segment. Buffer recycling
                // needs no barrier of its own: the last reader of a weight piece / halo buffer (group 1, loading the piece's last segment) has
                // waited for its reads before the barrier that precedes the first load slot of the next piece, where the refill DMAs are issued.
                // Measured and dropped again (DESIGN.md sections 4.2 / 8; profiles/r3/README.md): one designated DMA wave per slot, halo DMAs spread
                // over the pieces, the closing barrier in front of the burst's last MFMAs, both f16 chunks loaded in one slot, MX weights read with the
                // f16 chunks' loads, 96-bit reads of the code slots, the next segment's weight fragments requested from inside the burst.
                static_assert(SPLIT == 2 && C::PCH == 2 && SN_MX_FMT != 0 && BUFH, "ping-pong loop: f16m8 kernels with 6-bit MX operands");
                const unsigned koff_a = kbuf_a + xb * (C::KOFF_N * 4);
                const unsigned k2_a = koff_a + kq * 4;
                const unsigned xslab = xbuf_a + xb * C::XBUF;       // (wave-uniform; the per-lane fragment offsets xbase[] stay the only address registers)
                int koA, koB;          // tap offsets of this lane quarter's group in f16 chunks 2p, 2p+1
                long long k2;          // ... of its two groups 8p + 2kq, + 1 in the MX step
                lds_read32<0>(koA, koff_a);
                lds_read32<16>(koB, koff_a);
                lds_read64<0>(k2, k2_a);
                lgkm_wait<0>();
                constexpr int WCNT = C::PCH * NF * NPL;                     // 1 KiB DMAs per weight piece
                constexpr int WPW = (WCNT + C::NW - 1) / C::NW;             // ... per wave
                // items [k0, k1) of this wave's share of the piece at byte offset `off` of the weight stream, into weight buffer wb (branch-free: a wave
                // without an item of its own repeats the piece's last one - same bytes, same place)
                auto stage_w_part = [&](size_t off, int wb, int k0, int k1) {
                    const char *src = wsrc0 + off;
                    char *dst = wbuf + wb * C::WBUF;
                    for (int k = k0; k < k1; ++k) {
                        int i = k * C::NW + wave;
                        i = i < WCNT ? i : WCNT - 1;
                        dma16(src + (size_t)i * 1024 + lane * 16, dst + i * 1024);
                    }
                };
                int p = 0;
                do {             // (do-while: a possible zero-trip path made hipcc spill 88 accumulator registers around the loop)
                    const int ch0 = p * C::PCH;
                    const unsigned wp = wbuf_a + wbi * C::WBUF;
                    // the piece after this one: next piece of the slab, else the first piece of the next slab / tile (behind the last piece of all, piece 0
                    // of the layer is fetched into the idle buffer)
                    const size_t w_off = (p + 1 < npiece) ? woff + (size_t)(ch0 + C::PCH) * NF * C::FRAG : (!have_next ? 0 : nwoff);
                    const bool has_B = ch0 + 1 < nchunk;             // (the last piece of a slab with an odd chunk count has no second f16 chunk)
                    int hnow = 0;
                    half8 xf[2][MF], wf[2][NF];                     // operands of the f16 chunks 2p, 2p+1
                    constexpr int mxo = 2 * NF * 1024;
                    v4i wa4[NF];                                    // MX step: a lane's 192-bit weight operand = 128 + 64 bits ...
                    long long wb2[NF], wsc;                         // ... and the E8M0 block scales of its NF fragments
                    typedef int v2i_ __attribute__((ext_vector_type(2)));
                    using I0 = std::integral_constant<int, 0>;
                    using I1 = std::integral_constant<int, 1>;
                    auto load_f16 = [&](auto ccc, int ko) {           // the activation and weight fragments of f16 chunk cc
                        constexpr int cc = decltype(ccc)::value;
                        const unsigned kos = xslab + (unsigned)ko;
                        static_for<0, MF>([&](auto mc) { constexpr int m = decltype(mc)::value; lds_read128<0>(xf[cc][m], (unsigned)xbase[m] + kos); });
                        static_for<0, NF>([&](auto nc) { constexpr int n = decltype(nc)::value; lds_read128<(cc * NF + n) * 1024>(wf[cc][n], wp); });
                    };
                    auto compute_f16 = [&](auto ccc) {
                        constexpr int cc = decltype(ccc)::value;
                        static_for<0, NF>([&](auto nc) {
                            constexpr int n = decltype(nc)::value;
#pragma unroll
                            for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cc][n], xf[cc][m], acc[m][n], 0, 0, 0);
                        });
                    };
                    constexpr int WSPLIT = (WPW + 1) / 2;    // weight DMA instalments [0, WSPLIT) with chunk 2p, [WSPLIT, WPW) with chunk 2p+1
                    // ---- segment A: f16 chunk 2p; DMA: first part of the next weight piece; group 0 also writes the next slab's tap table
                    load_f16(I0{}, koA);
                    stage_w_part(w_off, wbi ^ 1, 0, WSPLIT);
                    if (p == 0 && have_next && wave < C::NW / 2) write_koff_part(nc8n, xb ^ 1, nslab_i, tid, C::NT / 2);
                    lgkm_wait<0>();
                    wg_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    compute_f16(I0{});
                    __builtin_amdgcn_sched_barrier(0);
                    wg_barrier();
                    // ---- segment B: f16 chunk 2p+1; DMA: the rest of the next weight piece
                    if (has_B) {
                        load_f16(I1{}, koB);
                        stage_w_part(w_off, wbi ^ 1, WSPLIT, WPW);
                        lgkm_wait<0>();
                        wg_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                        compute_f16(I1{});
                        __builtin_amdgcn_sched_barrier(0);
                        wg_barrier();
                    } else {
                        stage_w_part(w_off, wbi ^ 1, WSPLIT, WPW);
                    }
                    // ---- segment M: the MX step of the piece's 64 k; DMA: the next slab's halo tile (first piece of a slab); waits for the weights
                    {
                        v4i x8h[MF][2];                                       // whole code slots (4 LDS cycles per read instead of 8 for 96 bits; the pad dword is dropped below)
                        int koAn = 0, koBn = 0;
                        long long k2n = 0;
                        const unsigned ks0 = xslab + C::XPLANE + (unsigned)(int)k2, ks1 = xslab + C::XPLANE + (unsigned)(int)(k2 >> 32);
                        static_for<0, MF>([&](auto mc) {
                            constexpr int m = decltype(mc)::value;
                            lds_read128i<0>(x8h[m][0], (unsigned)xbase[m] + ks0);
                            lds_read128i<0>(x8h[m][1], (unsigned)xbase[m] + ks1);
                        });
                        lds_read64<mxo + 1024 + 8>(wsc, wp);                  // the MX weight fragments and their scales
                        static_for<0, NF>([&](auto nc) {
                            constexpr int n = decltype(nc)::value;
                            lds_read128i<mxo + n * 2048>(wa4[n], wp);
                            lds_read64<mxo + n * 2048 + 1024>(wb2[n], wp);
                        });
                        if (p + 1 < npiece) {
                            lds_read32<0>(koAn, koff_a + (unsigned)(ch0 + 2) * 16);
                            lds_read32<0>(koBn, koff_a + (unsigned)(ch0 + 3) * 16);
                            lds_read64<0>(k2n, k2_a + (unsigned)(8 * (p + 1)) * 4);
                        }
                        if (p == 0 && have_next)
                            hnow = stage_halo_buf(last_slab ? nxt_b : (K2D ? x0 : b), last_slab ? nxt_keep : cur_keep, last_slab ? nxt_toff : cur_toff, nc0, nc8n, xb ^ 1);
                        // the next weight piece (issued two load slots ago) has landed; this slab's halo DMAs, just issued, may still fly
                        if (p + 1 < npiece && hnow == HT) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(HT) : "memory");
                        else if (p + 1 < npiece && HT > 1 && hnow == HT - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(HT > 1 ? HT - 1 : 0) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        lgkm_wait<0>();
                        v8i x8[MF], wa[NF];
#pragma unroll
                        for (int m = 0; m < MF; ++m) {
                            asm volatile("" : "+v"(x8h[m][0]), "+v"(x8h[m][1]));
                            x8[m] = __builtin_shufflevector(x8h[m][0], x8h[m][1], 0, 1, 2, 4, 5, 6, -1, -1);
                        }
#pragma unroll
                        for (int n = 0; n < NF; ++n) {
                            const v2i_ b2 = __builtin_bit_cast(v2i_, wb2[n]);
                            const v4i b4 = __builtin_shufflevector(b2, b2, 0, 1, -1, -1);
                            wa[n] = __builtin_shufflevector(wa4[n], b4, 0, 1, 2, 3, 4, 5, -1, -1);
                        }
                        if (p + 1 < npiece) { koA = koAn; koB = koBn; k2 = k2n; }
                        // (pins the operand tuples - and the register moves per activation fragment that forming them costs - in front of the barrier,
                        // i.e. into the load segment: instruction selection otherwise sinks them to their first use, the head of the MFMA burst)
#pragma unroll
                        for (int m = 0; m < MF; ++m) asm volatile("" : "+v"(x8[m]));
                        __builtin_amdgcn_sched_barrier(0);
                        wg_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                        static_for<0, NF>([&](auto nc) {
                            constexpr int n = decltype(nc)::value;
                            const int sa = n < 4 ? (int)wsc : (int)(wsc >> 32);
#pragma unroll
                            for (int m = 0; m < MF; ++m)
                                acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa[n], x8[m], acc[m][n], SN_MX_FMT, SN_MX_FMT, n & 3, sa, 0, mx_sb);
                        });
                        __builtin_amdgcn_sched_barrier(0);
                        wg_barrier();
                    }
                    wbi ^= 1;
                } while (++p < npiece);
            } else {
            // ---- software-pipelined K loop (rounds 1-2) -------------------------------------------------------------
            // Still runs the kernels the newer loops do not cover: the 1x1x1 side convolutions (4-wave workgroups, every precision mode) and the 2-D
            // nets' one-plane f16 mode (generic halo path). One barrier per weight piece; inside a piece the loop is pipelined by hand:
            // X fragments of chunk c+1 and the tap offset of chunk c+2 are fetched while chunk c computes (the halo buffer is immutable for the
            // whole slab, so this runs across the per-piece barrier); weight fragment n+1 (or fragment 0 of the next chunk of the piece) is
            // fetched while fragment n's MFMAs issue. Every wait counts only the reads issued AFTER the one waited for (LDS returns in order).
            // (Its round-2 refinements for the 3x3x3 f16m8 kernels - the barrier in front of a piece's last two MFMA groups, prefetch reads spread
            // over the chunk, weight fragments two groups ahead - went with those kernels' move to the ping-pong loops: profiles/r2/README.md.)
            static_assert((KS == 1 && SPLIT <= 2) || (K2D != 0 && SPLIT == 0), "legacy K loop: 1x1x1 layers and the 2-D one-plane f16 mode");
            const unsigned koff_a = kbuf_a + xb * (C::KOFF_N * 4);
            unsigned xaddr[MF];
#pragma unroll
            for (int m = 0; m < MF; ++m) xaddr[m] = xbuf_a + xb * C::XBUF + (unsigned)xbase[m];
            constexpr int NPLM = C::NPLM;
            half8 xc[NPLM][MF], xn[NPLM][MF], wr[2][NPLM];
            int ko1, ko2;
            auto issue_x = [&](half8(&dst)[NPLM][MF], int ko) {
                static_for<0, MF>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    const unsigned ad = xaddr[m] + (unsigned)ko;
                    lds_read128<0>(dst[0][m], ad);
                    if constexpr (SPLIT == 1) {
                        if constexpr (C::XPLANE < 65536) lds_read128<(C::XPLANE < 65536 ? C::XPLANE : 0)>(dst[1][m], ad);
                        else lds_read128<0>(dst[1][m], ad + C::XPLANE);
                    }
                });
            };
            long long k2, k2n;                                                // f16m8: tap offsets of this lane's 2 groups (8p + 2kq, +1) in the MX step
            const unsigned k2_a = koff_a + kq * 4;
            {
                int k0;
                lds_read32<0>(k0, koff_a);
                lds_read32<16>(ko1, koff_a);
                if constexpr (SPLIT == 2) lds_read64<0>(k2, k2_a);
                lgkm_wait<0>();
                issue_x(xc, k0);
                lgkm_wait<0>();
            }
            for (int p = 0; p < npiece; ++p) {
                const int ch0 = p * C::PCH;
                const unsigned wp = wbuf_a + wbi * C::WBUF;
                // f16m8: 6-bit operands of this piece's MX step (a slot's 12 code bytes per read); fetched behind the first MFMA group of chunk 0
                v8i x8[SPLIT == 2 ? MF : 1];
                v3i x6[SPLIT == 2 ? MF : 1][2];
                // first weight fragment of this piece: its LDS read goes out before the (VALU-heavy) DMA address work
                lds_read128<0>(wr[0][0], wp);
                if constexpr (SPLIT == 1) lds_read128<1024>(wr[0][1], wp);
                // next weight piece (the following piece of this slab, else the first piece of what comes next) and, behind it, the next slab's halo tile
                int hnow = 0;
                if (p + 1 < npiece) {
                    const int rem = wchunk - (ch0 + C::PCH);
                    stage_w(woff + (size_t)(ch0 + C::PCH) * NF * C::FRAG, rem < C::PCH ? rem : C::PCH, wbi ^ 1);
                } else if (have_next) {
                    stage_w(nwoff, n_wchunk < C::PCH ? n_wchunk : C::PCH, wbi ^ 1);
                }
                if constexpr (BUFH) {
                    if (have_next && p == 0)
                        hnow = stage_halo_buf(last_slab ? nxt_b : (K2D ? x0 : b), last_slab ? nxt_keep : cur_keep, last_slab ? nxt_toff : cur_toff, nc0, nc8n, xb ^ 1);
                } else if (have_next && hdone < HT && (p + 1 < npiece || npiece == 1)) {
                    const int left = HT - hdone;
                    const int kn = (p + 2 >= npiece || left < HQ) ? left : HQ;
                    hnow = stage_halo(ntile, nc0, nc8n, xb ^ 1, hdone, kn);
                    hdone += kn;
                }
                static_for<0, C::PCH>([&](auto ccc) {
                    constexpr int cc = decltype(ccc)::value;
                    const int ch = ch0 + cc;
                    if (ch < nchunk) {
                        constexpr int par0 = (cc * NF) & 1;
                        static_for<0, NF>([&](auto nc) {
                            constexpr int n = decltype(nc)::value;
                            constexpr int G = cc * NF + n, GT = C::PCH * NF;                     // group index inside the piece / groups of a full piece
                            constexpr int cur = (par0 + n) & 1, nxt = cur ^ 1;
                            constexpr bool more_n = (n + 1 < NF), more_c = (cc + 1 < C::PCH);
                            // reads issued after this group's fragment and allowed to stay in flight while it is waited for: the fragment of the next
                            // group, plus (E) the tap offset / activation fragments [/ MX operands] that group 0 of the chunk issues behind its MFMAs
                            constexpr int E = 1 + MF * NPLM + ((SPLIT == 2 && cc == 0) ? 2 * MF + 1 : 0);
                            constexpr int EQ = sn_e_after(n - 1, E, E);                          // E reads younger than the awaited fragment
                            constexpr int Gn = more_n ? G + 1 : (cc + 1) * NF;                   // the group whose fragment is requested now
                            if constexpr (Gn < GT) {
                                lds_read128<Gn * C::MFRAG>(wr[nxt][0], wp);
                                if constexpr (SPLIT == 1) lds_read128<Gn * C::MFRAG + 1024>(wr[nxt][1], wp);
                            }
                            lgkm_wait<((more_n || more_c) ? NPLM : 0) + EQ>();
                            const half8 &w0 = wr[cur][0];
                            if constexpr (SPLIT == 1) {
#pragma unroll
                                for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[cur][1], xc[0][m], acc[m][n], 0, 0, 0);
#pragma unroll
                                for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xc[1][m], acc[m][n], 0, 0, 0);
                            }
#pragma unroll
                            for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xc[0][m], acc[m][n], 0, 0, 0);
                            if constexpr (n == 0) {
                                lds_read32<0>(ko2, koff_a + (unsigned)(ch + 2) * 16);
                                issue_x(xn, ko1);
                                if constexpr (SPLIT == 2 && cc == 0) {
                                    // operand fetch for the MX step that closes this chunk (its tap offsets k2 were read one piece ahead); also the next piece's tap offsets
                                    static_for<0, MF>([&](auto mc) {
                                        constexpr int m = decltype(mc)::value;
                                        const unsigned ad = xaddr[m] + C::XPLANE;               // the 12 code bytes of two slots = the lane's 192-bit operand
                                        lds_read96i<0>(x6[m][0], ad + (unsigned)(int)k2);
                                        lds_read96i<0>(x6[m][1], ad + (unsigned)(int)(k2 >> 32));
                                    });
                                    lds_read64<0>(k2n, k2_a + (unsigned)(8 * (p + 1)) * 4);
                                }
                            }
                        });
                        // X(c+1), koff(c+2) [, MX operands] landed (the reads E of group 0); the next chunk's first weight fragment, issued AFTER E, may still be in flight
                        constexpr int young = (cc + 1 < C::PCH && NF >= 2) ? 1 : 0;
                        lgkm_wait<NPLM * young>();
#pragma unroll
                        for (int m = 0; m < MF; ++m) {
                            xc[0][m] = xn[0][m];
                            if constexpr (SPLIT == 1) xc[1][m] = xn[1][m];
                        }
                        ko1 = ko2;
                    }
                    if constexpr (SPLIT == 2 && cc == 0) {
                        // MX step (placed between the piece's two f16 chunks so its operands are short-lived): both correction terms of this piece's 64 k
                        // in one MX-scaled MFMA per (cout, voxel) fragment pair
                        static_assert(SN_MX_FMT != 0 && NF <= 8, "the 6-bit MX forms use the two-slot operand layout");
                        constexpr int mxo = 2 * NF * 1024;
#pragma unroll
                        for (int m = 0; m < MF; ++m) {
                            // the second slot's three dwords cannot be read in place (a register tuple starts on an even register): 3 v_mov's per fragment.
                            // (the empty asm pins the copies BEHIND the chunk-end wait: to the compiler the registers were complete when the reads were issued)
                            asm volatile("" : "+v"(x6[m][0]), "+v"(x6[m][1]));
                            x8[m] = __builtin_shufflevector(x6[m][0], x6[m][1], 0, 1, 2, 3, 4, 5, -1, -1);
                        }
                        k2 = k2n;
                        // the weight fragment of a lane is its 192-bit operand, dwords 0..3 in the first lane-linear KiB of the fragment and 4..5 in the second: a
                        // 128-bit and a 64-bit read into 6 consecutive registers. The E8M0 block scales of the lane's NF fragments are the 8 bytes behind
                        // dwords 4..5 of fragment 0: one 64-bit read per piece, the byte is picked by the instruction's op_sel (pack_conv).
                        v4i wa4[2];
                        long long wb2[2], wsc;
                        lds_read64<mxo + 1024 + 8>(wsc, wp);
                        lds_read128i<mxo>(wa4[0], wp);
                        lds_read64<mxo + 1024>(wb2[0], wp);
                        static_for<0, NF>([&](auto nc) {
                            constexpr int n = decltype(nc)::value;
                            constexpr int cur = n & 1, nxt = cur ^ 1;
                            if constexpr (n + 1 < NF) {
                                lds_read128i<mxo + (n + 1) * 2048>(wa4[nxt], wp);
                                lds_read64<mxo + (n + 1) * 2048 + 1024>(wb2[nxt], wp);
                            }
                            lgkm_wait<(n + 1 < NF) ? 2 : 0>();
                            v8i wa;
                            wa[0] = wa4[cur][0]; wa[1] = wa4[cur][1]; wa[2] = wa4[cur][2]; wa[3] = wa4[cur][3];
                            wa[4] = (int)wb2[cur]; wa[5] = (int)(wb2[cur] >> 32); wa[6] = 0; wa[7] = 0;
                            const int sa = n < 4 ? (int)wsc : (int)(wsc >> 32);
#pragma unroll
                            for (int m = 0; m < MF; ++m)
                                acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, x8[m], acc[m][n], SN_MX_FMT, SN_MX_FMT, n & 3, sa, 0, mx_sb);
                        });
                    }
                });
                lgkm_wait<0>();
                // next weight piece landed; the newest HQ halo DMAs (issued after it) may still fly
                if (p + 1 != npiece && hnow >= HQ) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(HQ) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wg_barrier();                             // ... for every wave; this piece's buffers are free again
                wbi ^= 1;
            }
            }
            xb ^= 1;
            c0 += c8n;
            woff += (size_t)wchunk * NF * C::FRAG;
            sl_o = n_o;
            if constexpr (PWM) pw_hptr += VOL * 16;
        }

        if constexpr (PP) { if (wave < C::NW / 2) wg_barrier(); }     // pairs with group 1's last compute segment of the tile
        if constexpr (PWM) {      // the loop's MFMAs are inline asm: the wait states between the last of them and the epilogue's accumulator reads, by hand
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < NF; ++n) asm volatile("" : "+a"(acc[m][n]));
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        }
        const long long t_tile1 = SN_TIMING == 10 ? __builtin_readcyclecounter() : 0;
        if constexpr (SN_TIMING == 10) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- epilogue: folded BN affine + activation --------------------------------------------------------
        if constexpr (EPI == EPI_POOL2D) {
            // conv + bias + ReLU, then Pool2DLayer(2) in registers: the 2x2 pixel quad of an output lives in lanes {l, l^1} (columns)
            // x {l, l^8} (rows; l^4 for the 4x4-image fragments). max(split(y)) == split(max(y)) bit for bit because the
            // hi/lo rounding is monotone, so this equals storing the map and max-pooling the stored values. Output: [c/8][DX][D/2][D/2][8].
            static_assert(K2D != 0 && OSPLIT <= 2 && SPLIT <= 2, "EPI_POOL2D is a 2-D epilogue (formats 0..2)");
            const int Do = D >> 1;
            const size_t VOLo = (size_t)DX * Do * Do;
            constexpr int YX = C::F4 ? 4 : 8;                       // lane distance of the row partner
            // R2: the two rows of a fragment are not adjacent (conflict-free LDS reads, kRowGap2D): the row partner of a pixel is the SAME lane of
            // fragment m ^ 1 (rows r and r + 1), every lane row of fragment m = 0, 2 writes one pooled row
            constexpr bool R2 = (K2D == 1 && C::VS == 32);   // (64-byte pixels, f16 mode: adjacent rows are already 640 bytes apart)
            f32x4 scv[NF], shv[NF];           // all fragments' constants before the first store (see EPI_STORE)
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                const int nl = (blockIdx.y * NF + n) * 16 + kq * 4;
                scv[n] = *reinterpret_cast<const f32x4 *>(a.scale + nl);
                shv[n] = *reinterpret_cast<const f32x4 *>(a.shift + nl);
            }
#pragma unroll
            for (int m = 0; m < MF; m += (R2 ? 2 : 1)) {
                int hx_, hy_, hz_;
                frag_xyz(m, hx_, hy_, hz_);
                const int gx = x0 + hx_, gy = y0 + hy_, gz = z0 + hz_;
                const bool writer = !(v & 1) && (R2 || !(v & YX)) && gx < DX && gy < D && gz < D;     // D is even: the whole quad is inside
                const size_t vlin = ((size_t)gx * Do + (gy >> 1)) * Do + (gz >> 1);
#pragma unroll
                for (int n = 0; n < NF; ++n) {
                    const int nl = (blockIdx.y * NF + n) * 16 + kq * 4;
                    const f32x4 sc = scv[n], sh = shv[n];
                    half4 h, l;
                    float lo32[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t0 = acc[m][n][r] * sc[r] + sh[r], t1 = R2 ? acc[m + 1 < MF ? m + 1 : m][n][r] * sc[r] + sh[r] : t0;
                        float y = fmaxf(t0, 0.f);
                        if constexpr (R2) y = fmaxf(y, fmaxf(t1, 0.f));
                        y = fmaxf(y, __shfl_xor(y, 1));
                        if constexpr (!R2) y = fmaxf(y, __shfl_xor(y, YX));
                        if constexpr (OSPLIT == 1) {
                            _Float16 hh, ll;
                            sn_split(y, hh, ll);
                            h[r] = hh; l[r] = ll;
                        } else {
                            h[r] = (_Float16)y;
                            lo32[r] = (y - (float)h[r]) * kMxLoMul;
                        }
                    }
                    sn_track_acc(trk_acc, acc[m][n]);
                    if constexpr (R2) sn_track_acc(trk_acc, acc[m + 1 < MF ? m + 1 : m][n]);
                    { const uint2 hb = __builtin_bit_cast(uint2, h); sn_track_h2(trk_h, hb.x); sn_track_h2(trk_h, hb.y); }
                    if (writer && nl < a.out_cp) {
                        const int ch = a.out_coff + nl;
                        _Float16 *o = a.out + (size_t)b * VOLo * a.out_cs + ((size_t)(ch >> 3) * VOLo + vlin) * 8 + (ch & 7);
                        *reinterpret_cast<half4 *>(o) = h;
                        if constexpr (OSPLIT == 1) *reinterpret_cast<half4 *>(o + a.out_lo_off) = l;
                        if constexpr (OSPLIT == 2) {
                            char *slot = reinterpret_cast<char *>(o - (ch & 7) + a.out_lo_off);
                            const _Float16 hq[4] = {h[0], h[1], h[2], h[3]};
                            sn_mx6_store_unit(slot, (ch & 7) >> 2, hq, lo32, a.mx_out_e8);
                        }
                    }
                }
            }
        } else if constexpr (EPI == EPI_SIDEPOOL) {
            static_assert(NF * 16 <= 96 && OSPLIT <= 2 && SPLIT <= 2, "side conv K chunks (formats 0..2)");
            const int Do = D >> 1;
            const size_t VOLo = (size_t)Do * Do * Do;
            // this layer's outputs, in registers: y[m][n][r] = ReLU(BN(acc)), fp32
            float y[MF][NF][4];
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                const int nl = n * 16 + kq * 4;
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.scale + nl);
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.shift + nl);
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pre = acc[m][n][r] * sc[r] + sh[r], t = fmaxf(pre, 0.f);
                        y[m][n][r] = t;
                    }
#pragma unroll
                for (int m = 0; m < MF; ++m) sn_track_acc(trk_acc, acc[m][n]);     // (a value beyond fp16 shows up as +inf in the pooled store / the side conv's operand)
            }
            // ---- side conv (1x1x1 over all NF*16 channels) on the matrix cores: B operand = the outputs as this lane holds them
            f32x4 sacc[MF];
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                sacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    half8 bh, bl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int n = 2 * q + (j >> 2);
                        const float t = n < NF ? y[m][n < NF ? n : 0][j & 3] : 0.f;
                        bh[j] = (_Float16)t;
                        bl[j] = (_Float16)(t - (float)bh[j]);
                    }
                    if constexpr (SPLIT != 0) {
                        sacc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sw[q][1], bh, sacc[m], 0, 0, 0);
                        sacc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sw[q][0], bl, sacc[m], 0, 0, 0);
                    }
                    sacc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sw[q][0], bh, sacc[m], 0, 0, 0);
                }
            }
            // ---- side output: BN + activation, fragment pairs exchanged so that every lane stores one 16-byte group (as EPI_STORE)
            {
                const bool odd = kq & 1;
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const f32x4 ssc = *reinterpret_cast<const f32x4 *>(a.side_scale + kq * 4);
                const f32x4 ssh = *reinterpret_cast<const f32x4 *>(a.side_shift + kq * 4);
#pragma unroll
                for (int mp = 0; mp < MF; mp += 2) {
                    bool valid[2];
                    size_t vlin[2];
                    unsigned hw[2][2], lw[2][2];
                    _Float16 hq[2][4];
                    float loq[2][4];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        int hx, hy, hz;
                        frag_xyz(mp + e, hx, hy, hz);
                        const int gx = x0 + hx, gy = y0 + hy, gz = z0 + hz;
                        valid[e] = gx < DX && gy < D && gz < D;
                        vlin[e] = ((size_t)gx * D + gy) * D + gz;
                        half4 h, l;
                        float lo32[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pre = sacc[mp + e][r] * ssc[r] + ssh[r];
                            float t = a.side_act == 0 ? fmaxf(pre, 0.f) : sn_sigmoid(pre);
                            if constexpr (OSPLIT == 1) {
                                _Float16 hh, ll;
                                sn_split(t, hh, ll);
                                h[r] = hh; l[r] = ll;
                            } else {
                                h[r] = (_Float16)t;
                                lo32[r] = (t - (float)h[r]) * kMxLoMul;
                            }
                        }
                        const uint2 hb = __builtin_bit_cast(uint2, h);
                        hw[e][0] = hb.x; hw[e][1] = hb.y;
                        sn_track_acc(trk_acc, sacc[mp + e]);
                        sn_track_h2(trk_h, hb.x); sn_track_h2(trk_h, hb.y);
                        if constexpr (OSPLIT == 1) {
                            const uint2 lb = __builtin_bit_cast(uint2, l);
                            lw[e][0] = lb.x; lw[e][1] = lb.y;
                        } else if constexpr (OSPLIT == 2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { hq[e][r] = h[r]; loq[e][r] = lo32[r]; }
                        }
                    }
                    if constexpr (OSPLIT == 2) sn_mx6_units(hq[0], loq[0], hq[1], loq[1], a.mx_side_e8, lw);
                    const bool st = odd ? valid[1] : valid[0];
                    const size_t my_vlin = odd ? vlin[1] : vlin[0];
                    const auto h0 = __builtin_amdgcn_permlane16_swap(hw[0][0], hw[1][0], false, false);
                    const auto h1 = __builtin_amdgcn_permlane16_swap(hw[0][1], hw[1][1], false, false);
                    const int ch = a.side_coff + ((kq * 4) & ~7);
                    _Float16 *o = a.side_out + (size_t)b * VOL * a.side_cs + ((size_t)(ch >> 3) * VOL + my_vlin) * 8;
                    if (st) *reinterpret_cast<u32x4 *>(o) = u32x4{h0[0], h1[0], h0[1], h1[1]};
                    if constexpr (OSPLIT == 1) {
                        const auto l0 = __builtin_amdgcn_permlane16_swap(lw[0][0], lw[1][0], false, false);
                        const auto l1 = __builtin_amdgcn_permlane16_swap(lw[0][1], lw[1][1], false, false);
                        if (st) *reinterpret_cast<u32x4 *>(o + a.side_lo_off) = u32x4{l0[0], l1[0], l0[1], l1[1]};
                    } else if constexpr (OSPLIT == 2) {
                        const auto l0 = __builtin_amdgcn_permlane16_swap(lw[0][0], lw[1][0], false, false);
                        const auto l1 = __builtin_amdgcn_permlane16_swap(lw[0][1], lw[1][1], false, false);
                        // lw[e] = {low 32, high 16 bits} of the 48-bit unit of fragment e; [0]: channels 0..3, [1]: 4..7
                        unsigned d[3];
                        sn_mx6_join(l0[0], l1[0], l0[1], l1[1], d);
                        if (st) *reinterpret_cast<u32x4 *>(o + a.side_lo_off) = u32x4{d[0], d[1], d[2], 0u};
                    }
                }
            }
            // ---- 2x2x2 max-pool in registers: x partner = fragment mm+2, y partner = lane ^ 8, z partner = lane ^ 1; stored in the layer's
            // own format (max(split(y)) == split(max(y)): the hi/lo rounding is monotone, so this equals pooling the stored tensor)
            // A fragment's rows lie 4 apart (conflict-free LDS reads, lds_probe): wave w owns rows {k, k+1, k+4, k+5}, k = 2 (w >> 2);
            // the y partner is the same lane of fragment m ^ 1, so all four fragments of the wave collapse into ONE pooled value per lane pair
            // (MF = 8: the wave's fragments 4..7 are a second pair of x-slices: the same once more)
#pragma unroll
            for (int mset = 0; mset < MF; mset += 4)
#pragma unroll
            for (int mm = mset; mm < mset + 1; ++mm) {
                int hx, hy, hz;
                frag_xyz(mm, hx, hy, hz);
                const int gx = x0 + hx, gy = y0 + hy, gz = z0 + hz;
                const bool writer = !(v & 1) && gx < DX && gy < D && gz < D;     // D even: the whole cell is inside
                // per-lane part of the store address as ONE 32-bit byte offset (pooled voxel, the lane's half of its 8-channel group, + one group
                // plane for kq >= 2), the rest wave-uniform: see EPI_STORE (a spilled 64-bit address reloaded inside the store loop = vmcnt(0) per store)
                const unsigned pvoff = ((unsigned)(((gx >> 1) * Do + (gy >> 1)) * Do + (gz >> 1)) + (unsigned)(kq >> 1) * (unsigned)VOLo) * 16u + (unsigned)(kq & 1) * 8u;
#pragma unroll
                for (int n = 0; n < NF; ++n) {
                    half4 h, l;
                    float lo32[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = fmaxf(y[mm][n][r], y[mm + 2][n][r]);
                        t = fmaxf(t, fmaxf(y[mset + 1][n][r], y[mset + 3][n][r]));
                        t = fmaxf(t, __shfl_xor(t, 1));
                        if constexpr (SPLIT == 1) {
                            _Float16 hh, ll;
                            sn_split(t, hh, ll);
                            h[r] = hh; l[r] = ll;
                        } else {
                            h[r] = (_Float16)t;
                            lo32[r] = (t - (float)h[r]) * kMxLoMul;
                        }
                    }
                    const int ch = n * 16 + kq * 4;
                    { const uint2 hb = __builtin_bit_cast(uint2, h); sn_track_h2(trk_h, hb.x); sn_track_h2(trk_h, hb.y); }
                    if (writer && ch < a.out_cp) {
                        char *const plane = reinterpret_cast<char *>(a.pool_out) + 2 * ((size_t)b * VOLo * a.pool_cs + (size_t)(2 * n) * VOLo * 8);     // wave-uniform
                        _Float16 *o = reinterpret_cast<_Float16 *>(plane + pvoff);
                        *reinterpret_cast<half4 *>(o) = h;
                        if constexpr (SPLIT == 1) *reinterpret_cast<half4 *>(o + a.pool_lo_off) = l;
                        if constexpr (SPLIT == 2) {
                            char *slot = reinterpret_cast<char *>(o - (ch & 7) + a.pool_lo_off);
                            const _Float16 hq[4] = {h[0], h[1], h[2], h[3]};
                            sn_mx6_store_unit(slot, (ch & 7) >> 2, hq, lo32, a.mx_out_e8);
                        }
                    }
                }
            }
        } else if constexpr (EPI == EPI_STORE) {
            // Lane (v, kq) holds channels 4kq..4kq+3 of 16-channel fragment n: the 16-byte group of a voxel is split between lanes
            // l (kq even) and l+16 (kq odd). Two voxel fragments (m, m+1) are finished together and their halves exchanged with
            // v_permlane16_swap (lanes 16-31 / 48-63 of the first operand <-> lanes 0-15 / 32-47 of the second; tools/probe/
            // swap_probe.hip): afterwards the even-kq lanes hold the whole group of fragment m, the odd-kq lanes that of m+1,
            // and every lane issues ONE 16-byte store per plane instead of two 8-byte ones (the tail is store-issue bound).
            static_assert(MF % 2 == 0, "paired store epilogue");
            // The activation is a launch argument; the body is instantiated once per activation and the choice made ONCE per tile: with
            // `a.act == 0 ? relu : sigmoid` inside the value loop hipcc kept a (uniform) branch pair per VALUE - 2 x 112 taken branches per
            // wave and tile in merge_conv_a, with the sigmoid's division chain laid out in the fall-through path (round 3)
            auto store_tile = [&](auto act_c) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_c)::value;
            // storage format of the output (OSPLIT): 0 fp16 | 1 hi + lo fp16 | 2 hi + 6-bit code slots | 3 hi + fp8 code slots | 4 hi + lo + fp8 code slots
            constexpr bool O6 = OSPLIT == 2, O8 = OSPLIT == 3 || OSPLIT == 4, OLO = OSPLIT == 1 || OSPLIT == 4;
            static_assert(OSPLIT >= 0 && OSPLIT <= 4 && (!O6 || SN_MX_FMT != 0), "output storage format");
            const float f8s = sn_e8_to_float(254 - a.mx_out_e8), f8lo = 4096.0f * f8s;      // fp8 slots: premultiplier 2^s (E8M0 127 - s), lo parts additionally 2^12
            const bool odd = kq & 1;
            // (opaque per-tile lane offset into the constant table: otherwise hipcc hoists the 2 * NF LDS addresses out of the tile loop, keeps
            // them live across the K loop and spills them - a scratch reload + vmcnt(0), i.e. a wait for the stores in flight, per fragment pair)
            int cst_lane = kq * 4;
            asm volatile("" : "+v"(cst_lane));
            const float *const cstl = cst + cst_lane;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            // The folded BN constants of all NF fragments are fetched BEFORE the first store (the K loop's operand registers are free by now): a
            // global load between stores makes hipcc wait vmcnt(0) in front of its use, i.e. for the stores issued before it - one HBM round trip
            // per (fragment pair, cout fragment): 8-10 per tile, a third of conv1_x's tile time (round 3; wide layers keep theirs in LDS, CST_LDS)
            // (CST_LDS, round 4: likewise all 2 NF reads up front - one LDS round trip per tile instead of one exposed lgkmcnt(0) per (fragment pair, cout
            // fragment): 28 x ~120 clocks of merge_conv_a's 17,000-clock epilogue; the K loop's operand registers are free by now)
            f32x4 scv[NF], shv[NF];
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                if constexpr (C::CST_LDS) {
                    scv[n] = *reinterpret_cast<const f32x4 *>(cstl + n * 16);
                    shv[n] = *reinterpret_cast<const f32x4 *>(cstl + NF * 16 + n * 16);
                } else {
                    const int nl = (blockIdx.y * NF + n) * 16 + kq * 4;
                    const int nlc = nl < a.out_cp ? nl : 0;
                    scv[n] = *reinterpret_cast<const f32x4 *>(a.scale + nlc);
                    shv[n] = *reinterpret_cast<const f32x4 *>(a.shift + nlc);
                }
            }
#pragma unroll
            for (int mp = 0; mp < MF; mp += 2) {
                // This lane stores fragment mp (kq even) or mp + 1 (kq odd). Its store address = a wave-uniform plane base (sample, channel group of
                // fragment n: scalar registers) + ONE 32-bit per-lane byte offset (voxel slot, + one group plane for the lane pairs kq >= 2 that hold
                // channels 8..15 of the fragment): 64-bit per-lane addresses kept across the n loop were spilled, and a scratch reload inside a store
                // loop makes hipcc wait vmcnt(0), i.e. for every store issued before it - one HBM round trip per iteration (merge_conv_a: 34 spilled
                // registers, ~19 us of epilogue per tile, round 3)
                int hx_, hy_, hz_;
                frag_xyz(mp + (odd ? 1 : 0), hx_, hy_, hz_);
                const int gx = x0 + hx_, gy = y0 + hy_, gz = z0 + hz_;
                const bool my_valid = gx < DX && gy < D && gz < D;
                const unsigned my_voff = ((unsigned)((gx * D + gy) * D + gz) + (unsigned)(kq >> 1) * (unsigned)VOL) * 16u;
#pragma unroll
                for (int n = 0; n < NF; ++n) {
                    const int nl = (blockIdx.y * NF + n) * 16 + kq * 4;
                    const bool ch_ok = nl < a.out_cp;                    // out_cp is a multiple of 8: both lanes of a pair agree
                    const f32x4 sc = scv[n], sh = shv[n];
                    unsigned hw[2][2], lw[2][2], cw[2][2];                // [fragment of the pair][dword]: hi plane, lo plane, code plane
                    _Float16 hq[2][4];
                    float loq[2][4];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        half4 h, l;
                        float lo32[4];
                        // BN affine on register PAIRS (v_pk_mul_f32 + v_pk_add_f32: two values per instruction; separate roundings as before)
                        typedef float f32x2_ __attribute__((ext_vector_type(2)));
                        const f32x4 av = acc[mp + e][n];
                        f32x2_ pre01 = av.xy * sc.xy, pre23 = av.zw * sc.zw;
                        asm("" : "+v"(pre01), "+v"(pre23));           // (kept as register pairs: hipcc otherwise splits the packed operations back into scalar ones)
                        pre01 += sh.xy; pre23 += sh.zw;
                        const float prev[4] = {pre01.x, pre01.y, pre23.x, pre23.y};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pre = prev[r];
                            float y = ACT == 0 ? fmaxf(pre, 0.f) : sn_sigmoid(pre);
                            if constexpr (OLO) {
                                _Float16 hh, ll;
                                sn_split(y, hh, ll);
                                h[r] = hh; l[r] = ll;
                                if constexpr (O8) lo32[r] = (y - (float)hh) * f8lo;
                            } else {
                                h[r] = (_Float16)y;
                                lo32[r] = (y - (float)h[r]) * (O8 ? f8lo : kMxLoMul);
                            }
                        }
                        const uint2 hb = __builtin_bit_cast(uint2, h);
                        hw[e][0] = hb.x; hw[e][1] = hb.y;
                        sn_track_acc(trk_acc, acc[mp + e][n]);       // (the accumulator: ReLU would turn a NaN / -inf one into a clean 0)
                        sn_track_h2(trk_h, hb.x); sn_track_h2(trk_h, hb.y);
                        if constexpr (OLO) {
                            const uint2 lb = __builtin_bit_cast(uint2, l);
                            lw[e][0] = lb.x; lw[e][1] = lb.y;
                        }
                        if constexpr (O6) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { hq[e][r] = h[r]; loq[e][r] = lo32[r]; }
                        } else if constexpr (O8) {
                            // code plane, 16-byte slot per (voxel, group): [fp8(hi * 2^s) c0..c7 | fp8(lo * 2^12 * 2^s) c0..c7]
                            cw[e][0] = (unsigned)sn_pack_fp8x4((float)h[0] * f8s, (float)h[1] * f8s, (float)h[2] * f8s, (float)h[3] * f8s);
                            cw[e][1] = (unsigned)sn_pack_fp8x4(lo32[0], lo32[1], lo32[2], lo32[3]);
                        }
                    }
                    if constexpr (O6) sn_mx6_units(hq[0], loq[0], hq[1], loq[1], a.mx_out_e8, cw);   // mx_format.h
                    // r[0] = {own (even kq) | lower partner's fragment-(m+1) half (odd kq)}, r[1] = {upper partner's fragment-m half | own}
                    const auto h0 = __builtin_amdgcn_permlane16_swap(hw[0][0], hw[1][0], false, false);
                    const auto h1 = __builtin_amdgcn_permlane16_swap(hw[0][1], hw[1][1], false, false);
                    const int g0 = (a.out_coff + (blockIdx.y * NF + n) * 16) >> 3;      // first 8-channel group of fragment n (out_coff is a multiple of 8)
                    char *const plane = reinterpret_cast<char *>(a.out) + 2 * ((size_t)b * VOL * a.out_cs + (size_t)g0 * VOL * 8);     // wave-uniform
                    _Float16 *o = reinterpret_cast<_Float16 *>(plane + my_voff);
                    const bool st = my_valid && ch_ok;
                    if (st) *reinterpret_cast<u32x4 *>(o) = u32x4{h0[0], h1[0], h0[1], h1[1]};
                    if constexpr (OLO) {
                        const auto l0 = __builtin_amdgcn_permlane16_swap(lw[0][0], lw[1][0], false, false);
                        const auto l1 = __builtin_amdgcn_permlane16_swap(lw[0][1], lw[1][1], false, false);
                        if (st) *reinterpret_cast<u32x4 *>(o + a.out_lo_off) = u32x4{l0[0], l1[0], l0[1], l1[1]};
                    }
                    if constexpr (O6 || O8) {
                        const auto c0 = __builtin_amdgcn_permlane16_swap(cw[0][0], cw[1][0], false, false);   // 6-bit: low 32 bits of the units; fp8: the hi-code words
                        const auto c1 = __builtin_amdgcn_permlane16_swap(cw[0][1], cw[1][1], false, false);   // 6-bit: high 16 bits;               fp8: the lo-code words
                        _Float16 *const oc = o + (OSPLIT == 4 ? a.out_code_off : a.out_lo_off);
                        if constexpr (O6) {
                            unsigned d[3];
                            sn_mx6_join(c0[0], c1[0], c0[1], c1[1], d);
                            if (st) *reinterpret_cast<u32x4 *>(oc) = u32x4{d[0], d[1], d[2], 0u};
                        } else
                        if (st) *reinterpret_cast<u32x4 *>(oc) = u32x4{c0[0], c0[1], c1[0], c1[1]};
                    }
                }
            }
            };
            // (sigmoid store layers exist only as 1x1x1 side convolutions; the 3x3x3 kernels carry the ReLU body alone - the launcher refuses
            // anything else - because a second, sigmoid-sized body raised the register pressure of the ReLU one: spill reloads inside its store loop)
            if constexpr (KS == 1) { if (a.act == 0) store_tile(IntC<0>{}); else store_tile(IntC<1>{}); }
            else store_tile(IntC<0>{});
        } else {
            // fused merge_conv3: ReLU(BN(acc)) . w3 over the NF*16 channels; the per-channel constants of one cout fragment are fetched once
            // (L1/L2 hits) and applied to all MF voxel fragments - 3 NF loads per tile instead of 3 NF MF; every p[m] still sums n, r ascending
            float pm[MF];
#pragma unroll
            for (int m = 0; m < MF; ++m) pm[m] = 0.f;
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                const int nl = n * 16 + kq * 4;
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.scale + nl);   // (no stores in flight here: global is fine)
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.shift + nl);
                const f32x4 w3 = *reinterpret_cast<const f32x4 *>(a.w3 + nl);
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    // (register pairs: packed multiply / add / multiply, separate roundings and the same summation order as the scalar form)
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
                    const f32x4 av = acc[m][n];
                    f32x2_ t01 = av.xy * sc.xy, t23 = av.zw * sc.zw;
                    asm("" : "+v"(t01), "+v"(t23));
                    t01 += sh.xy; t23 += sh.zw;
                    f32x2_ q01 = f32x2_{fmaxf(t01.x, 0.f), fmaxf(t01.y, 0.f)} * w3.xy, q23 = f32x2_{fmaxf(t23.x, 0.f), fmaxf(t23.y, 0.f)} * w3.zw;
                    asm("" : "+v"(q01), "+v"(q23));
                    pm[m] += q01.x; pm[m] += q01.y; pm[m] += q23.x; pm[m] += q23.y;
                }
            }
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                float p = pm[m];
                int hx_, hy_, hz_;
                frag_xyz(m, hx_, hy_, hz_);
                const int gx = x0 + hx_, gy = y0 + hy_, gz = z0 + hz_;
                const bool valid = gx < DX && gy < D && gz < D;
                const size_t vox = ((size_t)(b * DX + gx) * D + gy) * D + gz;
                p += __shfl_xor(p, 16);
                p += __shfl_xor(p, 32);
                const float t = p * a.scale3 + a.shift3;
                bad |= !(fabsf(t) <= 3.0e38f);                 // NaN / inf logit (an overflowed accumulator upstream)
                if (valid && kq == 0) a.out_f32[vox] = sn_sigmoid(t);
            }
        }
        if constexpr (SN_TIMING == 10) { const long long t_tile2 = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_vm += t_tile2 - t_tile1; t_bar += t_tile1 - t_tile0; ++n_piece; }
    }
    if constexpr (C::NWB == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the ring's look-ahead DMAs of the last pieces: nothing may land in LDS after the workgroup has left)
    bad |= sn_tracked_bad(trk_acc, trk_h);
    if (a.status && __builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(a.status, a.status_bit);
    if (a.status && a.mx_sat_bits != 0 && __builtin_amdgcn_ballot_w64(sn_tracked_max_bits(trk_h) > a.mx_sat_bits) != 0 && lane == 0) atomicOr(a.status + 1, a.status_bit);
    if constexpr (SN_TIMING) {
        if (a.status && lane == 0) {        // [2..9]: per layer-bit slot of 4 x u64: kernel cycles, vmcnt-wait cycles, barrier-wait cycles, pieces (summed over waves)
            unsigned long long *t = reinterpret_cast<unsigned long long *>(a.status + 2) + 4 * (31 - __builtin_clz(a.status_bit));
            atomicAdd(t + 0, (unsigned long long)(__builtin_readcyclecounter() - t_kernel0));
            atomicAdd(t + 1, (unsigned long long)t_vm); atomicAdd(t + 2, (unsigned long long)t_bar); atomicAdd(t + 3, (unsigned long long)n_piece);
        }
    }
}

}  // namespace sn
