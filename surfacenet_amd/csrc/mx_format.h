// mx_format.h — the second activation plane of the f16m8 storage format ("slot": 16 bytes per voxel and 8-channel group) and the
// operand format of the MX step that consumes it (conv3d_mfma.h). Shared by every kernel that writes or reads a slot.
//
// A value y is stored as hi = fp16(y) in the first plane; the slot carries what the two correction terms hi_w*lo_x + lo_w*hi_x of a
// product need, as low-precision codes: hi itself and the residual lo = y - hi, multiplied by 2^kMxLoExp (so that both code sets have
// the magnitude of y) and by the tensor's static premultiplier 2^s (ConvArgs::mx_*_e8 = 127 - s, an E8M0 exponent).
//
//   SN_MX_FMT 0 (round 1): fp8 e4m3, 16 one-byte codes [hi c0..c7 | lo c0..c7]; kMxLoExp = 12, no premultiplier.
//   SN_MX_FMT 2 / 3 (round 2): fp6 e2m3 / bf6 e3m2 — v_mfma_scale_f32_16x16x128_f8f6f4 issues 6-bit operands at TWICE the fp8 rate
//     (16 instead of 32 clocks; MI355X_MICROARCH.md: fp6 at the fp4 rate), and a correction term needs 3 mantissa bits, which e4m3 and
//     e2m3 both have. 16 six-bit codes = the first 12 bytes of the slot, code p at bits [6p, 6p+6) — exactly half of a lane's 192-bit
//     MFMA operand (tools/probe/fp6_probe.hip) — in the order [hi c0..c3 | lo c0..c3 | hi c4..c7 | lo c4..c7], so that a lane that owns
//     4 channels produces one 48-bit unit. Bytes 12..15 are unused. kMxLoExp = 11.
//     Codes come from v_cvt_scalef32_pk32_{fp6,bf6}_f16 (round-to-nearest-even, saturating, result = value / scale; probe).
#pragma once
#include <hip/hip_runtime.h>

#ifndef SN_MX_FMT
#define SN_MX_FMT 2
#endif
static_assert(SN_MX_FMT == 2 || SN_MX_FMT == 3, "SN_MX_FMT: 2 fp6 e2m3, 3 bf6 e3m2 (fp8 e4m3 is a per-kernel format since round 5: conv3d_mfma.h SPLIT 3)");
constexpr int kMxLoExp = 11;
constexpr float kMxLoMul = (float)(1 << kMxLoExp);
// Static premultipliers 2^s of the code planes (fp6 forms), as E8M0 exponents 127 - s. Sized for what the tensors hold after the exact
// power-of-two renormalisation of sn_load_weights: ReLU(BN(.)) outputs of O(1) ("act"), and the concat buffer of sigmoid side outputs in
// (0, 1) ("cat"). A value above the format's range saturates and one far below it rounds to zero: either way only that element's
// correction term degrades to plain fp16 accuracy.
#ifndef SN_MX_S_ACT
#define SN_MX_S_ACT 0
#endif
#ifndef SN_MX_S_CAT
#define SN_MX_S_CAT 2
#endif
#ifndef SN_MX_S_C4
#define SN_MX_S_C4 0        // the FP8 e4m3 code planes of the conv4 chain (conv3d_mfma.h SPLIT 3; conv3_3's, conv4_1's, conv4_2's outputs): codes of the values themselves -
#endif                      // normal range 2^-6 .. 448. Measured on the device, s = -2 .. +1 are equivalent (worst L_inf 1.56e-4 .. 1.74e-4), s = 3: 2.3e-4, s = 4: 8.6e-4 -
                            // scene cubes hold activations beyond 28, which is what the 6-bit codes of the merge layers could not represent here (profiles/r5/README.md)
constexpr int kMxActE8 = 127 - SN_MX_S_ACT;
constexpr int kMxC4E8 = 127 - (SN_MX_S_C4);
constexpr int kMxCatE8 = 127 - SN_MX_S_CAT;
constexpr int kMxX0E8 = 127 + 5;      // the network input (f16m8 mode only): mean-subtracted 8-bit colours, |x| < 256 -> 2^-5

typedef int mx_v6i __attribute__((ext_vector_type(6)));
typedef _Float16 mx_v32h __attribute__((ext_vector_type(32)));
typedef float mx_v32f __attribute__((ext_vector_type(32)));

// E8M0 exponent byte -> the power of two it denotes (127 -> 1.0)
__device__ __forceinline__ float sn_e8_to_float(int e8) { return __builtin_bit_cast(float, e8 << 23); }

// 32 halfs -> 32 six-bit codes of value * 2^(127 - e8)
__device__ __forceinline__ mx_v6i sn_mx6_cvt(const mx_v32h &v, int e8)
{
#if SN_MX_FMT == 3
    return __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, sn_e8_to_float(e8));
#else
    return __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, sn_e8_to_float(e8));
#endif
}
// 32 six-bit codes -> their values (as stored, i.e. still premultiplied)
__device__ __forceinline__ mx_v32f sn_mx6_decode(const mx_v6i &c)
{
#if SN_MX_FMT == 3
    return __builtin_amdgcn_cvt_scalef32_pk32_f32_bf6(c, 1.0f);
#else
    return __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(c, 1.0f);
#endif
}

// One 48-bit unit (the codes of 4 channels: hi x4, lo x4) for each of two voxels: unit e = {w[e][0] (32 bits), w[e][1] (low 16 bits)}.
// h: the fp16 values, lo: (y - h) * 2^kMxLoExp.
__device__ __forceinline__ void sn_mx6_units(const _Float16 (&h0)[4], const float (&lo0)[4], const _Float16 (&h1)[4], const float (&lo1)[4], int e8,
                                             unsigned (&w)[2][2])
{
    typedef _Float16 mx_v8h __attribute__((ext_vector_type(8)));
    mx_v8h a, b;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a[r] = h0[r]; a[4 + r] = (_Float16)lo0[r];
        b[r] = h1[r]; b[4 + r] = (_Float16)lo1[r];
    }
    // elements 0..7 = a, 16..23 = b (-> bits 96..143 = dwords 3 and 4); the other 16 inputs are don't-cares (their codes are dropped). Zeroing them
    // cost 8 register moves per call; they are FROZEN unspecified values instead (__builtin_nondeterministic_value: whatever the registers hold,
    // but a defined value to the optimiser - a poison operand would license it to fold the whole conversion away). No instruction.
    typedef _Float16 mx_v16h __attribute__((ext_vector_type(16)));
    const mx_v16h ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    mx_v16h dc0 = {};
    const mx_v16h dc = __builtin_nondeterministic_value(dc0);
    const mx_v32h v = __builtin_shufflevector(ab, dc, 0, 1, 2, 3, 4, 5, 6, 7, 16, 17, 18, 19, 20, 21, 22, 23, 8, 9, 10, 11, 12, 13, 14, 15, 24, 25, 26, 27, 28, 29, 30, 31);
    const mx_v6i c = sn_mx6_cvt(v, e8);
    w[0][0] = (unsigned)c[0]; w[0][1] = (unsigned)c[1];
    w[1][0] = (unsigned)c[3]; w[1][1] = (unsigned)c[4];
}
// Two units (channels 0..3 = a, 4..7 = b of one voxel and group) -> the three code dwords of the slot
__device__ __forceinline__ void sn_mx6_join(unsigned a_lo32, unsigned a_hi16, unsigned b_lo32, unsigned b_hi16, unsigned (&d)[3])
{
    d[0] = a_lo32;
    d[1] = (a_hi16 & 0xFFFFu) | (b_lo32 << 16);
    d[2] = (b_lo32 >> 16) | (b_hi16 << 16);
}
