// sn_api.hip — C ABI (include/surfacenet_hip.h) over the gfx950 kernels: context, weight folding and
// MFMA-fragment packing, activation workspace, the layer schedule of the SurfaceNet graph
// (nets/SurfaceNet.py:18-76), and HIP-event profiling. Build: surfacenet_amd/csrc/Makefile.
#include "sn_internal.h"

// ------------------------------------------------------------------------------------------------
// network description (restated from nets/SurfaceNet.py:18-76; order = weight-file order, App. B)
// ------------------------------------------------------------------------------------------------
enum Kind { K_CONV3, K_CONV1, K_DIL3, K_DIL1, K_UP };
struct LayerSpec { const char *name; Kind kind; int cin, cout, act; };  // K_UP: cin = kernel size, cout = factor
static const LayerSpec kSpecs[] = {
    {"conv1_1", K_CONV3, 6, 32, 0},    {"conv1_2", K_CONV3, 32, 32, 0},   {"conv1_3", K_CONV3, 32, 32, 0},
    {"side_op1", K_CONV1, 32, 16, 1},
    {"conv2_1", K_CONV3, 32, 80, 0},   {"conv2_2", K_CONV3, 80, 80, 0},   {"conv2_3", K_CONV3, 80, 80, 0},
    {"side_op2", K_CONV1, 80, 16, 1},  {"side_op2_deconv", K_UP, 3, 2, 0},
    {"conv3_1", K_CONV3, 80, 160, 0},  {"conv3_2", K_CONV3, 160, 160, 0}, {"conv3_3", K_CONV3, 160, 160, 0},
    {"side_op3", K_CONV1, 160, 16, 1}, {"side_op3_deconv", K_UP, 5, 4, 0},
    {"conv4_1", K_DIL3, 160, 300, 0},  {"conv4_2", K_DIL3, 300, 300, 0},  {"conv4_3", K_DIL3, 300, 300, 0},
    {"side_op4", K_DIL1, 300, 16, 1},  {"side_op4_deconv", K_UP, 5, 4, 0},
    {"merge_conv_a", K_CONV3, 64, 100, 0}, {"merge_conv_b", K_CONV3, 100, 100, 0}, {"merge_conv3", K_CONV1, 100, 1, 1},
};
static constexpr int kNumSpecs = sizeof(kSpecs) / sizeof(kSpecs[0]);
static constexpr int kNetParams = 98, kAllParams = 105, kDFeature = 258, kHidden = 100;


// ------------------------------------------------------------------------------------------------
// weight preparation
// ------------------------------------------------------------------------------------------------
// OCP fp8 e4m3fn encoder (round-to-nearest-even, saturating at +-448), the format v_mfma_scale_f32_16x16x128_f8f6f4
// consumes with cbsz/blgp = 0 on gfx950.
static unsigned char fp8_e4m3(float v)
{
    if (v != v) return 0x7f;
    const unsigned char sgn = std::signbit(v) ? 0x80 : 0;
    float a = std::fabs(v);
    if (a >= 448.f) return sgn | 0x7e;                       // max finite 1.75 * 2^8
    if (a < std::ldexp(1.0f, -10)) return sgn;               // below half of the smallest subnormal (2^-9)
    int e;
    std::frexp(a, &e);                                       // a = m * 2^e, m in [0.5, 1)
    int E = e - 1;                                           // a = (1 + f) * 2^E
    if (E < -6) E = -6;                                      // subnormal range: fixed exponent 2^-6, mantissa step 2^-9
    const float q = std::nearbyint(std::ldexp(a, 3 - E));    // mantissa in units of 2^(E-3): 8..15 normal, 0..7 subnormal
    int mant = (int)q, be = E + 7;
    if (E == -6 && mant < 8) be = 0;                         // subnormal encoding
    else { if (mant == 16) { mant = 8; be += 1; } mant -= 8; }
    if (be > 15 || (be == 15 && mant > 6)) return sgn | 0x7e;
    return sgn | (unsigned char)(be << 3) | (unsigned char)mant;
}

// 6-bit minifloat encoder for the MX forms (mx_format.h): fmt 2 = fp6 e2m3 (bias 1, max 7.5), fmt 3 = bf6 e3m2 (bias 3, max 28);
// round-to-nearest-even, saturating, with subnormals (decoder: tools/probe/fp6_probe.hip dec6()).
static unsigned char mx6_encode(float v, int fmt)
{
    const int mb = fmt == 2 ? 3 : 2, bias = fmt == 2 ? 1 : 3, emax = fmt == 2 ? 3 : 7;
    const float vmax = std::ldexp((float)((2 << mb) - 1), emax - bias - mb);
    if (v != v) return 0;
    const unsigned char sgn = std::signbit(v) ? 0x20 : 0;
    float a = std::min(std::fabs(v), vmax);
    int e;
    std::frexp(a, &e);
    int E = (a > 0.f) ? e - 1 : 1 - bias;                    // a = (1 + f) * 2^E
    if (E < 1 - bias) E = 1 - bias;                          // subnormal range: the exponent of the smallest normal binade
    int mant = (int)std::nearbyint(std::ldexp(a, mb - E));   // in units of 2^(E-mb): [2^mb, 2^(mb+1)) normal, below 2^mb subnormal
    int be = E + bias;
    if (mant < (1 << mb)) be = 0;
    else { if (mant == (2 << mb)) { mant = 1 << mb; be += 1; } mant -= 1 << mb; }
    if (be > emax) { be = emax; mant = (1 << mb) - 1; }
    return sgn | (unsigned char)(be << mb) | (unsigned char)mant;
}
static float mx6_max(int fmt) { return fmt == 2 ? 7.5f : 28.f; }

// W is given as (cout, cin, k,k,k) row-major fp32 (dilated layers are transposed by the caller).
// Packed layouts (one stream per cout split, slabs back to back):
//   split 0 (f16)  : [slab][chunk][nf]{ hi fragment: 64 lanes x 8 halfs }
//   split 1 (f16x3): [slab][chunk][nf]{ hi fragment, lo fragment }
//   split 2 (f16m8): [slab][piece of 8 groups]{ chunk 2p: nf hi fragments | chunk 2p+1: nf hi fragments |
//                     nf MX fragments (2 KiB: k bytes 0-15 of all 64 lanes, then 16-31); lane (row = l&15, q = l>>4): groups 8p+2q, 8p+2q+1,
//                     four 8-byte sections [fp8(w_lo*2^12) g0 | fp8(w_hi) g0 | fp8(w_lo*2^12) g1 | fp8(w_hi) g1] }   (every piece full-size, zero padded)
// Dynamic-range normalisation (exact: every factor is a power of two). The split-fp16 storage of weights and activations has
// fp16's exponent range, so before packing
//   * input channel c of the layer arrives pre-multiplied by 2^in_exp[c] (its producer's out_exp): W[o][c] *= 2^-in_exp[c];
//   * every output row o is scaled by 2^r_o so that max_k |W[o][k]| lies in [1,2) (weights of any magnitude keep their full
//     22 bits); the BN scale absorbs 2^-r_o;
//   * a ReLU layer stores y * 2^out_exp[o] (ReLU commutes with positive scaling): scale and shift absorb 2^out_exp[o].
// conv(2^a x) * 2^b == 2^(a+b) conv(x) exactly in binary floating point as long as nothing over/underflows, so the network
// function is unchanged; in_exp / out_exp may be null (all zero).
// Host half: everything up to the three arrays that go to the device (h: packed fragments, sc / sh: folded BN). No device call, so the
// address-sanitizer build exercises its index arithmetic on the CPU (tests/test_asan.py through sn_debug_pack_host).
static int pack_conv_host(PackedConv &L, const float *W_in, const float *beta, const float *gamma, const float *mean,
                          const float *inv_std, int nf, int nsplit, int cs8max, int split, const int *in_exp, const int *out_exp,
                          std::vector<_Float16> &h, std::vector<float> &sc, std::vector<float> &sh)
{
    L.nf = nf; L.nsplit = nsplit; L.cs8max = cs8max; L.split = split;
    L.cin_p = round_up(L.cin, 8);
    const int ntap = (L.k2d ? 1 : L.ks) * L.ks * L.ks;
    const int c8_total = L.cin_p / 8;
    const int npl = split == 1 ? 2 : 1;
    L.slab_c8.clear();
    for (int left = c8_total; left > 0; left -= cs8max) L.slab_c8.push_back((unsigned char)std::min(left, cs8max));
    if ((int)L.slab_c8.size() > kMaxSlab) return fail(SN_ERR_ARG, "%s: too many channel slabs", L.name.c_str());
    std::vector<float> Wn((size_t)L.cout * L.cin * ntap);
    std::vector<int> row_exp(L.cout, 0);
    for (int o = 0; o < L.cout; ++o) {
        float mx = 0.f;
        for (int ci = 0; ci < L.cin; ++ci)
            for (int t = 0; t < ntap; ++t) {
                const size_t i = ((size_t)o * L.cin + ci) * ntap + t;
                const float w = in_exp ? std::ldexp(W_in[i], -in_exp[ci]) : W_in[i];
                if (!std::isfinite(w)) return fail(SN_ERR_ARG, "%s: non-finite weight (output channel %d, input channel %d)", L.name.c_str(), o, ci);
                Wn[i] = w;
                mx = std::max(mx, std::fabs(w));
            }
        if (mx > 0.f) row_exp[o] = -std::ilogb(mx);
        if (row_exp[o] != 0)
            for (size_t i = (size_t)o * L.cin * ntap; i < (size_t)(o + 1) * L.cin * ntap; ++i) Wn[i] = std::ldexp(Wn[i], row_exp[o]);
    }
    if (ntap == 1) L.w_norm = Wn;
    const float *W = Wn.data();
    auto wat = [&](int o, int c8abs, int j, int tap) -> float {
        const int ci = c8abs * 8 + j;
        return (o < L.cout && ci < L.cin) ? W[((size_t)o * L.cin + ci) * ntap + tap] : 0.f;
    };
    h.clear();
    // bridge chunks (conv3d_mfma.h, write_koff_part): asked for by the caller (L.bridge), granted to f16x3 3x3(x3) layers whose slabs all hold the
    // same number of channel groups and whose units per slab are not a multiple of 4: the last K-chunk of a slab is filled up with the first
    // b units of the next slab, which starts at its unit o = b
    {
        const int um = split >= 2 ? 8 : 4;                  // units per K-chunk (f16x3) / per weight piece (f16m8)
        bool ok = L.bridge && (split == 1 || (split >= 2 && !L.k2d)) && L.ks == 3 && L.slab_c8.size() >= 2 && (ntap * cs8max) % um != 0;
        for (unsigned char c8n : L.slab_c8) ok = ok && c8n == cs8max;
        L.bridge = ok ? 1 : 0;
    }
    const int nslab = (int)L.slab_c8.size();
    // units of slab si in its chunks / pieces: GU - o of its own + b of the next slab's (the kernel's slab_units)
    auto slab_units = [&](int si, int c8n, int &o, int &b) {
        const int GU = ntap * c8n, um = split >= 2 ? 8 : 4;
        o = 0; b = 0;
        if (L.bridge) { o = (si * ((um - GU % um) % um)) % um; b = (si + 1 == nslab) ? 0 : (um - (GU - o) % um) % um; }
        return GU - o + b;
    };
    if (split < 2) {
        long long chunks = 0;
        for (int si = 0; si < nslab; ++si) { int o, b; chunks += (slab_units(si, L.slab_c8[si], o, b) + 3) / 4; }
        L.wsplit_stride = chunks * nf * 512 * npl;
        h.assign((size_t)L.wsplit_stride * nsplit, (_Float16)0.f);
        for (int ns = 0; ns < nsplit; ++ns) {
            _Float16 *dst = h.data() + (size_t)ns * L.wsplit_stride;
            int c8_0 = 0;
            for (int si = 0; si < nslab; ++si) {
                const int c8n = L.slab_c8[si];
                int uo, ub;
                const int own = slab_units(si, c8n, uo, ub) - ub, nchunk = (own + ub + 3) / 4;
                for (int ch = 0; ch < nchunk; ++ch)
                    for (int f = 0; f < nf; ++f)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int o = (ns * nf + f) * 16 + (lane & 15);
                            const int g = 4 * ch + (lane >> 4);
                            _Float16 *d8 = dst + (((size_t)ch * nf + f) * npl * 64 + lane) * 8;
                            if (g >= own + ub) continue;                                    // zero padding (a tile's last slab)
                            const int u = g < own ? g + uo : g - own;                       // unit of this slab | bridge: of the next one
                            const int cb = g < own ? c8_0 : c8_0 + c8n;
                            for (int j = 0; j < 8; ++j) {
                                const float w = wat(o, cb + u % c8n, j, u / c8n);
                                const _Float16 hi = (_Float16)w;
                                d8[j] = hi;
                                if (split == 1) d8[512 + j] = (_Float16)(w - (float)hi);
                            }
                        }
                dst += (size_t)nchunk * nf * 512 * npl;
                c8_0 += c8n;
            }
        }
    } else {
        long long pieces = 0;
        for (int si = 0; si < nslab; ++si) { int o, b; pieces += (slab_units(si, L.slab_c8[si], o, b) + 7) / 8; }
        const size_t piece_halfs = (size_t)nf * 2048;          // 2 chunks x nf x 1 KiB + nf x 2 KiB = 4*nf KiB
        L.wsplit_stride = pieces * piece_halfs;
        h.assign((size_t)L.wsplit_stride * nsplit, (_Float16)0.f);
        for (int ns = 0; ns < nsplit; ++ns) {
            _Float16 *dst = h.data() + (size_t)ns * L.wsplit_stride;
            int c8_0 = 0;
            for (int si = 0; si < nslab; ++si) {
                const int c8n = L.slab_c8[si];
                int uo, ub;
                const int own = slab_units(si, c8n, uo, ub) - ub, G = own + ub, npiece = (G + 7) / 8;
                // weight of (output channel o, element j) of slot g of this slab's unit sequence: its own units uo.., then ub of the next slab's
                auto wslot = [&](int o, int g, int j) -> float {
                    const int u = g < own ? g + uo : g - own, cb = g < own ? c8_0 : c8_0 + c8n;
                    return wat(o, cb + u % c8n, j, u / c8n);
                };
                for (int p = 0; p < npiece; ++p, dst += piece_halfs) {
                    for (int cc = 0; cc < 2; ++cc)
                        for (int f = 0; f < nf; ++f)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int o = (ns * nf + f) * 16 + (lane & 15);
                                const int g = 8 * p + 4 * cc + (lane >> 4);
                                if (g >= G) continue;
                                _Float16 *d8 = dst + (((size_t)cc * nf + f) * 64 + lane) * 8;
                                for (int j = 0; j < 8; ++j) d8[j] = (_Float16)wslot(o, g, j);
                            }
                    unsigned char *mx = reinterpret_cast<unsigned char *>(dst + (size_t)2 * nf * 512);
                    for (int f = 0; f < nf; ++f)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int o = (ns * nf + f) * 16 + (lane & 15), q = lane >> 4;
                            unsigned char *frag = mx + (size_t)f * 2048;   // two lane-linear 1 KiB halves: k bytes 0-15 | 16-31
                            if (split == 2) {
                                // 6-bit forms: lane quarter q covers groups 8p+2q (elements 0..15 of its 32-element block) and 8p+2q+1 (16..31), one
                                // E8M0 scale per block. Code position within a group follows the
                                // activation slot [x_hi c0..3 | x_lo c0..3 | x_hi c4..7 | x_lo c4..7] with the OTHER part of the weight: w_lo * 2^L
                                // against x_hi, w_hi against x_lo * 2^L; the common 2^-L and the block exponent go into the scale.
                                float val[32];
                                float amax = 0.f;
                                for (int i = 0; i < 2; ++i) {
                                    const int g = 8 * p + 2 * q + i;
                                    for (int pos = 0; pos < 16; ++pos) {
                                        const int j = (pos & 3) + 4 * (pos >> 3);
                                        const bool lo_part = !(pos & 4);
                                        float t = 0.f;
                                        if (g < G) {
                                            const float w = wslot(o, g, j);
                                            const float hi = (float)(_Float16)w;
                                            t = lo_part ? (w - hi) * kMxLoMul : hi;
                                        }
                                        val[i * 16 + pos] = t;
                                        amax = std::max(amax, std::fabs(t));
                                    }
                                }
                                int E = 0;
                                if (amax > 0.f) {
                                    E = std::ilogb(amax / mx6_max(SN_MX_FMT));
                                    if (std::ldexp(amax, -E) > mx6_max(SN_MX_FMT)) ++E;
                                }
                                E = std::max(-100, std::min(100, E));
                                unsigned w6[6] = {0u, 0u, 0u, 0u, 0u, 0u};
                                for (int el = 0; el < 32; ++el) {
                                    const unsigned code = mx6_encode(std::ldexp(val[el], -E), SN_MX_FMT);
                                    const int bit = 6 * el;
                                    w6[bit >> 5] |= code << (bit & 31);
                                    if ((bit & 31) > 26) w6[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
                                }
                                // operand dwords 0..3 in the first lane-linear KiB, dwords 4..5 in the second (a 128-bit and a 64-bit read)
                                unsigned *d0 = reinterpret_cast<unsigned *>(frag + lane * 16), *d1 = reinterpret_cast<unsigned *>(frag + 1024 + lane * 16);
                                d0[0] = w6[0]; d0[1] = w6[1]; d0[2] = w6[2]; d0[3] = w6[3];
                                d1[0] = w6[4]; d1[1] = w6[5];
                                // block scales of the lane's nf fragments: bytes 8.. of its 16 bytes in the second KiB of fragment 0
                                (mx + 1024 + lane * 16 + 8)[f] = (unsigned char)std::max(0, std::min(254, 127 + E - kMxLoExp));
                                continue;
                            }
                            constexpr float kLo8 = 4096.0f;     // fp8 e4m3 form (split 3): lo parts premultiplied by 2^12, undone by the MX step's A-side scale (E8M0 115)
                            for (int i = 0; i < 4; ++i) {
                                // fp8 form: lane quarter q covers groups 8p+2q, 8p+2q+1, 8-byte sections [w_lo | w_hi | w_lo | w_hi] (the activation slots read [x_hi | x_lo])
                                const int g = 8 * p + 2 * q + (i >> 1);
                                const bool lo_part = !(i & 1);
                                if (g >= G) continue;
                                for (int j = 0; j < 8; ++j) {
                                    const float w = wslot(o, g, j);
                                    const float hi = (float)(_Float16)w;
                                    const int kb = i * 8 + j;
                                    frag[(kb >> 4) * 1024 + lane * 16 + (kb & 15)] = lo_part ? fp8_e4m3((w - hi) * kLo8) : fp8_e4m3(hi);
                                }
                            }
                        }
                }
                c8_0 += c8n;
            }
        }
    }
    sc.assign((size_t)nsplit * nf * 16 + 16, 0.f);
    sh.assign((size_t)nsplit * nf * 16 + 16, 0.f);
    for (int o = 0; o < L.cout; ++o) {
        const float s = gamma[o] * inv_std[o];   // Lasagne BatchNormLayer, deterministic=True
        const int oe = out_exp ? out_exp[o] : 0;
        sc[o] = std::ldexp(s, oe - row_exp[o]);
        sh[o] = std::ldexp(beta[o] - mean[o] * s, oe);
        if (!std::isfinite(sc[o]) || !std::isfinite(sh[o]) || (s != 0.f && sc[o] == 0.f))
            return fail(SN_ERR_ARG, "%s: folded BatchNorm scale / shift of output channel %d leaves the fp32 range (gamma %g, inv_std %g, "
                                    "row exponent %d, output exponent %d)", L.name.c_str(), o, gamma[o], inv_std[o], row_exp[o], oe);
    }
    L.macs_per_voxel = (double)L.cin * L.cout * ntap;
    return SN_OK;
}

int pack_conv(sn_ctx *c, PackedConv &L, const float *W_in, const float *beta, const float *gamma, const float *mean,
              const float *inv_std, int nf, int nsplit, int cs8max, int split, const int *in_exp, const int *out_exp)
{
    std::vector<_Float16> h;
    std::vector<float> sc, sh;
    int rc;
    if ((rc = pack_conv_host(L, W_in, beta, gamma, mean, inv_std, nf, nsplit, cs8max, split, in_exp, out_exp, h, sc, sh)) != SN_OK) return rc;
    if ((rc = dev_alloc(c, &L.wpack, h.size() + 8192)) != SN_OK) return rc;
    if ((rc = dev_alloc(c, &L.scale, sc.size())) != SN_OK) return rc;
    if ((rc = dev_alloc(c, &L.shift, sh.size())) != SN_OK) return rc;
    HIPCHK(hipMemcpy(L.wpack, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(L.scale, sc.data(), sc.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(L.shift, sh.data(), sh.size() * sizeof(float), hipMemcpyHostToDevice));
    return SN_OK;
}

// A fragments of a 16-output 1x1x1 layer for the EPI_SIDEPOOL epilogue of its producer (NF 16-channel fragments per lane group):
// [K-chunk q][hi | lo][lane][8 halfs], lane (o = lane & 15, kq = lane >> 4), k = 8*kq + j <-> input channel 16*(2q + (j >= 4)) + 4*kq + (j & 3).
static int pack_side_frag(sn_ctx *c, PackedConv &S, int producer_nf)
{
    if (S.cout != 16 || S.w_norm.size() != (size_t)16 * S.cin) return fail(SN_ERR_STATE, "%s: not a 16-output 1x1x1 layer", S.name.c_str());
    const int nq = (producer_nf + 1) / 2;
    std::vector<_Float16> h((size_t)nq * 2 * 64 * 8, (_Float16)0.f);
    for (int q = 0; q < nq; ++q)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int o = lane & 15, kq = lane >> 4, ci = 16 * (2 * q + (j >> 2)) + 4 * kq + (j & 3);
                const float w = ci < S.cin ? S.w_norm[(size_t)o * S.cin + ci] : 0.f;
                const _Float16 hi = (_Float16)w;
                h[((size_t)(q * 2 + 0) * 64 + lane) * 8 + j] = hi;
                h[((size_t)(q * 2 + 1) * 64 + lane) * 8 + j] = (_Float16)(w - (float)hi);
            }
    int rc;
    if (S.side_frag) { dev_free_owned(c, S.side_frag); S.side_frag = nullptr; }
    if ((rc = dev_alloc(c, &S.side_frag, h.size())) != SN_OK) return rc;
    HIPCHK(hipMemcpy(S.side_frag, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    return SN_OK;
}

struct TileChoice { int nf, nsplit, cs8max; };
// Must agree with the kernel instantiations in run_net_t<SPLIT> (launch_conv verifies it).
static TileChoice tile_for(const LayerSpec &sp, int split)
{
    if (sp.kind == K_CONV1 || sp.kind == K_DIL1) return {1, 1, 5};
    if (sp.kind == K_DIL3) return {5, 4, 1};        // conv4: dilation-2 halo is big -> 8-channel slabs; 4 x 80 output channels
    if (sp.cout == 32) return {2, 1, 1};
    if (sp.cout == 80) return {5, 1, 1};
    if (sp.cout == 160) return {5, 2, 1};
    // cout 100 (merge_conv_a/b). f16 mode has LDS room for two 8-channel groups per slab in merge_conv_a (-18 % there)
    return {7, 1, (split == 0 && sp.cin == 64) ? 2 : 1};
}

template <int SPLIT>
static int launch_pool(sn_ctx *c, const char *tag, Act in, Act out, int B, int D, int C)
{
    const long long total = (long long)B * (D / 2) * (D / 2) * (D / 2) * (C / 8);
    ProfScope ps(c, tag, 0, (double)B * D * D * D * C * 2.0 * 1.125 * (SPLIT ? 2 : 1));
    hipLaunchKernelGGL((maxpool2_kernel<SPLIT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, in.p, out.p, D, C,
                       total, in.lo, out.lo);
    HIPCHK(hipGetLastError());
    return SN_OK;
}

template <int SPLIT, int OSPLIT = SPLIT>
static int launch_up3(sn_ctx *c, Act s2, Act s3, Act s4, Act cat, int B, int Do, int cat_cs)
{
    const long long total = (long long)B * Do * Do * Do * 6;
    ProfScope ps(c, "side_op234_deconv", 0, (double)B * Do * Do * Do * 48 * 2.0 * (SPLIT ? 2 : 1));
    static const bool per_voxel = sn_ab_switch("SN_UPSAMPLE_PER_VOXEL") != nullptr;       // A/B: the round-1 kernel (one thread per output voxel, corners from L2)
    if (Do % 8 == 0 && Do <= 64 && !per_voxel)
        hipLaunchKernelGGL((upsample3_cat_tiled_kernel<SPLIT, OSPLIT>), dim3((unsigned)(B * 6 * (Do / 8) * (Do / 8))), dim3(256), 0, c->stream, s2.p, s3.p, s4.p,
                           cat.p, Do, cat_cs, s2.lo, s3.lo, s4.lo, cat.lo, c->mx_cat_e8);
    else
    hipLaunchKernelGGL((upsample3_cat_kernel<SPLIT, OSPLIT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, s2.p, s3.p, s4.p,
                       cat.p, Do, cat_cs, total, s2.lo, s3.lo, s4.lo, cat.lo, c->mx_cat_e8);
    HIPCHK(hipGetLastError());
    return SN_OK;
}

// x0 [S][s^3][8] fp16 -> unf [S][s^3] fp32 surface probabilities (nets/SurfaceNet.py:18-76).
// Kernel configurations <KS, DIL, MF, NF, EPI, SPLIT, CS8, PCH> per layer family; LDS budgets in DESIGN.md.
template <int SP>
static int run_net_t(sn_ctx *c, int S, float *unf)
{
    const int s = c->s, D2 = s / 2, D3 = s / 4;
    const long long M = c->max_samples, v1 = (long long)s * s * s, v2 = v1 / 8, v3 = v1 / 64;
    auto A = [&](_Float16 *p, long long vox, int ch) { return Act{p, SP ? M * vox * ch : 0}; };
    const Act x0 = A(c->x0, v1, 8), a1 = A(c->a1, v1, 32), b1 = A(c->b1, v1, 32), cat = A(c->cat, v1, 64), p1 = A(c->p1, v2, 32),
              a2 = A(c->a2, v2, 80), b2 = A(c->b2, v2, 80), s2 = A(c->s2, v2, 16), p2 = A(c->p2, v3, 80), a3 = A(c->a3, v3, 160),
              b3 = A(c->b3, v3, 160), a4 = A(c->a4, v3, 304), b4 = A(c->b4, v3, 304), s3 = A(c->s3, v3, 16), s4 = A(c->s4, v3, 16),
              ma = A(c->ma, v1, 104), none = Act{nullptr, 0};
    int rc;
#define RUN(x) do { if ((rc = (x)) != SN_OK) return rc; } while (0)
// conv1_x: 8 voxel fragments per wave = 16x8x8 tiles (half the weight staging and weight-fragment reads per MFMA of an 8x8x8 tile: -6..7 %, A/B r3l),
// 4-chunk weight pieces - the LDS holds no more
#define CONV1 3, 1, 8, 2, EPI_STORE, SP, 1, (SP == 2 ? 2 : 4), 8, 0
#define SIDE  1, 1, 4, 1, EPI_STORE, SP, 5, 2, 4, 0
// conv2_x / conv3_x: the ping-pong loop needs >= 2 chunks per weight piece, which fits the LDS only with 8-channel slabs (4-chunk pieces: a 7-chunk slab = pieces of 4 + 3)
#define C23_CS8 1
#define C23_PCH (SP == 2 ? 2 : 4)
#define CONV2 3, 1, 4, 5, EPI_STORE, SP, C23_CS8, C23_PCH, 8, 0
#define CONV3 3, 1, 4, 5, EPI_STORE, SP, C23_CS8, C23_PCH, 8, 0
#define CONV4 3, 2, 4, 5, EPI_STORE, SP, 1, 2, 8, 0
// f16 mode (one activation plane): 4-chunk weight pieces halve the barriers; merge_conv_a also takes 16-channel slabs
#define MERGA 3, 1, 4, 7, EPI_STORE, SP, (SP == 0 ? 2 : 1), (SP == 0 ? 4 : 2), 8, 0
#define MERGB 3, 1, 4, 7, EPI_FINAL, SP, 1, (SP == 0 ? 4 : 2), 8, 0
    auto &L = c->conv;
    RUN((launch_conv<CONV1>(c, L["conv1_1"], x0, 8, a1, 32, 0, 32, nullptr, S, s)));
    RUN((launch_conv<CONV1>(c, L["conv1_2"], a1, 32, b1, 32, 0, 32, nullptr, S, s)));
    // conv1_3 with its two consumers in the epilogue: side_op1 (1x1x1 + BN + sigmoid -> concat channels 0..15) and pool1; the 32-channel
    // full-resolution tensor itself is never written (nets/SurfaceNet.py:35-38).
    // f16x3 default (tail_m8 == 2): merge_conv_a AND merge_conv_b compute in f16m8 (main term f16, both correction terms on one MX-fp8
    // MFMA), so everything that writes the concat buffer stores it in the f16m8 format (OSPLIT = 2); upstream stays three-fp16-MFMA.
    const bool cat_m8 = SP == 1 && c->tail_m8 >= 2;
    static const bool unfused = sn_ab_switch("SN_NO_EPI_FUSION") != nullptr;      // A/B measurements: the three separate launches
    if (!unfused) {
        const SideFuse sf1{&L["side_op1"], cat, 64, 0, p1, 32};
        // (f16x3: 16x8x8 tiles as conv1_1 / conv1_2)
        if (cat_m8) { if constexpr (SP == 1) RUN((launch_conv<3, 1, 8, 2, EPI_SIDEPOOL, 1, 1, 4, 8, 0, 0, 2>(c, L["conv1_3"], b1, 32, none, 0, 0, 32, nullptr, S, s, 0, &sf1))); }
        else RUN((launch_conv<3, 1, 4, 2, EPI_SIDEPOOL, SP, 1, (SP == 2 ? 2 : 7), 8, 0>(c, L["conv1_3"], b1, 32, none, 0, 0, 32, nullptr, S, s, 0, &sf1)));
    } else {
        RUN((launch_conv<CONV1>(c, L["conv1_3"], b1, 32, a1, 32, 0, 32, nullptr, S, s)));
        if (cat_m8) { if constexpr (SP == 1) RUN((launch_conv<1, 1, 4, 1, EPI_STORE, 1, 5, 2, 4, 0, 0, 2>(c, L["side_op1"], a1, 32, cat, 64, 0, 16, nullptr, S, s))); }
        else RUN((launch_conv<SIDE>(c, L["side_op1"], a1, 32, cat, 64, 0, 16, nullptr, S, s)));
        RUN((launch_pool<SP>(c, "pool1", a1, p1, S, s, 32)));
    }
    RUN((launch_conv<CONV2>(c, L["conv2_1"], p1, 32, a2, 80, 0, 80, nullptr, S, D2)));
    RUN((launch_conv<CONV2>(c, L["conv2_2"], a2, 80, b2, 80, 0, 80, nullptr, S, D2)));
    if (!unfused) {
        // conv2_3 likewise: side_op2 (-> the 16-channel half-resolution side map) and pool2 in its epilogue (nets/SurfaceNet.py:44-47)
        const SideFuse sf2{&L["side_op2"], s2, 16, 0, p2, 80};
        RUN((launch_conv<3, 1, 4, 5, EPI_SIDEPOOL, SP, C23_CS8, C23_PCH, 8, 0>(c, L["conv2_3"], b2, 80, none, 0, 0, 80, nullptr, S, D2, 0, &sf2)));
    } else {
        RUN((launch_conv<CONV2>(c, L["conv2_3"], b2, 80, a2, 80, 0, 80, nullptr, S, D2)));
        RUN((launch_conv<SIDE>(c, L["side_op2"], a2, 80, s2, 16, 0, 16, nullptr, S, D2)));
        RUN((launch_pool<SP>(c, "pool2", a2, p2, S, D2, 80)));
    }
    RUN((launch_conv<CONV3>(c, L["conv3_1"], p2, 80, a3, 160, 0, 160, nullptr, S, D3)));
    RUN((launch_conv<CONV3>(c, L["conv3_2"], a3, 160, b3, 160, 0, 160, nullptr, S, D3)));
    // Round 5, default mode: the dilated chain conv4_1 .. conv4_3 in the f16m8e arithmetic - the main term on the f16 MFMA, both correction terms on ONE fp8 e4m3
    // MX MFMA per 64 k (2 MFMA units per product instead of 3) - on the one-wave-per-SIMD loop. fp8, not the merge layers' 6-bit codes: those ran 20 % faster
    // still but their static range cannot hold the conv4 chain's data-dependent outliers (3.6e-4 .. 4.3e-4 on scene cubes; profiles/r5/README.md). conv3_3's
    // output has readers of both kinds - side_op3 reads hi + lo planes, conv4_1 hi + code slots - and is stored with three planes; conv4_1 / conv4_2 store
    // hi + codes; conv4_3 stores hi + lo again (side_op4).
    bool c4_done = false;
    if constexpr (SP == 1) {
        if (c->c4_m8 == 2) {      // (A/B: conv4_1 stays on three fp16 MFMAs and stores hi + codes for conv4_2)
            RUN((launch_conv<CONV3>(c, L["conv3_3"], b3, 160, a3, 160, 0, 160, nullptr, S, D3)));
            RUN((launch_conv<SIDE>(c, L["side_op3"], a3, 160, s3, 16, 0, 16, nullptr, S, D3)));
            RUN((launch_conv<3, 2, 4, 5, EPI_STORE, 1, 1, 2, 8, 0, 0, 3>(c, L["conv4_1"], a3, 160, a4, 304, 0, 304, nullptr, S, D3)));
            RUN((launch_conv<3, 2, 8, 5, EPI_STORE, 3, 1, 2, 4, 0>(c, L["conv4_2"], a4, 304, b4, 304, 0, 304, nullptr, S, D3)));
            RUN((launch_conv<3, 2, 8, 5, EPI_STORE, 3, 1, 2, 4, 0, 0, 1>(c, L["conv4_3"], b4, 304, a4, 304, 0, 304, nullptr, S, D3)));
            c4_done = true;
        } else
        if (c->c4_m8) {
            const Act a3o{a3.p, a3.lo, (long long)(c->a3c - a3.p)}, a3i{a3.p, (long long)(c->a3c - a3.p)};
            RUN((launch_conv<3, 1, 4, 5, EPI_STORE, 1, C23_CS8, C23_PCH, 8, 0, 0, 4>(c, L["conv3_3"], b3, 160, a3o, 160, 0, 160, nullptr, S, D3)));
            RUN((launch_conv<SIDE>(c, L["side_op3"], a3, 160, s3, 16, 0, 16, nullptr, S, D3)));
            RUN((launch_conv<3, 2, 8, 5, EPI_STORE, 3, 1, 2, 4, 0>(c, L["conv4_1"], a3i, 160, a4, 304, 0, 304, nullptr, S, D3)));
            RUN((launch_conv<3, 2, 8, 5, EPI_STORE, 3, 1, 2, 4, 0>(c, L["conv4_2"], a4, 304, b4, 304, 0, 304, nullptr, S, D3)));
            RUN((launch_conv<3, 2, 8, 5, EPI_STORE, 3, 1, 2, 4, 0, 0, 1>(c, L["conv4_3"], b4, 304, a4, 304, 0, 304, nullptr, S, D3)));
            c4_done = true;
        }
    }
    if (!c4_done) {
    RUN((launch_conv<CONV3>(c, L["conv3_3"], b3, 160, a3, 160, 0, 160, nullptr, S, D3)));
    RUN((launch_conv<SIDE>(c, L["side_op3"], a3, 160, s3, 16, 0, 16, nullptr, S, D3)));
    RUN((launch_conv<CONV4>(c, L["conv4_1"], a3, 160, a4, 304, 0, 304, nullptr, S, D3)));
    RUN((launch_conv<CONV4>(c, L["conv4_2"], a4, 304, b4, 304, 0, 304, nullptr, S, D3)));
    RUN((launch_conv<CONV4>(c, L["conv4_3"], b4, 304, a4, 304, 0, 304, nullptr, S, D3)));
    }
    RUN((launch_conv<SIDE>(c, L["side_op4"], a4, 304, s4, 16, 0, 16, nullptr, S, D3)));
    if (cat_m8) { if constexpr (SP == 1) RUN((launch_up3<1, 2>(c, s2, s3, s4, cat, S, s, 64))); }
    else RUN((launch_up3<SP>(c, s2, s3, s4, cat, S, s, 64)));
    if constexpr (SP == 1) {
        if (c->tail_m8 >= 2) {
            // 4-wave workgroups, one wave per SIMD with 8 x 7 fragment tiles and the accumulators in AGPRs (conv3d_mfma.h, the one-wave-per-SIMD loop)
            RUN((launch_conv<3, 1, 8, 7, EPI_STORE, 2, 1, 2, 4, 0>(c, L["merge_conv_a"], cat, 64, ma, 104, 0, 104, nullptr, S, s)));
            RUN((launch_conv<3, 1, 8, 7, EPI_FINAL, 2, 1, 2, 4, 0>(c, L["merge_conv_b"], ma, 104, none, 0, 0, 0, unf, S, s)));
            return SN_OK;
        }
    }
    RUN((launch_conv<MERGA>(c, L["merge_conv_a"], cat, 64, ma, 104, 0, 104, nullptr, S, s)));
    RUN((launch_conv<MERGB>(c, L["merge_conv_b"], ma, 104, none, 0, 0, 0, unf, S, s)));
#undef RUN
    return SN_OK;
}

static int run_net(sn_ctx *c, int S, float *unf)
{
    c->last_run_samples = 0;
    const int rc = c->split == 2 ? run_net_t<2>(c, S, unf) : (c->split == 1 ? run_net_t<1>(c, S, unf) : run_net_t<0>(c, S, unf));
    if (rc == SN_OK) c->last_run_samples = S;      // (what sn_calibrate_dev may scan)
    return rc;
}

static int launch_fuse(sn_ctx *c, const float *unf, const float *w_dev, float *fused, int n, int n_vp)
{
    const int s3 = c->s * c->s * c->s;
    const long long total = (long long)n * s3;
    ProfScope ps(c, "fusion", 0, (double)total * 4.0 * (n_vp + 1));
    hipLaunchKernelGGL(fuse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, unf,
                       n_vp == 1 ? (const float *)nullptr : w_dev, fused, n_vp, s3, total);
    HIPCHK(hipGetLastError());
    return SN_OK;
}

static int launch_cvc(sn_ctx *c, int n, int n_vp, const int64_t *pairs_dev, const float *xyz_dev, const float *resol_dev,
                      const float *mean6, float *out_ncdhw, bool sub_mean, _Float16 *out_x0)
{
    if (!c->img_base || !c->cams) return fail(SN_ERR_STATE, "sn_set_images / sn_set_cameras must be called before the CVC warp");
    if (c->V_img != c->V_cam) return fail(SN_ERR_STATE, "image count (%d) != camera count (%d)", c->V_img, c->V_cam);
    CvcArgs a;
    memset(&a, 0, sizeof a);
    a.pairs = pairs_dev; a.xyz = xyz_dev; a.resol = resol_dev; a.cams = c->cams;
    a.img_base = c->img_base; a.img_off = c->img_off; a.img_h = c->img_h; a.img_w = c->img_w;
    a.out_ncdhw = out_ncdhw; a.out_x0 = out_x0;
    a.x0_lo_off = (out_x0 && c->split) ? (long long)c->max_samples * c->s * c->s * c->s * 8 : 0;
    a.x0_mode = c->split;
    static const float kVggMean[6] = {123.68f, 116.779f, 103.939f, 123.68f, 116.779f, 103.939f};  // params.py:129
    for (int i = 0; i < 6; ++i) a.mean[i] = mean6 ? mean6[i] : kVggMean[i];
    a.sub_mean_ncdhw = sub_mean ? 1 : 0;
    a.n_vp = n_vp; a.s = c->s; a.V = c->V_img;
    const int s3 = c->s * c->s * c->s;
    const double samples = (double)n * n_vp;
    // algorithmic bytes: 2 views x s^3 x 3 B gathered + written planes
    const double bytes = samples * s3 * (6.0 + (out_ncdhw ? 24.0 : 0.0) + (out_x0 ? 16.0 * (c->split ? 2 : 1) : 0.0));
    ProfScope ps(c, "cvc_warp", 0, bytes);
    hipLaunchKernelGGL(cvc_warp_kernel, dim3((unsigned)((s3 + 255) / 256), (unsigned)(n * n_vp)), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    return SN_OK;
}

static void comm_destroy_impl(sn_ctx *c);   // RCCL communicator teardown (defined with the RCCL glue below)

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *sn_last_error(void) { return g_err.c_str(); }
int sn_version(void) { return SN_ABI_VERSION; }

static int create_impl(sn_ctx *c)
{
    HIPCHK(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SN_ERR_STATE, "device %d is %s; this library is built for gfx950 (MI355X) only", c->device, prop.gcnArchName);
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    { unsigned char *z = nullptr; int rz = dev_alloc(c, &z, 4096); if (rz != SN_OK) return rz; HIPCHK(hipMemset(z, 0, 4096)); c->zero_page = z; }
    const size_t S = (size_t)c->max_samples, s = (size_t)c->s;
    const size_t v1 = s * s * s;
    int rc;
#define AL(p, n) do { if ((rc = dev_alloc(c, &c->p, (n))) != SN_OK) return rc; } while (0)
    AL(unf_ws, S * v1); AL(d_fused, S * v1);
    AL(d_num, 2 + 32 * 8 + 2048 * 8 * 4); HIPCHK(hipMemset(c->d_num, 0, sizeof(unsigned) * (2 + 32 * 8 + 2048 * 8 * 4)));     // [0] status bits; [2..] SN_TIMING diagnostic slots
    AL(d_pairs, S * 2); AL(d_xyz, S * 3); AL(d_resol, S); AL(d_w, S);
#undef AL
    return SN_OK;
}

// Activation workspace (channels-last fp16; two planes per tensor in f16x3 mode). Sized for max_samples.
static int ensure_workspace(sn_ctx *c)
{
    if (c->ws_ready && c->ws_split == c->split) return SN_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (void *p : c->ws_owned) dev_free_owned(c, p);
    c->ws_owned.clear();
    const size_t S = (size_t)c->max_samples, s = (size_t)c->s, npl = c->split ? 2 : 1;
    const size_t v1 = s * s * s, v2 = v1 / 8, v3 = v1 / 64;
    int rc;
#define AL(p, n) do { if ((rc = dev_alloc(c, &c->p, (n) * npl)) != SN_OK) return rc; c->ws_owned.push_back(c->p); } while (0)
    AL(x0, S * v1 * 8); AL(a1, S * v1 * 32); AL(b1, S * v1 * 32); AL(cat, S * v1 * 64);
    AL(p1, S * v2 * 32); AL(a2, S * v2 * 80); AL(b2, S * v2 * 80); AL(s2, S * v2 * 16);
    AL(p2, S * v3 * 80); AL(a3, S * v3 * 160); AL(b3, S * v3 * 160); AL(a4, S * v3 * 304); AL(b4, S * v3 * 304);
    AL(s3, S * v3 * 16); AL(s4, S * v3 * 16);
    AL(ma, S * v1 * 104);
#undef AL
    if (c->split == 1) {      // default mode: the fp8 code plane of conv3_3's output (read by conv4_1; side_op3 reads the hi / lo planes)
        if ((rc = dev_alloc(c, &c->a3c, (size_t)(S * v3 * 160))) != SN_OK) return rc;
        c->ws_owned.push_back(c->a3c);
    }

    c->ws_ready = true; c->ws_split = c->split;
    return SN_OK;
}

sn_ctx *sn_create(int device_id, int cube_D, int max_samples)
{
    if (cube_D < 8 || cube_D % 4 != 0 || cube_D > 96) { fail(SN_ERR_ARG, "cube_D must be a multiple of 4 in [8,96], got %d (the reference uses 32 and 64, params.py:65)", cube_D); return nullptr; }
    {
        // the conv kernel's halo addressing needs, per layer extent n (cube_D, /2, /4) and halo radius R: n <= 8, n % 8 == 0 or n % 8 >= R
        // (launch_conv checks it per launch; refuse here what would fail there: the dilated layers, R = 2, at extent cube_D/4)
        const int n4 = cube_D / 4;
        if (n4 > 8 && n4 % 8 == 1) { fail(SN_ERR_ARG, "cube_D = %d is not supported (cube_D/4 = %d leaves a 1-voxel partial tile under the dilation-2 layers)", cube_D, n4); return nullptr; }
    }
    if (max_samples < 1) { fail(SN_ERR_ARG, "max_samples must be >= 1"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fail(SN_ERR_HIP, "no HIP device visible: the MI355X path has no CPU fallback"); return nullptr; }
    if (device_id < 0 || device_id >= ndev) { fail(SN_ERR_ARG, "device_id %d out of range [0,%d)", device_id, ndev); return nullptr; }
    sn_ctx *c = new sn_ctx();
    c->device = device_id; c->s = cube_D; c->max_samples = max_samples;
    if (create_impl(c) != SN_OK) { std::string keep = g_err; sn_destroy(c); g_err = keep; return nullptr; }
    return c;
}

void sn_destroy(sn_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    comm_destroy_impl(c);
    for (auto &r : c->prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto e : c->marks) if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->comm_stream) { (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamDestroy(c->comm_stream); }
    for (auto e : c->comm_ev) if (e) (void)hipEventDestroy(e);
    if (c->comm_fork) (void)hipEventDestroy(c->comm_fork);
    for (void *p : c->owned) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// the static premultipliers of the 6-bit code planes (mx_format.h); sn_calibrate_dev replaces them with measured ones until the next call of this
static void reset_mx_exponents(sn_ctx *c)
{
    c->mx_act_e8 = kMxActE8; c->mx_cat_e8 = kMxCatE8; c->mx_c4_e8 = kMxC4E8;
    if (c->mode == SN_PRECISION_F16X3 && SN_MX_FMT != 0) {                                    // accuracy sweeps only (mx_format.h)
        if (sn_ab_switch("SN_MX_S_C4")) c->mx_c4_e8 = 127 - std::max(-8, std::min(8, atoi(sn_ab_switch("SN_MX_S_C4"))));      // (the fp8 code planes of the conv4 chain)
        if (sn_ab_switch("SN_MX_S_ACT")) c->mx_act_e8 = 127 - std::max(-8, std::min(8, atoi(sn_ab_switch("SN_MX_S_ACT"))));
        if (sn_ab_switch("SN_MX_S_CAT")) c->mx_cat_e8 = 127 - std::max(-8, std::min(8, atoi(sn_ab_switch("SN_MX_S_CAT"))));
    }
}

int sn_set_precision(sn_ctx *c, int mode)
{
    if (!c) return fail(SN_ERR_ARG, "null context");
    if (mode != SN_PRECISION_F16 && mode != SN_PRECISION_F16X3 && mode != SN_PRECISION_F16M8 && mode != SN_PRECISION_F16X3_PURE)
        return fail(SN_ERR_ARG, "unknown precision mode %d", mode);
    if (c->have_weights && mode != c->mode) c->have_weights = false;   // weights must be re-packed for the new mode
    c->mode = mode;
    c->split = mode == SN_PRECISION_F16X3_PURE ? 1 : mode;
    c->tail_m8 = mode == SN_PRECISION_F16X3 ? 2 : 0;
    c->last_run_samples = 0;
    c->c4_m8 = mode == SN_PRECISION_F16X3 ? 1 : 0;                                                  // conv4_1 .. conv4_3 on the fp8 MX step (run_net_t)
    if (c->c4_m8 && sn_ab_switch("SN_C4_M8")) c->c4_m8 = std::max(0, std::min(2, atoi(sn_ab_switch("SN_C4_M8"))));       // A/B measurements only (test-only twin)
    if (mode == SN_PRECISION_F16X3 && sn_ab_switch("SN_M8_TAIL")) c->tail_m8 = atoi(sn_ab_switch("SN_M8_TAIL")) >= 2 ? 2 : 0;   // A/B measurements only (0 = f16x3p's arithmetic)
    if (c->tail_m8 < 2) c->c4_m8 = 0;
    reset_mx_exponents(c);
    return SN_OK;
}

// Public opt-out of the one arithmetic change of round 5: conv4_1 .. conv4_3 back on three fp16 MFMAs inside the default mode
int sn_set_conv4_fp8(sn_ctx *c, int on)
{
    if (!c) return fail(SN_ERR_ARG, "null context");
    if (c->mode != SN_PRECISION_F16X3) return fail(SN_ERR_STATE, "sn_set_conv4_fp8 applies to SN_PRECISION_F16X3 only (the other modes have no fp8 conv4 step)");
    if (on < 0 || on > 2) return fail(SN_ERR_ARG, "sn_set_conv4_fp8: 0 (none), 1 (conv4_1 .. conv4_3, the default) or 2 (conv4_2 and conv4_3 only)");
    const int v = on;
    if (v != c->c4_m8) { c->have_weights = false; c->last_run_samples = 0; }   // the conv4 weights are packed for the arithmetic they run in
    c->c4_m8 = v;
    return SN_OK;
}

int sn_get_precision(sn_ctx *c) { return c ? c->mode : SN_ERR_ARG; }
void *sn_stream(sn_ctx *c) { return c ? (void *)c->stream : nullptr; }

// Waits for the context's stream and reports the asynchronous error words (numeric status of the conv layers, ray-pooling range).
static int sync_check(sn_ctx *c)
{
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->comm_stream) HIPCHK(hipStreamSynchronize(c->comm_stream));
    if (c->d_num) {
        unsigned st = 0;
        HIPCHK(hipMemcpy(&st, c->d_num, sizeof st, hipMemcpyDeviceToHost));
        if (st) {
            HIPCHK(hipMemset(c->d_num, 0, sizeof st));
            std::string names;
            for (size_t i = 0; i < 32; ++i)
                if (st & (1u << i)) names += (names.empty() ? "" : ", ") + (i < c->num_names.size() ? c->num_names[i] : std::string("similarityNet"));
            return fail(SN_ERR_RANGE, "non-finite or fp16-overflowing activation (|y| > 65504) stored by layer(s): %s - the results of the calls since the last "
                                      "sn_synchronize are invalid (weights outside the supported dynamic range, see DESIGN.md section 5)", names.c_str());
        }
    }
    if (c->d_err) {
        int e = 0;
        HIPCHK(hipMemcpy(&e, c->d_err, sizeof e, hipMemcpyDeviceToHost));
        if (e) {
            HIPCHK(hipMemset(c->d_err, 0, sizeof e));
            return fail(SN_ERR_ARG, "ray pooling: a projected pixel or depth bin fell outside the int32 range (cube on a camera plane?)");
        }
    }
    return SN_OK;
}

int sn_synchronize(sn_ctx *c)
{
    if (!c) return fail(SN_ERR_ARG, "null context");
    return sync_check(c);
}

// Warning-level numeric status (ConvArgs::mx_sat_bits): layers whose stored outputs exceeded the range of their 6-bit code plane since the last call.
int sn_numeric_status(sn_ctx *c, unsigned *saturated_bits, char *names, int names_cap)
{
    if (!c || !saturated_bits) return fail(SN_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned st = 0;
    if (c->d_num) {
        HIPCHK(hipMemcpy(&st, c->d_num + 1, sizeof st, hipMemcpyDeviceToHost));
        if (st) HIPCHK(hipMemset(c->d_num + 1, 0, sizeof st));
    }
    *saturated_bits = st;
    if (names && names_cap > 0) {
        std::string all;
        for (auto &n : c->num_names) all += n + ",";
        snprintf(names, (size_t)names_cap, "%s", all.c_str());
    }
    return SN_OK;
}

// Data-driven premultipliers of the code planes (header). Scans the hi planes of the concat buffer and of merge_conv_a's output as the last
// forward call left them.
int sn_calibrate_dev(sn_ctx *c, int n_samples, double max_sat_fraction, sn_calibration *out)
{
    if (!c || !out) return fail(SN_ERR_ARG, "null argument");
    // The default (hybrid) mode only: there the two exponents belong to exactly the two tensors scanned below - the concat buffer and merge_conv_a's
    // output. In the all-MX mode (SN_PRECISION_F16M8) mx_act_e8 is also the exponent of every tensor upstream, which the pooling / upsampling
    // kernels decode with the compile-time default: calibrating it from merge_conv_a's output alone would desynchronise them.
    if (!(c->split == 1 && c->tail_m8 >= 2) || SN_MX_FMT == 0)
        return fail(SN_ERR_STATE, "sn_calibrate_dev: only the default precision mode (SN_PRECISION_F16X3) carries calibratable 6-bit code planes");
    if (!c->cat || !c->ma || n_samples > c->max_samples) return fail(SN_ERR_ARG, "sn_calibrate_dev: n_samples must be 1..max_samples (or <= 0: all samples of the last forward call)");
    if (c->last_run_samples <= 0)
        return fail(SN_ERR_STATE, "sn_calibrate_dev: no forward call has run since the weights / the precision mode were set - run a representative batch first");
    if (n_samples <= 0) n_samples = c->last_run_samples;
    if (n_samples > c->last_run_samples)
        return fail(SN_ERR_STATE, "sn_calibrate_dev: the last forward call ran %d samples, %d were asked for (the rest of the workspace holds older data)", c->last_run_samples, n_samples);
    const bool measure_only = max_sat_fraction < 0.0;      // report the saturated fractions under the exponents in force, change nothing
    if (!measure_only && !(max_sat_fraction >= 0.0 && max_sat_fraction < 1.0)) return fail(SN_ERR_ARG, "sn_calibrate_dev: max_sat_fraction must be in [0, 1) (negative: measure only)");
    HIPCHK(hipSetDevice(c->device));
    const long long vox = (long long)c->s * c->s * c->s;
    const float lim = SN_MX_FMT == 2 ? 7.5f : 28.f;
    // two 6-bit code planes: 0 "act" = merge_conv_a's output, 1 "cat" = the concat buffer (the conv4 chain's fp8 codes have the exponent range: nothing to calibrate)
    constexpr int NT = 2, HB = kMxScanBins + 2;
    TmpDev tmp;
    unsigned long long *d_hist = tmp.get<unsigned long long>(NT * HB);
    if (!d_hist) return fail(SN_ERR_HIP, "out of device memory");
    HIPCHK(hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * NT * HB, c->stream));
    const _Float16 *tens[NT] = {c->ma, c->cat};
    const long long halfs[NT] = {(long long)n_samples * vox * 104, (long long)n_samples * vox * 64};
    for (int t = 0; t < NT; ++t)
        hipLaunchKernelGGL(mx_scan_kernel, dim3((unsigned)std::min<long long>(2048, (halfs[t] / 8 + 255) / 256)), dim3(256), 0, c->stream, tens[t], halfs[t], lim,
                           d_hist + t * HB);
    HIPCHK(hipGetLastError());
    unsigned long long h[NT][HB];
    HIPCHK(hipMemcpyAsync(h, d_hist, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int s_new[NT];
    double sat_new[NT], sat_old[NT];
    const int s_old[NT] = {127 - c->mx_act_e8, 127 - c->mx_cat_e8};
    for (int t = 0; t < NT; ++t) {
        if (h[t][kMxScanBins] == 0) {      // an all-zero tensor says nothing about its range: the exponent stays
            s_new[t] = s_old[t]; sat_new[t] = sat_old[t] = 0.0;
            continue;
        }
        const double nz = (double)h[t][kMxScanBins];
        auto frac = [&](int s) { const int j = std::max(0, std::min(kMxScanBins - 1, s + kMxScanBins / 2)); return (double)h[t][j] / nz; };
        int s = -kMxScanBins / 2;
        for (int cand = kMxScanBins / 2; cand >= -kMxScanBins / 2; --cand)
            if (frac(cand) <= max_sat_fraction) { s = cand; break; }
        if (measure_only) s = s_old[t];
        s_new[t] = s; sat_new[t] = frac(s); sat_old[t] = frac(std::max(-kMxScanBins / 2, std::min(kMxScanBins / 2, s_old[t])));
    }
    auto f16_of = [](unsigned long long bits) { unsigned short b = (unsigned short)bits; _Float16 v; memcpy(&v, &b, 2); return (float)v; };
    out->s_act_before = s_old[0]; out->s_cat_before = s_old[1]; out->s_act = s_new[0]; out->s_cat = s_new[1];
    out->sat_act_before = sat_old[0]; out->sat_cat_before = sat_old[1]; out->sat_act = sat_new[0]; out->sat_cat = sat_new[1];
    out->max_act = f16_of(h[0][kMxScanBins + 1]); out->max_cat = f16_of(h[1][kMxScanBins + 1]);
    c->mx_act_e8 = 127 - s_new[0]; c->mx_cat_e8 = 127 - s_new[1];
    return SN_OK;
}

int sn_load_weights(sn_ctx *c, const float *blob, size_t n_floats, const sn_param_desc *descs, int n_params)
{
    if (!c || !blob || !descs) return fail(SN_ERR_ARG, "null argument");
    if (n_params != kNetParams && n_params != kAllParams)
        return fail(SN_ERR_ARG, "expected %d (network) or %d (network + relative-weight MLP) parameter arrays, got %d", kNetParams, kAllParams, n_params);
    HIPCHK(hipSetDevice(c->device));
    auto count = [](const sn_param_desc &d) { size_t n = 1; for (int i = 0; i < d.ndim; ++i) n *= (size_t)d.shape[i]; return n; };
    for (int i = 0; i < n_params; ++i) {
        if (descs[i].ndim < 1 || descs[i].ndim > 5) return fail(SN_ERR_ARG, "param %d: ndim %d", i, descs[i].ndim);
        if (descs[i].offset < 0 || (size_t)descs[i].offset + count(descs[i]) > n_floats)
            return fail(SN_ERR_ARG, "param %d: [offset, offset+size) outside the blob", i);
    }
    { int rcw = ensure_workspace(c); if (rcw != SN_OK) return rcw; }
    reset_mx_exponents(c);      // (a calibration belongs to the weights it was measured with)
    // free previously loaded weights
    for (auto &kv : c->conv) { dev_free_owned(c, kv.second.wpack); dev_free_owned(c, kv.second.scale); dev_free_owned(c, kv.second.shift); dev_free_owned(c, kv.second.side_frag); }
    c->conv.clear();
    c->have_weights = false;
    c->last_run_samples = 0;

    // per-channel output exponents of the ReLU layers (see pack_conv): the stored activation is y * 2^e with e chosen from the
    // layer's own BatchNorm so that its typical magnitude (|gamma| + |beta|: z ~ N(beta, gamma^2) under true statistics) is O(1)
    std::map<std::string, std::vector<int>> out_exps;
    static const std::map<std::string, std::string> kInputOf = {
        {"conv1_2", "conv1_1"}, {"conv1_3", "conv1_2"}, {"side_op1", "conv1_3"}, {"conv2_1", "conv1_3"}, {"conv2_2", "conv2_1"},
        {"conv2_3", "conv2_2"}, {"side_op2", "conv2_3"}, {"conv3_1", "conv2_3"}, {"conv3_2", "conv3_1"}, {"conv3_3", "conv3_2"},
        {"side_op3", "conv3_3"}, {"conv4_1", "conv3_3"}, {"conv4_2", "conv4_1"}, {"conv4_3", "conv4_2"}, {"side_op4", "conv4_3"},
        {"merge_conv_b", "merge_conv_a"}, {"merge_conv3", "merge_conv_b"}};     // conv1_1 <- CVC, merge_conv_a <- sigmoid side outputs: exponent 0
    int pi = 0, rc;
    for (int li = 0; li < kNumSpecs; ++li) {
        const LayerSpec &sp = kSpecs[li];
        if (sp.kind == K_UP) {
            const int k = sp.cin;
            if (!shape_is(descs[pi], {1, 1, k, k, k})) return fail(SN_ERR_ARG, "%s: W must be (1,1,%d,%d,%d)", sp.name, k, k, k);
            // fixed kernel of nets/layers.py:363-374; the closed form in upsample_cat_kernel assumes it
            const float *W = blob + descs[pi].offset;
            const int factor = (k + 1) / 2;
            for (int a = 0; a < k; ++a) for (int b2 = 0; b2 < k; ++b2) for (int d = 0; d < k; ++d) {
                const float e = (1.f - fabsf((float)(a - (factor - 1))) / factor) * (1.f - fabsf((float)(b2 - (factor - 1))) / factor) *
                                (1.f - fabsf((float)(d - (factor - 1))) / factor);
                if (fabsf(W[(a * k + b2) * k + d] - e) > 1e-5f)
                    return fail(SN_ERR_ARG, "%s: interpolation kernel differs from the fixed __W_5D__ stencil; unsupported", sp.name);
            }
            ++pi;
            continue;
        }
        const int k = (sp.kind == K_CONV3 || sp.kind == K_DIL3) ? 3 : 1;
        const bool transposed = (sp.kind == K_DIL3 || sp.kind == K_DIL1);  // stored (C_in, C_out, ...): nets/layers.py:200-213
        const sn_param_desc &dW = descs[pi];
        if (!(transposed ? shape_is(dW, {sp.cin, sp.cout, k, k, k}) : shape_is(dW, {sp.cout, sp.cin, k, k, k})))
            return fail(SN_ERR_ARG, "%s: unexpected W shape", sp.name);
        for (int j = 1; j <= 4; ++j)
            if (!shape_is(descs[pi + j], {sp.cout})) return fail(SN_ERR_ARG, "%s: BN vector %d must have shape (%d,)", sp.name, j, sp.cout);
        const float *W = blob + dW.offset;
        const float *beta = blob + descs[pi + 1].offset, *gamma = blob + descs[pi + 2].offset;
        const float *mean = blob + descs[pi + 3].offset, *inv_std = blob + descs[pi + 4].offset;
        pi += 5;
        std::vector<float> Wt;
        const int ntap = k * k * k;
        if (transposed) {
            Wt.resize((size_t)sp.cin * sp.cout * ntap);
            for (int ci = 0; ci < sp.cin; ++ci)
                for (int o = 0; o < sp.cout; ++o)
                    for (int t = 0; t < ntap; ++t) Wt[((size_t)o * sp.cin + ci) * ntap + t] = W[((size_t)ci * sp.cout + o) * ntap + t];
            W = Wt.data();
        }
        const int *in_exp = nullptr;
        {
            auto src = kInputOf.find(sp.name);
            if (src != kInputOf.end()) in_exp = out_exps.at(src->second).data();
        }
        if (strcmp(sp.name, "merge_conv3") == 0) {
            // fused into merge_conv_b's epilogue in fp32
            std::vector<float> w3(7 * 16 + 16, 0.f);
            for (int ci = 0; ci < sp.cin; ++ci) w3[ci] = std::ldexp(W[ci], -in_exp[ci]);
            if (c->w3) dev_free_owned(c, c->w3);
            if ((rc = dev_alloc(c, &c->w3, w3.size())) != SN_OK) return rc;
            HIPCHK(hipMemcpy(c->w3, w3.data(), w3.size() * sizeof(float), hipMemcpyHostToDevice));
            c->scale3 = gamma[0] * inv_std[0];
            c->shift3 = beta[0] - mean[0] * c->scale3;
            bool fin = std::isfinite(c->scale3) && std::isfinite(c->shift3);
            for (int ci = 0; ci < sp.cin; ++ci) fin = fin && std::isfinite(w3[ci]);
            if (!fin) return fail(SN_ERR_ARG, "merge_conv3: non-finite weight or folded BatchNorm scale / shift");
            continue;
        }
        PackedConv L;
        L.name = sp.name; L.cin = sp.cin; L.cout = sp.cout; L.ks = k; L.dil = (sp.kind == K_DIL3) ? 2 : 1; L.act = sp.act;
        const int lsplit = (c->split == 1 && c->tail_m8 >= 2 && (L.name == "merge_conv_b" || L.name == "merge_conv_a")) ? 2 :
                           ((c->split == 1 && c->c4_m8 && sp.kind == K_DIL3 && !(c->c4_m8 == 2 && L.name == "conv4_1")) ? 3 : c->split);   // see run_net_t
        const TileChoice tc = tile_for(sp, lsplit);
        std::vector<int> &oe = out_exps[sp.name];
        oe.assign(sp.cout, 0);
        if (sp.act == 0)                                     // ReLU layers only: a sigmoid output lies in (0,1) as it is
            for (int o = 0; o < sp.cout; ++o) {
                const float m = std::max(std::fabs(gamma[o]), std::fabs(beta[o]));
                if (m > 0.f && std::isfinite(m)) oe[o] = std::max(-60, std::min(60, -std::ilogb(m)));
            }
        static const bool no_bridge = sn_ab_switch("SN_NO_BRIDGE") != nullptr;               // (A/B switch)
        L.bridge = (lsplit >= 1 && k == 3 && !no_bridge) ? 1 : 0;   // 27 K-chunks per four slabs instead of 28 (f16x3), 27 weight pieces per eight slabs instead of 32 (f16m8): pack_conv_host decides
        if ((rc = pack_conv(c, L, W, beta, gamma, mean, inv_std, tc.nf, tc.nsplit, tc.cs8max, lsplit, in_exp, oe.data())) != SN_OK) return rc;
        c->conv[L.name] = L;
    }
    // side_op1 / side_op2 run inside the epilogues of conv1_3 / conv2_3 (EPI_SIDEPOOL): their A fragments in the producers' register order
    if ((rc = pack_side_frag(c, c->conv["side_op1"], c->conv["conv1_3"].nf)) != SN_OK) return rc;
    if ((rc = pack_side_frag(c, c->conv["side_op2"], c->conv["conv2_3"].nf)) != SN_OK) return rc;
    c->have_relw = false;
    if (n_params == kAllParams) {
        const sn_param_desc *d = descs + pi;
        if (!shape_is(d[0], {kDFeature, kHidden}) || !shape_is(d[5], {kHidden, 1}) || !shape_is(d[6], {1}))
            return fail(SN_ERR_ARG, "relative-weight MLP arrays have unexpected shapes");
        for (int j = 1; j <= 4; ++j) if (!shape_is(d[j], {kHidden})) return fail(SN_ERR_ARG, "feature_fc1 BN vector shape");
        std::vector<float> sc(kHidden), sh(kHidden);
        const float *beta = blob + d[1].offset, *gamma = blob + d[2].offset, *mean = blob + d[3].offset, *inv_std = blob + d[4].offset;
        for (int j = 0; j < kHidden; ++j) { sc[j] = gamma[j] * inv_std[j]; sh[j] = beta[j] - mean[j] * sc[j]; }
        if (c->relw_W1) { dev_free_owned(c, c->relw_W1); dev_free_owned(c, c->relw_scale); dev_free_owned(c, c->relw_shift); dev_free_owned(c, c->relw_w2); }
        if ((rc = dev_alloc(c, &c->relw_W1, (size_t)kDFeature * kHidden)) != SN_OK) return rc;
        if ((rc = dev_alloc(c, &c->relw_scale, kHidden)) != SN_OK) return rc;
        if ((rc = dev_alloc(c, &c->relw_shift, kHidden)) != SN_OK) return rc;
        if ((rc = dev_alloc(c, &c->relw_w2, kHidden)) != SN_OK) return rc;
        HIPCHK(hipMemcpy(c->relw_W1, blob + d[0].offset, sizeof(float) * kDFeature * kHidden, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->relw_scale, sc.data(), sizeof(float) * kHidden, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->relw_shift, sh.data(), sizeof(float) * kHidden, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->relw_w2, blob + d[5].offset, sizeof(float) * kHidden, hipMemcpyHostToDevice));
        c->relw_b2 = blob[d[6].offset];
        c->have_relw = true;
    }
    c->have_weights = true;
    return SN_OK;
}

int sn_set_images(sn_ctx *c, int V, const uint8_t *const *imgs, const int *H, const int *W)
{
    if (!c || V < 1 || !imgs || !H || !W) return fail(SN_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    std::vector<long long> off(V);
    long long total = 0;
    for (int v = 0; v < V; ++v) {
        if (!imgs[v] || H[v] < 1 || W[v] < 1) return fail(SN_ERR_ARG, "image %d: null or empty", v);
        off[v] = total;
        total += (long long)H[v] * W[v] * 3;
        total = (total + 15) / 16 * 16;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    dev_free_owned(c, c->img_base); dev_free_owned(c, c->img_off); dev_free_owned(c, c->img_h); dev_free_owned(c, c->img_w);
    c->img_base = nullptr; c->img_off = nullptr; c->img_h = c->img_w = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &c->img_base, (size_t)total + 16)) != SN_OK) return rc;
    if ((rc = dev_alloc(c, &c->img_off, V)) != SN_OK) return rc;
    if ((rc = dev_alloc(c, &c->img_h, V)) != SN_OK) return rc;
    if ((rc = dev_alloc(c, &c->img_w, V)) != SN_OK) return rc;
    for (int v = 0; v < V; ++v) HIPCHK(hipMemcpy(c->img_base + off[v], imgs[v], (size_t)H[v] * W[v] * 3, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->img_off, off.data(), sizeof(long long) * V, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->img_h, H, sizeof(int) * V, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->img_w, W, sizeof(int) * V, hipMemcpyHostToDevice));
    c->V_img = V;
    c->h_img_off = off; c->h_img_h.assign(H, H + V); c->h_img_w.assign(W, W + V);
    return SN_OK;
}

int sn_set_cameras(sn_ctx *c, int V, const double *P)
{
    if (!c || V < 1 || !P) return fail(SN_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    dev_free_owned(c, c->cams);
    c->cams = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &c->cams, (size_t)V * 12)) != SN_OK) return rc;
    HIPCHK(hipMemcpy(c->cams, P, sizeof(double) * 12 * V, hipMemcpyHostToDevice));
    c->V_cam = V;
    return SN_OK;
}

// ---- device-resident entry points ------------------------------------------------------------------
static int check_batch(sn_ctx *c, int n, int n_vp, bool need_w)
{
    if (!c) return fail(SN_ERR_ARG, "null context");
    if (n < 1 || n_vp < 1) return fail(SN_ERR_ARG, "n and n_vp must be >= 1 (got %d, %d)", n, n_vp);
    if ((long long)n * n_vp > c->max_samples)
        return fail(SN_ERR_ARG, "n*n_vp = %lld exceeds the context's max_samples = %d", (long long)n * n_vp, c->max_samples);
    if (need_w && !c->have_weights) return fail(SN_ERR_STATE, "sn_load_weights has not been called");
    return SN_OK;
}

int sn_cvc_dev(sn_ctx *c, int n, int n_vp, const int64_t *pairs_dev, const float *xyz_dev, const float *resol_dev,
               const float *mean6, float *out_dev)
{
    int rc = check_batch(c, n, n_vp, false);
    if (rc != SN_OK) return rc;
    if (!pairs_dev || !xyz_dev || !resol_dev || !out_dev) return fail(SN_ERR_ARG, "null device pointer");
    HIPCHK(hipSetDevice(c->device));
    return launch_cvc(c, n, n_vp, pairs_dev, xyz_dev, resol_dev, mean6, out_dev, mean6 != nullptr, nullptr);
}

int sn_forward_dev(sn_ctx *c, int n, int n_vp, const float *X_dev, const float *w_dev, float *fused_dev, float *unfused_dev)
{
    int rc = check_batch(c, n, n_vp, true);
    if (rc != SN_OK) return rc;
    if (!X_dev || !fused_dev || (n_vp > 1 && !w_dev)) return fail(SN_ERR_ARG, "null device pointer (w is required when n_vp >= 2)");
    HIPCHK(hipSetDevice(c->device));
    const int S = n * n_vp, s3 = c->s * c->s * c->s;
    {
        ProfScope ps(c, "ncdhw_to_x0", 0, (double)S * s3 * (24.0 + 16.0));
        hipLaunchKernelGGL(ncdhw_to_x0_kernel, dim3((unsigned)((s3 + 255) / 256), (unsigned)S), dim3(256), 0, c->stream, X_dev, c->x0, s3, S,
                           c->split ? (long long)c->max_samples * s3 * 8 : 0LL, c->split);
        HIPCHK(hipGetLastError());
    }
    float *unf = unfused_dev ? unfused_dev : c->unf_ws;
    if ((rc = run_net(c, S, unf)) != SN_OK) return rc;
    return launch_fuse(c, unf, w_dev, fused_dev, n, n_vp);
}

int sn_cvc_forward_dev(sn_ctx *c, int n, int n_vp, const int64_t *pairs_dev, const float *xyz_dev, const float *resol_dev,
                       const float *mean6, const float *w_dev, float *fused_dev, float *unfused_dev, float *cvc_out_dev)
{
    int rc = check_batch(c, n, n_vp, true);
    if (rc != SN_OK) return rc;
    if (!pairs_dev || !xyz_dev || !resol_dev || !fused_dev || (n_vp > 1 && !w_dev)) return fail(SN_ERR_ARG, "null device pointer");
    HIPCHK(hipSetDevice(c->device));
    if ((rc = launch_cvc(c, n, n_vp, pairs_dev, xyz_dev, resol_dev, mean6, cvc_out_dev, true, c->x0)) != SN_OK) return rc;
    float *unf = unfused_dev ? unfused_dev : c->unf_ws;
    if ((rc = run_net(c, n * n_vp, unf)) != SN_OK) return rc;
    return launch_fuse(c, unf, w_dev, fused_dev, n, n_vp);
}

// ---- host-buffer entry points (synchronous, chunked by max_samples) ----------------------------------
static int validate_pairs(sn_ctx *c, long long count, const int64_t *pairs, std::vector<int64_t> &wrapped)
{
    wrapped.assign(pairs, pairs + count);
    for (long long i = 0; i < count; ++i) {
        int64_t v = wrapped[i];
        if (v < -(int64_t)c->V_img || v >= (int64_t)c->V_img)
            return fail(SN_ERR_ARG, "view index %lld out of range for %d views (the reference raises IndexError here)", (long long)v, c->V_img);
        if (v < 0) wrapped[i] = v + c->V_img;  // numpy negative indexing
    }
    return SN_OK;
}

static int ensure_dX(sn_ctx *c)
{
    if (c->d_X) return SN_OK;
    return dev_alloc(c, &c->d_X, (size_t)c->max_samples * 6 * c->s * c->s * c->s);
}

static int upload_batch(sn_ctx *c, int n, int n_vp, const int64_t *pairs, const float *xyz, const float *resol, const float *w)
{
    if (pairs) HIPCHK(hipMemcpyAsync(c->d_pairs, pairs, sizeof(int64_t) * 2 * n * n_vp, hipMemcpyHostToDevice, c->stream));
    if (xyz) HIPCHK(hipMemcpyAsync(c->d_xyz, xyz, sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream));
    if (resol) HIPCHK(hipMemcpyAsync(c->d_resol, resol, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    if (w) HIPCHK(hipMemcpyAsync(c->d_w, w, sizeof(float) * n * n_vp, hipMemcpyHostToDevice, c->stream));
    return SN_OK;
}

int sn_cvc(sn_ctx *c, int n, int n_vp, const int64_t *pairs, const float *xyz, const float *resol, const float *mean6, float *out)
{
    if (!c || !pairs || !xyz || !resol || !out) return fail(SN_ERR_ARG, "null argument");
    if (n < 0 || n_vp < 1) return fail(SN_ERR_ARG, "bad n / n_vp");
    if (n == 0) return SN_OK;
    if (!c->img_base || !c->cams) return fail(SN_ERR_STATE, "sn_set_images / sn_set_cameras must be called first");
    HIPCHK(hipSetDevice(c->device));
    std::vector<int64_t> wp;
    int rc;
    if ((rc = validate_pairs(c, (long long)n * n_vp * 2, pairs, wp)) != SN_OK) return rc;
    if ((rc = ensure_dX(c)) != SN_OK) return rc;
    if (n_vp > c->max_samples) return fail(SN_ERR_ARG, "n_vp exceeds max_samples");
    const size_t per = (size_t)6 * c->s * c->s * c->s;
    const int step = c->max_samples / n_vp;
    for (int i0 = 0; i0 < n; i0 += step) {
        const int m = std::min(step, n - i0);
        if ((rc = upload_batch(c, m, n_vp, wp.data() + (size_t)i0 * n_vp * 2, xyz + 3 * (size_t)i0, resol + i0, nullptr)) != SN_OK) return rc;
        if ((rc = launch_cvc(c, m, n_vp, c->d_pairs, c->d_xyz, c->d_resol, mean6, c->d_X, mean6 != nullptr, nullptr)) != SN_OK) return rc;
        HIPCHK(hipMemcpyAsync(out + (size_t)i0 * n_vp * per, c->d_X, sizeof(float) * per * m * n_vp, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return SN_OK;
}

int sn_forward(sn_ctx *c, int n, int n_vp, const float *X, const float *w, float *fused, float *unfused)
{
    if (!c || !X || !fused) return fail(SN_ERR_ARG, "null argument");
    if (n < 0 || n_vp < 1) return fail(SN_ERR_ARG, "bad n / n_vp");
    if (n_vp > 1 && !w) return fail(SN_ERR_ARG, "w is required when n_vp >= 2 (nets/SurfaceNet.py:365-372)");
    if (n == 0) return SN_OK;
    if (!c->have_weights) return fail(SN_ERR_STATE, "sn_load_weights has not been called");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_dX(c)) != SN_OK) return rc;
    if (n_vp > c->max_samples) return fail(SN_ERR_ARG, "n_vp exceeds max_samples");
    const size_t s3 = (size_t)c->s * c->s * c->s, per = 6 * s3;
    const int step = c->max_samples / n_vp;
    for (int i0 = 0; i0 < n; i0 += step) {
        const int m = std::min(step, n - i0);
        HIPCHK(hipMemcpyAsync(c->d_X, X + (size_t)i0 * n_vp * per, sizeof(float) * per * m * n_vp, hipMemcpyHostToDevice, c->stream));
        if ((rc = upload_batch(c, m, n_vp, nullptr, nullptr, nullptr, n_vp > 1 ? w + (size_t)i0 * n_vp : nullptr)) != SN_OK) return rc;
        if ((rc = sn_forward_dev(c, m, n_vp, c->d_X, c->d_w, c->d_fused, c->unf_ws)) != SN_OK) return rc;
        HIPCHK(hipMemcpyAsync(fused + (size_t)i0 * s3, c->d_fused, sizeof(float) * s3 * m, hipMemcpyDeviceToHost, c->stream));
        if (unfused) HIPCHK(hipMemcpyAsync(unfused + (size_t)i0 * n_vp * s3, c->unf_ws, sizeof(float) * s3 * m * n_vp, hipMemcpyDeviceToHost, c->stream));
        if ((rc = sync_check(c)) != SN_OK) return rc;
    }
    return SN_OK;
}

int sn_cvc_forward(sn_ctx *c, int n, int n_vp, const int64_t *pairs, const float *xyz, const float *resol, const float *mean6,
                   const float *w, float *fused, float *unfused, float *cvc_out)
{
    if (!c || !pairs || !xyz || !resol || !fused) return fail(SN_ERR_ARG, "null argument");
    if (n < 0 || n_vp < 1) return fail(SN_ERR_ARG, "bad n / n_vp");
    if (n_vp > 1 && !w) return fail(SN_ERR_ARG, "w is required when n_vp >= 2");
    if (n == 0) return SN_OK;
    if (!c->have_weights) return fail(SN_ERR_STATE, "sn_load_weights has not been called");
    if (!c->img_base || !c->cams) return fail(SN_ERR_STATE, "sn_set_images / sn_set_cameras must be called first");
    HIPCHK(hipSetDevice(c->device));
    std::vector<int64_t> wp;
    int rc;
    if ((rc = validate_pairs(c, (long long)n * n_vp * 2, pairs, wp)) != SN_OK) return rc;
    if (cvc_out && (rc = ensure_dX(c)) != SN_OK) return rc;
    if (n_vp > c->max_samples) return fail(SN_ERR_ARG, "n_vp exceeds max_samples");
    const size_t s3 = (size_t)c->s * c->s * c->s, per = 6 * s3;
    const int step = c->max_samples / n_vp;
    for (int i0 = 0; i0 < n; i0 += step) {
        const int m = std::min(step, n - i0);
        if ((rc = upload_batch(c, m, n_vp, wp.data() + (size_t)i0 * n_vp * 2, xyz + 3 * (size_t)i0, resol + i0,
                               n_vp > 1 ? w + (size_t)i0 * n_vp : nullptr)) != SN_OK) return rc;
        if ((rc = sn_cvc_forward_dev(c, m, n_vp, c->d_pairs, c->d_xyz, c->d_resol, mean6, c->d_w, c->d_fused, c->unf_ws,
                                     cvc_out ? c->d_X : nullptr)) != SN_OK) return rc;
        HIPCHK(hipMemcpyAsync(fused + (size_t)i0 * s3, c->d_fused, sizeof(float) * s3 * m, hipMemcpyDeviceToHost, c->stream));
        if (unfused) HIPCHK(hipMemcpyAsync(unfused + (size_t)i0 * n_vp * s3, c->unf_ws, sizeof(float) * s3 * m * n_vp, hipMemcpyDeviceToHost, c->stream));
        if (cvc_out) HIPCHK(hipMemcpyAsync(cvc_out + (size_t)i0 * n_vp * per, c->d_X, sizeof(float) * per * m * n_vp, hipMemcpyDeviceToHost, c->stream));
        if ((rc = sync_check(c)) != SN_OK) return rc;
    }
    return SN_OK;
}

int sn_relative_weights(sn_ctx *c, int n, int n_vp, const float *features, float *weights)
{
    if (!c || !features || !weights) return fail(SN_ERR_ARG, "null argument");
    if (n < 0 || n_vp < 1) return fail(SN_ERR_ARG, "bad n / n_vp");
    if (n == 0) return SN_OK;
    if (!c->have_relw) return fail(SN_ERR_STATE, "the relative-weight MLP arrays (params 98..104) were not loaded");
    HIPCHK(hipSetDevice(c->device));
    const size_t rows = (size_t)n * n_vp;
    TmpDev t;      // freed on every return path
    float *d_f = t.get<float>(rows * kDFeature), *d_z = t.get<float>(rows), *d_o = t.get<float>(rows);
    if (!d_f || !d_z || !d_o) return fail(SN_ERR_NOMEM, "sn_relative_weights: device allocation failed");
    HIPCHK(hipMemcpyAsync(d_f, features, rows * kDFeature * sizeof(float), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(relw_mlp_kernel, dim3((unsigned)rows), dim3(128), 0, c->stream, d_f, c->relw_W1, c->relw_scale, c->relw_shift,
                       c->relw_w2, c->relw_b2, d_z, kDFeature, kHidden);
    hipLaunchKernelGGL(relw_softmax_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, c->stream, d_z, d_o, n, n_vp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(weights, d_o, rows * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SN_OK;
}

// viewPairSelection's weight computation for EVERY 2-combination of views at once (utils/viewPairSelection.py:63-77): the
// (n_cubes*P, 258) feature matrix is never materialised, embeddings cross PCIe once.
int sn_viewpair_weights(sn_ctx *c, int n_cubes, int n_views, const float *embeddings, const float *dissimilarity, const float *theta,
                        float *weights)
{
    if (!c || !embeddings || !dissimilarity || !theta || !weights) return fail(SN_ERR_ARG, "null argument");
    if (n_cubes < 0 || n_views < 2) return fail(SN_ERR_ARG, "need n_cubes >= 0 and n_views >= 2");
    if (n_cubes == 0) return SN_OK;
    if (!c->have_relw) return fail(SN_ERR_STATE, "the relative-weight MLP arrays (params 98..104) were not loaded");
    if (kDFeature != 258 || kHidden > 128) return fail(SN_ERR_STATE, "unexpected MLP geometry");
    HIPCHK(hipSetDevice(c->device));
    const int P = n_views * (n_views - 1) / 2;
    std::vector<int> pairs;
    for (int i = 0; i < n_views; ++i)
        for (int j = i + 1; j < n_views; ++j) { pairs.push_back(i); pairs.push_back(j); }
    const size_t rows = (size_t)n_cubes * P, ne = (size_t)n_cubes * n_views * 128;
    TmpDev t;
    float *d_e = t.get<float>(ne), *d_d = t.get<float>(rows), *d_t = t.get<float>(rows), *d_z = t.get<float>(rows), *d_o = t.get<float>(rows);
    int *d_p = t.get<int>(pairs.size());
    if (!d_e || !d_d || !d_t || !d_z || !d_o || !d_p) return fail(SN_ERR_NOMEM, "sn_viewpair_weights: device allocation failed");
    HIPCHK(hipMemcpyAsync(d_e, embeddings, sizeof(float) * ne, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_d, dissimilarity, sizeof(float) * rows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_t, theta, sizeof(float) * rows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_p, pairs.data(), sizeof(int) * pairs.size(), hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, "relw_mlp_pairs", 2.0 * rows * kDFeature * kHidden, (double)rows * 12.0 + (double)ne * 4.0);
        hipLaunchKernelGGL(relw_mlp_pairs_kernel, dim3((unsigned)rows), dim3(128), 0, c->stream, d_e, d_p, d_d, d_t, n_views, P, c->relw_W1, c->relw_scale,
                           c->relw_shift, c->relw_w2, c->relw_b2, d_z, kHidden);
        hipLaunchKernelGGL(relw_softmax_kernel, dim3((unsigned)((n_cubes + 127) / 128)), dim3(128), 0, c->stream, d_z, d_o, n_cubes, P);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(weights, d_o, sizeof(float) * rows, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SN_OK;
}

// utils.generate_voxelLevelWeighted_coloredCubes (utils/utils.py:8-42) on device-resident tensors
int sn_color_fuse_dev(sn_ctx *c, int n, int n_vp, const float *cvc_dev, const float *mean6, const float *unfused_dev,
                      const float *w_dev, unsigned char *rgb_dev)
{
    if (!c || !cvc_dev || !unfused_dev || !w_dev || !rgb_dev) return fail(SN_ERR_ARG, "null argument");
    if (n < 1 || n_vp < 1) return fail(SN_ERR_ARG, "bad n / n_vp");
    HIPCHK(hipSetDevice(c->device));
    static const float kVggMean[6] = {123.68f, 116.779f, 103.939f, 123.68f, 116.779f, 103.939f};
    const float *m = mean6 ? mean6 : kVggMean;
    const int s3 = c->s * c->s * c->s;
    const long long total = (long long)n * s3;
    ProfScope ps(c, "color_fuse", 0, (double)total * (n_vp * 28.0 + 3.0));
    hipLaunchKernelGGL(color_fuse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, cvc_dev, unfused_dev, w_dev,
                       rgb_dev, n_vp, s3, total, m[0], m[1], m[2], m[3], m[4], m[5]);
    HIPCHK(hipGetLastError());
    return SN_OK;
}

int sn_color_fuse(sn_ctx *c, int n, int n_vp, const float *cvc, const float *mean6, const float *unfused, const float *w,
                  unsigned char *rgb)
{
    if (!c || !cvc || !unfused || !w || !rgb) return fail(SN_ERR_ARG, "null argument");
    if (n < 0 || n_vp < 1) return fail(SN_ERR_ARG, "bad n / n_vp");
    if (n == 0) return SN_OK;
    HIPCHK(hipSetDevice(c->device));
    const size_t s3 = (size_t)c->s * c->s * c->s;
    float *d_c = nullptr, *d_u = nullptr, *d_wt = nullptr; unsigned char *d_r = nullptr;
    hipError_t e = hipMalloc((void **)&d_c, sizeof(float) * 6 * s3 * n * n_vp);
    if (e == hipSuccess) e = hipMalloc((void **)&d_u, sizeof(float) * s3 * n * n_vp);
    if (e == hipSuccess) e = hipMalloc((void **)&d_wt, sizeof(float) * n * n_vp);
    if (e == hipSuccess) e = hipMalloc((void **)&d_r, 3 * s3 * n);
    int rc = SN_OK;
    if (e == hipSuccess) e = hipMemcpyAsync(d_c, cvc, sizeof(float) * 6 * s3 * n * n_vp, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_u, unfused, sizeof(float) * s3 * n * n_vp, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_wt, w, sizeof(float) * n * n_vp, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) rc = sn_color_fuse_dev(c, n, n_vp, d_c, mean6, d_u, d_wt, d_r);
    if (e == hipSuccess && rc == SN_OK) e = hipMemcpyAsync(rgb, d_r, 3 * s3 * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_c); (void)hipFree(d_u); (void)hipFree(d_wt); (void)hipFree(d_r);
    if (e != hipSuccess) return fail(SN_ERR_HIP, "sn_color_fuse: %s", hipGetErrorString(e));
    return rc;
}

// ---- multi-GPU exchange: RCCL all-gather over xGMI (the one collective of the path) ---------------------
// librccl is dlopen'ed on first use, so single-GPU users never load it and a host that already carries an RCCL
// (e.g. torch) keeps its own copy. ncclUniqueId = 128 opaque bytes; ncclFloat32 = 7.
namespace {
struct Rccl {
    void *h = nullptr;
    int version = 0;                              // ncclGetVersion code (major*10000 + minor*100 + patch)
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, const char (*)[128], int) = nullptr;   // real ABI passes the 128-byte struct by value
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
struct UniqueId { char b[128]; };
Rccl g_rccl;
std::string g_rccl_file;                          // the shared object the entry points were bound from (dladdr)
int rccl_load()
{
    if (g_rccl.h) return SN_OK;
    // A process that already carries an RCCL (torch.distributed's, say) must not get a second copy with its own topology state and its own view of the
    // devices: an already-mapped library is taken first (RTLD_NOLOAD finds it by soname whatever directory it came from); only then is one loaded.
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    bool was_mapped = true;
    for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
    if (!h) {
        was_mapped = false;
        for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!h) return fail(SN_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void **, int, const char (*)[128], int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy)
        return fail(SN_ERR_COMM, "librccl lacks an expected symbol");
    // The five entry points above are bound through hand-written prototypes (no rccl.h dependency at build time): they are
    // the NCCL 2.x ABI (ncclUniqueId = 128 bytes by value, ncclDataType_t ncclFloat32 = 7). Refuse anything else.
    int (*get_version)(int *) = (int (*)(int *))dlsym(h, "ncclGetVersion");
    int ver = 0;
    if (!get_version || get_version(&ver) != 0 || ver < 20000 || ver >= 30000) {
        dlclose(h);
        return fail(SN_ERR_COMM, "librccl reports version code %d: this library binds the NCCL 2.x ABI only", ver);
    }
    Dl_info info;
    g_rccl_file = (dladdr((void *)g_rccl.AllGather, &info) && info.dli_fname) ? info.dli_fname : "?";
    g_rccl_file += was_mapped ? " (already mapped by the host process)" : " (loaded by libsurfacenet_hip)";
    g_rccl.version = ver;
    g_rccl.h = h;
    return SN_OK;
}
const char *rccl_err(int e) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "rccl error"; }
}  // namespace

}  // extern "C" (reopened below)
static void comm_destroy_impl(sn_ctx *c)
{
    if (c->rccl_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->rccl_comm);
    c->rccl_comm = nullptr;
}
extern "C" {

int sn_comm_unique_id(char *id128)
{
    if (!id128) return fail(SN_ERR_ARG, "null argument");
    int rc = rccl_load();
    if (rc != SN_OK) return rc;
    const int e = g_rccl.GetUniqueId(id128);
    if (e != 0) return fail(SN_ERR_COMM, "ncclGetUniqueId: %s", rccl_err(e));
    return SN_OK;
}

// Which RCCL the communicator entry points are bound to: the file (and whether the host process had it mapped already) and its ncclGetVersion code.
// Loads the library if nothing has yet. A process with TWO RCCL copies (torch's and this one's) shows up here, not as a hang at the first collective.
int sn_comm_info(char *file, int file_cap, int *version_code)
{
    int rc = rccl_load();
    if (rc != SN_OK) return rc;
    if (file && file_cap > 0) snprintf(file, (size_t)file_cap, "%s", g_rccl_file.c_str());
    if (version_code) *version_code = g_rccl.version;
    return SN_OK;
}

// ncclCommInitRank blocks until every rank of the group has called it. sn_comm_init_deadline bounds that wait: the call runs in a helper thread
// and the caller waits at most timeout_s for it; on a timeout the context stays without a communicator (SN_ERR_COMM), the helper thread - still
// inside RCCL, where nothing can interrupt it - is abandoned together with the few bytes of state it owns, and the context remains usable for
// everything but the exchange. sn_comm_init = the same without a deadline (the caller vouches that all ranks arrive).
// The 8-byte-per-rank exchanges of sn_allgatherv_* (counts, per-rank status) run through a buffer that exists from the moment the communicator does:
// an allocation that can fail must not sit between a rank and a collective its peers have already entered. [W slots | this rank's word]
static int comm_small_alloc(sn_ctx *c)
{
    if (c->comm_small) { dev_free_owned(c, c->comm_small); c->comm_small = nullptr; }
    int rc = dev_alloc(c, &c->comm_small, (size_t)(c->comm_world + 1) * 8);
    if (rc != SN_OK) { comm_destroy_impl(c); c->comm_world = 0; c->comm_rank = 0; }      // no communicator without it: the caller falls back before any collective
    return rc;
}

struct CommInitJob {
    std::mutex m; std::condition_variable cv;
    bool done = false; int err = 0; void *comm = nullptr; bool abandoned = false;
    int device = 0, world = 0, rank = 0; UniqueId uid;
};

int sn_comm_init_deadline(sn_ctx *c, int world, int rank, const char *id128, double timeout_s)
{
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(SN_ERR_ARG, "bad argument");
    int rc = rccl_load();
    if (rc != SN_OK) return rc;
    HIPCHK(hipSetDevice(c->device));
    if (c->rccl_comm) { g_rccl.CommDestroy(c->rccl_comm); c->rccl_comm = nullptr; }
    c->comm_world = 0; c->comm_rank = 0;
    typedef int (*init_fn)(void **, int, UniqueId, int);                   // ncclUniqueId is passed BY VALUE
    if (!(timeout_s > 0.0)) {
        UniqueId uid;
        memcpy(uid.b, id128, 128);
        const int e = ((init_fn)g_rccl.CommInitRank)(&c->rccl_comm, world, uid, rank);
        if (e != 0) { c->rccl_comm = nullptr; return fail(SN_ERR_COMM, "ncclCommInitRank: %s", rccl_err(e)); }
        c->comm_world = world; c->comm_rank = rank;
        return comm_small_alloc(c);
    }
    auto job = std::make_shared<CommInitJob>();
    job->device = c->device; job->world = world; job->rank = rank;
    memcpy(job->uid.b, id128, 128);
    std::thread([job]() {
        void *comm = nullptr;
        int e = (hipSetDevice(job->device) == hipSuccess) ? ((init_fn)g_rccl.CommInitRank)(&comm, job->world, job->uid, job->rank) : -1;
        std::unique_lock<std::mutex> lk(job->m);
        if (job->abandoned) {                                   // the caller gave up on us: nobody will ever use this communicator
            if (e == 0 && comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm);
            return;
        }
        job->err = e; job->comm = comm; job->done = true;
        job->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(job->m);
    if (!job->cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return job->done; })) {
        job->abandoned = true;
        return fail(SN_ERR_COMM, "ncclCommInitRank (rank %d of %d) did not return within %.0f s: not every rank joined, or the ranks' RCCL copies cannot reach each other "
                                 "(sn_comm_info names the library bound here)", rank, world, timeout_s);
    }
    if (job->err != 0) return fail(SN_ERR_COMM, "ncclCommInitRank: %s", job->err == -1 ? "hipSetDevice failed in the helper thread" : rccl_err(job->err));
    c->rccl_comm = job->comm;
    c->comm_world = world; c->comm_rank = rank;
    return comm_small_alloc(c);
}

int sn_comm_init(sn_ctx *c, int world, int rank, const char *id128) { return sn_comm_init_deadline(c, world, rank, id128, 0.0); }

// Every rank contributes n_local floats (device); global_dev receives world*n_local floats in rank order. Asynchronous
// on the context's stream (use sn_synchronize).
int sn_allgather_f32_dev(sn_ctx *c, const float *local_dev, size_t n_local, float *global_dev)
{
    if (!c || !local_dev || !global_dev) return fail(SN_ERR_ARG, "null argument");
    if (!c->rccl_comm) return fail(SN_ERR_STATE, "sn_comm_init has not been called");
    HIPCHK(hipSetDevice(c->device));
    ProfScope ps(c, "rccl_allgather", 0, (double)n_local * 4.0 * c->comm_world);
    const int e = g_rccl.AllGather(local_dev, global_dev, n_local, /*ncclFloat32*/ 7, c->rccl_comm, c->stream);
    if (e != 0) return fail(SN_ERR_COMM, "ncclAllGather: %s", rccl_err(e));
    return SN_OK;
}

// The same collective on the context's COMM stream, ordered behind everything submitted to the kernel stream so far, so that it overlaps the
// kernels submitted next (the all-gather of batch i runs under the CVC + CNN of batch i + 1). `slot` (0..7) names the completion event:
// sn_comm_wait(ctx, slot) makes the kernel stream wait for that collective - call it before local_dev / global_dev are written again.
// sn_synchronize waits for both streams.
int sn_allgather_f32_dev_overlap(sn_ctx *c, const float *local_dev, size_t n_local, float *global_dev, int slot)
{
    if (!c || !local_dev || !global_dev) return fail(SN_ERR_ARG, "null argument");
    if (slot < 0 || slot >= 8) return fail(SN_ERR_ARG, "sn_allgather_f32_dev_overlap: slot must be 0..7");
    if (!c->rccl_comm) return fail(SN_ERR_STATE, "sn_comm_init has not been called");
    HIPCHK(hipSetDevice(c->device));
    if (!c->comm_stream) HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    if (!c->comm_fork) HIPCHK(hipEventCreateWithFlags(&c->comm_fork, hipEventDisableTiming));
    if (!c->comm_ev[slot]) HIPCHK(hipEventCreateWithFlags(&c->comm_ev[slot], hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->comm_fork, c->stream));
    HIPCHK(hipStreamWaitEvent(c->comm_stream, c->comm_fork, 0));
    const int e = g_rccl.AllGather(local_dev, global_dev, n_local, /*ncclFloat32*/ 7, c->rccl_comm, c->comm_stream);
    if (e != 0) return fail(SN_ERR_COMM, "ncclAllGather: %s", rccl_err(e));
    HIPCHK(hipEventRecord(c->comm_ev[slot], c->comm_stream));
    return SN_OK;
}

int sn_comm_wait(sn_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= 8) return fail(SN_ERR_ARG, "sn_comm_wait: slot must be 0..7");
    if (!c->comm_ev[slot]) return SN_OK;                       // nothing was ever issued under this slot
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamWaitEvent(c->stream, c->comm_ev[slot], 0));
    return SN_OK;
}

// Variable-length all-gather of bytes ("gather sparse voxels ... counts then all-gather-v"): every rank contributes n_local bytes
// (device memory; n_local may be 0 and may differ from rank to rank), global_dev receives the contributions back to back in rank order, counts[r]
// (host, `world` entries) their sizes. Two RCCL all-gathers on the context's stream - the 8-byte counts, then the payloads padded to the largest -
// and one device-to-device copy per rank that closes the gaps. Synchronous (the host needs the counts to size the second step).
// EVERY rank issues the SAME sequence of collectives (counts, an 8-byte status word per rank, payloads) whatever its own arguments are: a destination that
// cannot hold the total is reported AFTER the payload all-gather has run (SN_ERR_ARG with counts[] filled in), so a rank that fails never leaves its peers inside a
// collective it does not take part in (the first version returned between the two all-gathers, and a caller that retried on its own then
// issued a counts all-gather against its peers' payload all-gather). sn_allgatherv_counts is the sizing query: the counts all-gather alone.
static int comm_stage_need(sn_ctx *c, size_t bytes)
{
    if (bytes <= c->comm_stage_cap) return SN_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->comm_stage) { dev_free_owned(c, c->comm_stage); c->comm_stage = nullptr; c->comm_stage_cap = 0; }
    int rc = dev_alloc(c, &c->comm_stage, bytes);
    if (rc != SN_OK) return rc;
    c->comm_stage_cap = bytes;
    return SN_OK;
}
// one 8-byte word per rank, gathered to the host of every rank (counts; per-rank status)
static int comm_gather_u64(sn_ctx *c, unsigned long long mine, unsigned long long *all, const char *what)
{
    const int W = c->comm_world;
    if (!c->comm_small) return fail(SN_ERR_STATE, "the communicator's exchange buffer is missing (sn_comm_init did not complete)");
    HIPCHK(hipMemcpyAsync(c->comm_small + (size_t)W * 8, &mine, 8, hipMemcpyHostToDevice, c->stream));
    const int e = g_rccl.AllGather(c->comm_small + (size_t)W * 8, c->comm_small, 8, /*ncclUint8*/ 1, c->rccl_comm, c->stream);
    if (e != 0) return fail(SN_ERR_COMM, "ncclAllGather (%s): %s", what, rccl_err(e));
    HIPCHK(hipMemcpyAsync(all, c->comm_small, (size_t)W * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SN_OK;
}
static int comm_gather_counts(sn_ctx *c, size_t n_local, unsigned long long *counts) { return comm_gather_u64(c, n_local, counts, "counts"); }

int sn_allgatherv_counts(sn_ctx *c, size_t n_local, unsigned long long *counts)
{
    if (!c || !counts) return fail(SN_ERR_ARG, "null argument");
    if (!c->rccl_comm) return fail(SN_ERR_STATE, "sn_comm_init has not been called");
    HIPCHK(hipSetDevice(c->device));
    return comm_gather_counts(c, n_local, counts);
}

int sn_allgatherv_bytes_dev(sn_ctx *c, const void *local_dev, size_t n_local, void *global_dev, size_t global_cap, unsigned long long *counts)
{
    if (!c || !counts || (n_local && !local_dev)) return fail(SN_ERR_ARG, "null argument");
    if (!c->rccl_comm) return fail(SN_ERR_STATE, "sn_comm_init has not been called");
    HIPCHK(hipSetDevice(c->device));
    const int W = c->comm_world;
    int rc = comm_gather_counts(c, n_local, counts);
    if (rc != SN_OK) return rc;
    unsigned long long total = 0, cap = 0;
    for (int r = 0; r < W; ++r) { total += counts[r]; cap = std::max(cap, counts[r]); }
    if (total == 0) return SN_OK;                                  // (the same on every rank: nobody enters the payload step)
    cap = (cap + 15) & ~15ull;
    // The staging buffer is sized by the LARGEST rank's payload, known only now - and its allocation is the one step between the two collectives that a
    // single rank can fail. Its outcome is exchanged (8 bytes per rank) before anyone enters the payload all-gather: all ranks go on, or all
    // ranks return the same error.
    const int stage_rc = comm_stage_need(c, (size_t)(W + 1) * cap);              // [W padded payloads | this rank's padded payload]
    const std::string stage_err = stage_rc != SN_OK ? g_err : std::string();
    std::vector<unsigned long long> flags((size_t)W, 0);
    rc = comm_gather_u64(c, stage_rc != SN_OK ? 1ull : 0ull, flags.data(), "status");
    if (rc != SN_OK) return rc;
    for (int r = 0; r < W; ++r)
        if (flags[(size_t)r])
            return fail(stage_rc != SN_OK ? stage_rc : SN_ERR_NOMEM, "sn_allgatherv_bytes_dev: rank %d could not stage %llu bytes%s%s - no rank entered the payload all-gather",
                        r, (unsigned long long)(W + 1) * cap, stage_err.empty() ? "" : ": ", stage_err.c_str());
    unsigned char *mine_pad = c->comm_stage + (size_t)W * cap;
    if (n_local) HIPCHK(hipMemcpyAsync(mine_pad, local_dev, n_local, hipMemcpyDeviceToDevice, c->stream));
    int e;
    {
        ProfScope ps(c, "rccl_allgatherv", 0, (double)cap * W);
        e = g_rccl.AllGather(mine_pad, c->comm_stage, cap, /*ncclUint8*/ 1, c->rccl_comm, c->stream);
    }
    if (e != 0) return fail(SN_ERR_COMM, "ncclAllGather (payload): %s", rccl_err(e));
    // rank-local checks only from here on: every collective of the call has been issued
    if (total > global_cap || !global_dev) {
        HIPCHK(hipStreamSynchronize(c->stream));
        return fail(SN_ERR_ARG, "sn_allgatherv_bytes_dev: %llu bytes gathered, the destination holds %zu (size it with sn_allgatherv_counts)", total,
                    global_dev ? global_cap : (size_t)0);
    }
    size_t off = 0;
    for (int r = 0; r < W; ++r) {
        if (counts[r]) HIPCHK(hipMemcpyAsync(static_cast<unsigned char *>(global_dev) + off, c->comm_stage + (size_t)r * cap, counts[r], hipMemcpyDeviceToDevice, c->stream));
        off += counts[r];
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return SN_OK;
}

// ---- raw device memory helpers (for hosts without a GPU array library) -------------------------------
void *sn_dev_alloc(sn_ctx *c, size_t bytes)
{
    if (!c) { fail(SN_ERR_ARG, "null context"); return nullptr; }
    (void)hipSetDevice(c->device);
    unsigned char *p = nullptr;
    if (dev_alloc(c, &p, bytes) != SN_OK) return nullptr;
    return p;
}
int sn_dev_free(sn_ctx *c, void *p) { if (!c) return fail(SN_ERR_ARG, "null context"); HIPCHK(hipSetDevice(c->device)); HIPCHK(hipStreamSynchronize(c->stream)); return dev_free_owned(c, p); }
int sn_memcpy_h2d(sn_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return fail(SN_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SN_OK;
}
int sn_memcpy_d2h(sn_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return fail(SN_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SN_OK;
}

int sn_mark(sn_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= 8) return fail(SN_ERR_ARG, "sn_mark: slot must be 0..7");
    HIPCHK(hipSetDevice(c->device));
    if (!c->marks[slot]) HIPCHK(hipEventCreateWithFlags(&c->marks[slot], hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->marks[slot], c->stream));
    return SN_OK;
}

int sn_memcpy_d2h_after(sn_ctx *c, int slot, void *dst, const void *src, size_t bytes)
{
    if (!c || !dst || !src) return fail(SN_ERR_ARG, "null argument");
    if (slot < 0 || slot >= 8 || !c->marks[slot]) return fail(SN_ERR_STATE, "sn_memcpy_d2h_after: slot %d was never marked", slot);
    HIPCHK(hipSetDevice(c->device));
    if (!c->copy_stream) HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamWaitEvent(c->copy_stream, c->marks[slot], 0));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(hipStreamSynchronize(c->copy_stream));
    return SN_OK;
}

#ifdef SN_DEBUG_HOOKS     // test-only twin library (Makefile target dbg): not in the product .so, not in the ABI header
// Test hook (not part of the ABI header): raw copy of an internal activation buffer ("cat": concat buffer, "ma": merge_conv_a output;
// both planes, layout of DESIGN.md section 3) for comparing the device's stored codes with oracle/net_emulation.py.
SN_API int sn_debug_tensor(sn_ctx *c, const char *name, void *host, size_t bytes)
{
    if (!c || !name || !host) return fail(SN_ERR_ARG, "null argument");
    const _Float16 *p = !strcmp(name, "cat") ? c->cat : (!strcmp(name, "ma") ? c->ma : nullptr);
    if (!p) return fail(SN_ERR_ARG, "unknown tensor %s", name);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(host, p, bytes, hipMemcpyDeviceToHost));
    return SN_OK;
}

// Test hook: the HOST half of pack_conv for one layer description (row-major (cout, cin, taps) weights, taps = ks^3 or ks^2 with k2d) - lets
// the address-sanitizer twin (make asan, tests/test_asan.py) run the fragment packer's index arithmetic for every layer shape and
// precision mode without a GPU. out[0..2] = sizes of the three packed arrays, out[3] = a byte checksum of the packed fragments.
SN_API int sn_debug_pack_host(int cin, int cout, int ks, int dil, int k2d, int nf, int nsplit, int cs8max, int split, const float *W, const float *beta,
                              const float *gamma, const float *mean, const float *inv_std, unsigned long long *out)
{
    if (!W || !beta || !gamma || !mean || !inv_std || !out) return fail(SN_ERR_ARG, "null argument");
    PackedConv L;
    L.name = "debug"; L.cin = cin; L.cout = cout; L.ks = ks; L.dil = dil; L.k2d = k2d;
    L.bridge = 1;                         // as the nets ask for them (granted to the f16x3 3x3 layers with uniform slabs only)
    std::vector<_Float16> h;
    std::vector<float> sc, sh;
    const int rc = pack_conv_host(L, W, beta, gamma, mean, inv_std, nf, nsplit, cs8max, split, nullptr, nullptr, h, sc, sh);
    if (rc != SN_OK) return rc;
    unsigned long long sum = 0;
    const unsigned char *b = reinterpret_cast<const unsigned char *>(h.data());
    for (size_t i = 0; i < h.size() * sizeof(_Float16); ++i) sum = sum * 1099511628211ull + b[i];
    out[0] = h.size(); out[1] = sc.size(); out[2] = sh.size(); out[3] = sum;
    return SN_OK;
}

// Test hook (not part of the ABI header): the host-side 6-bit encoder the weight packer uses, so that a CPU test can pin it against the
// format's decode table (tests/test_abi.py) - the device side of the format is pinned by tools/probe/fp6_probe.hip.
SN_API int sn_debug_mx6_encode(float v, int fmt) { return (fmt == 2 || fmt == 3) ? (int)mx6_encode(v, fmt) : -1; }

// Diagnostic builds only (-DSN_TIMING=1, conv3d_mfma.h): per-layer shader-clock totals {kernel, vmcnt wait, barrier wait, pieces} summed over
// waves, in layer-bit order (names: sn_synchronize's message order). Not part of the ABI header; reads and clears the slots.
SN_API int sn_debug_timing(sn_ctx *c, unsigned long long *out, int n_layers, char *names, int names_cap)
{
    if (!c || !out || !c->d_num) return fail(SN_ERR_ARG, "null argument");
    HIPCHK(hipStreamSynchronize(c->stream));
    n_layers = std::min(n_layers, 32);
    HIPCHK(hipMemcpy(out, c->d_num + 2, sizeof(unsigned long long) * 4 * n_layers, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(c->d_num + 2, 0, sizeof(unsigned) * 32 * 8));
    std::string all;
    for (auto &nm : c->num_names) all += nm + ",";
    if (names && names_cap > 0) { strncpy(names, all.c_str(), names_cap - 1); names[names_cap - 1] = 0; }
    return SN_OK;
}
// workgroup 0 of the last EPI_FINAL launch: [2048 pieces][8 waves]{arrival at the barrier, release} shader clocks
SN_API int sn_debug_trace(sn_ctx *c, long long *out)
{
    if (!c || !out || !c->d_num) return fail(SN_ERR_ARG, "null argument");
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->d_num + 2 + 32 * 8, sizeof(long long) * 2048 * 8 * 2, hipMemcpyDeviceToHost));
    return SN_OK;
}

#endif  // SN_DEBUG_HOOKS

// ---- profiling ----------------------------------------------------------------------------------------
int sn_profile_enable(sn_ctx *c, int on)
{
    if (!c) return fail(SN_ERR_ARG, "null context");
    int rc = prof_drain(c);
    c->prof_on = on != 0;
    return rc;
}
int sn_profile_reset(sn_ctx *c)
{
    if (!c) return fail(SN_ERR_ARG, "null context");
    int rc = prof_drain(c);
    for (auto &st : c->prof_stats) { st.ms = 0; st.launches = 0; st.flops = 0; st.bytes = 0; }
    return rc;
}
int sn_profile_count(sn_ctx *c)
{
    if (!c) return fail(SN_ERR_ARG, "null context");
    if (prof_drain(c) != SN_OK) return SN_ERR_HIP;
    return (int)c->prof_stats.size();
}
int sn_profile_get(sn_ctx *c, int idx, char *name, int name_cap, double *ms_total, int64_t *launches, double *flops, double *bytes)
{
    if (!c) return fail(SN_ERR_ARG, "null context");
    int rc = prof_drain(c);
    if (rc != SN_OK) return rc;
    if (idx < 0 || idx >= (int)c->prof_stats.size()) return fail(SN_ERR_ARG, "profile index out of range");
    const ProfStat &st = c->prof_stats[idx];
    if (name && name_cap > 0) { strncpy(name, st.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (ms_total) *ms_total = st.ms;
    if (launches) *launches = st.launches;
    if (flops) *flops = st.flops;
    if (bytes) *bytes = st.bytes;
    return SN_OK;
}

}  // extern "C"
