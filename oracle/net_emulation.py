"""oracle/net_emulation.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU MODEL OF THE HIP PATH'S ARITHMETIC (not of the reference): the same network as net_oracle.forward_torch, computed in float64
but with every rounding the device applies in its two MX-assisted modes, so that a parity test can tell "the kernel does what its
design says" (device vs this model: ~1e-5) apart from "the design is accurate enough" (this model vs the exact oracle: ~1e-4).

What is modelled (surfacenet_amd/csrc/mx_format.h, conv3d_mfma.h, sn_api.hip pack_conv / sn_load_weights):
  * exact power-of-two renormalisation: a ReLU layer stores y * 2^oe[c], oe = -ilogb(max(|gamma|, |beta|)); its consumer's weights are
    multiplied by 2^-oe[c_in] and every weight row by 2^row_exp, row_exp = -ilogb(max |row|) — exact, but it decides where the 6-bit
    codes saturate or flush;
  * storage "x3": value = hi + fp16(value - hi), hi = fp16(value);   storage "m6": hi + q6((value - hi) * 2^11 * 2^s) / 2^(11+s);
  * product "m6" (f16m8 arithmetic): w*x = wh*xh  +  q6b(wl * 2^11) / 2^11 * q6(xh * 2^s) / 2^s  +  q6b(wh) * lo6,
    q6 = fp6 e2m3 (round to nearest even, saturating at 7.5, subnormal step 0.125), q6b = the same with one power-of-two scale per
    weight block (one output channel x 8 input channels x two consecutive (tap, group) units of the layer's K stream - group-major, taps
    inside a group; bridge pieces let a block span two groups - for the 3x3x3 layers, channel-group pairs inside a 5-group slab for the
    1x1x1 layers);
  * product "x3": exact on the stored values (the device drops lo*lo, 2^-22 relative, and accumulates in fp32).
How close can device and model be? Stored tensors were compared directly (sn_debug_tensor, round 2, s=16): the concat buffer's fp16 plane
differs from the model's in 0.24 % of its elements (fp32 vs fp64 upstream: one-ulp flips) and its lo codes in 2.7 %; a flipped (hi, lo)
pair still represents the same value to 2^-15, which is also the size of the 6-bit quantisation error - so merge_conv_a's output, computed
from those codes, already differs in 2.9 % of its fp16 values and 11 % of its lo codes. The final difference device - model (rms 0.5 of
the device's error, correlation 0.82 between the two error fields) is this sensitivity, not a modelling gap: every parameter variant
tried (premultipliers, lo exponent, block shape, renormalisation on / off) lowers the correlation.
mode "f16x3" (the default): everything "x3" except the concat buffer and merge_conv_a's output (storage m6, premultipliers s = 2 / 0)
and merge_conv_a / merge_conv_b (product m6) and - round 5 - the dilated chain conv4_1 .. conv4_3 (product m8: fp8 e4m3 codes of activations * 2^2 and of
the plain row-normalised weights, no block scales); conv1_3 / conv2_3 feed their side conv and pool from unrounded registers.
mode "f16m8": every tensor m6, every conv product m6 (the network input with s = -5).
Round 5: a per-layer CORRECTION-FORMAT TABLE for the default mode (`table`, layer name -> "x3" | "m6" | "m8"; LAYER_FORMATS_DEFAULT is what
the library ships) answers "which 3x3x3 layers may leave the three-fp16-MFMA arithmetic at an unchanged tolerance" without a GPU:
  * "m8": the same two-term correction as "m6" on fp8 e4m3 codes (2 MFMA units per product instead of 1.5 / 3): q8 = e4m3, round to nearest
    even, saturating at 448, subnormal step 2^-9; lo parts premultiplied by 2^12; the same 32-element weight blocks with one power-of-two scale
    (block maximum in (224, 448]); activations with a static premultiplier 2^s8 (default 0).
  A tensor is stored in the format its 3x3x3 reader asks for (plus the fp16 lo plane where a 1x1x1 side convolution reads it as well)."""
import numpy as np

from oracle import net_oracle

LO_EXP = 11
LO_EXP8 = 12
S_ACT, S_CAT, S_X0 = 0, 2, -5
S8_ACT = 0                   # premultiplier of the fp8 code planes (mx_format.h SN_MX_S_C4)
# correction format per 3x3x3 layer in the default ("f16x3") mode, as shipped: x3 = three fp16 MFMAs, m6 = fp6 e2m3 MX step, m8 = fp8 e4m3 MX step
LAYER_FORMATS_DEFAULT = {"conv1_1": "x3", "conv1_2": "x3", "conv1_3": "x3", "conv2_1": "x3", "conv2_2": "x3", "conv2_3": "x3",
                         "conv3_1": "x3", "conv3_2": "x3", "conv3_3": "x3", "conv4_1": "m8", "conv4_2": "m8", "conv4_3": "m8",
                         "merge_conv_a": "m6", "merge_conv_b": "m6"}
LAYER_FORMATS_R4 = dict(LAYER_FORMATS_DEFAULT, conv4_1="x3", conv4_2="x3", conv4_3="x3")      # rounds 2-4: only the merge layers on the MX step
S_C4 = -2                    # premultiplier of the code planes conv4_2 / conv4_3 read (conv4_1's and conv4_2's outputs; mx_format.h SN_MX_S_C4)
S6_OF_DEFAULT = {"conv3_3": S_C4, "conv4_1": S_C4, "conv4_2": S_C4}      # (conv3_3's: for what-if tables that put conv4_1 on an MX format as well)


def _ilogb(a):
    a = np.asarray(a, dtype=np.float64)
    out = np.zeros(a.shape, dtype=np.int64)
    nz = a > 0
    out[nz] = np.floor(np.log2(a[nz])).astype(np.int64)
    # guard against log2 rounding at exact powers of two
    out[nz] += (np.ldexp(1.0, out[nz] + 1) <= a[nz]).astype(np.int64)
    out[nz] -= (np.ldexp(1.0, out[nz]) > a[nz]).astype(np.int64)
    return out


def forward_emulated(X, values, w=None, n_vp=1, mode="f16x3", table=None, s8_act=S8_ACT, s6_of=None, m8_block_scale=False):
    import torch
    import torch.nn.functional as F
    assert mode in ("f16x3", "f16m8")
    td = torch.float64
    P = net_oracle.params_to_dict(values)
    full = mode == "f16m8"
    fmt_of = dict(LAYER_FORMATS_DEFAULT)
    s6_tab = {} if full else dict(S6_OF_DEFAULT)
    s6_tab.update(s6_of or {})
    if table:
        assert not full and set(table) <= set(fmt_of) and set(table.values()) <= {"x3", "m6", "m8", "b6"}, table
        fmt_of.update(table)

    def f16(t):
        return t.to(torch.float16).to(td)

    def q6(v):
        """fp6 e2m3 of v (already premultiplied): RNE, saturating, subnormals."""
        a = v.abs().clamp(max=7.5)
        step = torch.where(a < 2, torch.full_like(a, 0.125), torch.where(a < 4, torch.full_like(a, 0.25), torch.full_like(a, 0.5)))
        return torch.sign(v) * torch.round(a / step) * step

    def q8(v):
        """fp8 e4m3 (OCP e4m3fn) of v: RNE, saturating at 448, subnormal step 2^-9 below 2^-6."""
        a = v.abs().clamp(max=448.0)
        e = torch.floor(torch.log2(torch.clamp(a, min=2.0 ** -6)))
        e = e + (torch.exp2(e + 1) <= a).to(td) - (torch.exp2(e) > torch.clamp(a, min=2.0 ** -6)).to(td)       # (log2 rounding at exact powers of two)
        step = torch.exp2(e - 3)
        return (torch.sign(v) * torch.round(a / step) * step).clamp(min=-448.0, max=448.0)

    def qb6(v):
        """bf6 e3m2 of v: RNE, saturating at 28, subnormal step 2^-4 below 2^-2."""
        a = v.abs().clamp(max=28.0)
        e = torch.floor(torch.log2(torch.clamp(a, min=0.25)))
        e = e + (torch.exp2(e + 1) <= a).to(td) - (torch.exp2(e) > torch.clamp(a, min=0.25)).to(td)
        step = torch.exp2(e - 2)
        return (torch.sign(v) * torch.round(a / step) * step).clamp(min=-28.0, max=28.0)

    QF = {"m6": (q6, LO_EXP, 7.5), "m8": (q8, LO_EXP8, 448.0), "b6": (qb6, LO_EXP, 28.0)}

    class T:  # an activation tensor: the producer's fp32 result in ORIGINAL units + how its readers see what the device stores
        def __init__(self, v, oe, s6=S_ACT, s8=None, via16=False):
            self.via16 = via16
            self.oe = torch.from_numpy(np.asarray(oe, dtype=np.float64)).view(1, -1, 1, 1, 1)      # stored = v * 2^oe
            self.r = v * torch.exp2(self.oe)
            self.hi = f16(self.r)
            self.s = {"m6": s6, "b6": s6, "m8": s8_act if s8 is None else s8}

        def lo(self, fmt):                       # the stored residual as a reader of format fmt reconstructs it
            if fmt == "x3":
                return f16(self.r - self.hi)
            q, le, _ = QF[fmt]
            k = 2.0 ** (le + self.s[fmt])
            res = self.r - self.hi
            return q((f16(res) if self.via16 else res) * k) / k

        def hi_q(self, fmt):                     # the code of hi that multiplies the weights' lo parts
            q, _, _ = QF[fmt]
            k = 2.0 ** self.s[fmt]
            return q(self.hi * k) / k

        def v(self, fmt):                        # value in original units
            return (self.hi + self.lo(fmt)) / torch.exp2(self.oe)

    def block_exp(amax, vmax):
        am = amax.numpy()
        E = _ilogb(am / vmax)
        E = E + (np.ldexp(am, -E) > vmax)
        E[am == 0] = 0
        return torch.from_numpy(E.astype(np.float64))

    def block_q(wh, wl, kind, fmt):
        """wh, wl*2^LO: (O, Cp, T) renormalised weights, Cp = cin padded to 8. -> their block-scaled low-precision values."""
        q, _, vmax = QF[fmt]
        O, Cp, Tn = wh.shape
        G = Cp // 8
        a, b = wh.reshape(O, G, 8, Tn), wl.reshape(O, G, 8, Tn)
        if Tn > 1:
            # 3x3x3: one channel group per slab; K is the stream of (tap, group) units, group-major - since the bridge pieces of round 3 a
            # slab continues where its predecessor stopped, so a block = two CONSECUTIVE units of the whole layer's stream (the pair
            # (tap 26 of group g, tap 0 of group g + 1) included); only the very end of the stream is padded
            U = G * Tn
            Up = U + (U & 1)
            def stream(z):                           # (O, G, 8, Tn) -> (O, Up / 2, 2, 8)
                z = z.permute(0, 1, 3, 2).reshape(O, U, 8)
                return torch.cat([z, torch.zeros(O, Up - U, 8, dtype=td)], dim=1).reshape(O, Up // 2, 2, 8)
            a, b = stream(a), stream(b)
            amax = torch.maximum(a.abs().amax(dim=(2, 3)), b.abs().amax(dim=(2, 3)))                   # (O, Up/2)
            E = block_exp(amax, vmax).view(O, Up // 2, 1, 1)
            back = lambda z: z.reshape(O, Up, 8)[:, :U].reshape(O, G, Tn, 8).permute(0, 1, 3, 2).reshape(O, Cp, Tn)
            return back(q(a / 2.0 ** E) * 2.0 ** E), back(q(b / 2.0 ** E) * 2.0 ** E)
        # 1x1x1: slabs of up to 5 channel groups (tile_for), blocks = consecutive group pairs inside a slab
        aq, bq = torch.zeros_like(a), torch.zeros_like(b)
        g0 = 0
        while g0 < G:
            n = min(5, G - g0)
            for j in range(0, n, 2):
                sl = slice(g0 + j, min(g0 + j + 2, g0 + n))
                amax = torch.maximum(a[:, sl].abs().amax(dim=(1, 2, 3)), b[:, sl].abs().amax(dim=(1, 2, 3)))
                E = block_exp(amax, vmax).view(O, 1, 1, 1)
                aq[:, sl], bq[:, sl] = q(a[:, sl] / 2.0 ** E) * 2.0 ** E, q(b[:, sl] / 2.0 ** E) * 2.0 ** E
            g0 += n
        return aq.reshape(O, Cp, Tn), bq.reshape(O, Cp, Tn)

    def conv(x, name, kind, act, prod, s6_out=S_ACT, raw_out=False):
        """x: T (or a raw tensor for unrounded register inputs); prod: the layer's arithmetic. Returns T [or the fp32 result in original units]."""
        p = P[name]
        W = p["W"].astype(np.float64)
        if kind in ("dil3", "dil1"):
            W = np.transpose(W, (1, 0, 2, 3, 4))
        O, C, k = W.shape[0], W.shape[1], W.shape[2]

        def cv(xx, ww):
            ww = ww.reshape(O, -1, k, k, k)[:, :C]
            if kind == "dil3":
                return F.conv3d(F.pad(xx, (2,) * 6), ww, dilation=2)
            return F.conv3d(xx, ww, padding=k // 2)

        raw = not isinstance(x, T)
        oe_in = torch.zeros(1, C, 1, 1, 1, dtype=td) if raw else x.oe
        Wt = torch.from_numpy(np.ascontiguousarray(W)).to(td).reshape(O, C, -1)
        if prod == "x3":
            xv = x if raw else x.v("x3")
            wh = f16(Wt)
            y = cv(xv, wh + f16(Wt - wh))
        else:
            _, le, _ = QF[prod]
            # renormalised weights: W' = W * 2^-oe_in[c] * 2^row_exp[o]
            Wr = Wt / torch.exp2(oe_in.view(1, C, 1))
            row = torch.from_numpy(-_ilogb(Wr.abs().amax(dim=(1, 2)).numpy()).astype(np.float64)).view(O, 1, 1)
            Wr = Wr * torch.exp2(row)
            Cp = (C + 7) // 8 * 8
            Wr = torch.cat([Wr, torch.zeros(O, Cp - C, Wr.shape[2], dtype=td)], dim=1)
            wh = f16(Wr)
            if prod == "m8" and not m8_block_scale:      # fp8 codes of the row-normalised weights as they are (e4m3 has the exponent range): no block scale
                whq, wlq = q8(wh), q8((Wr - wh) * 2.0 ** le)
            else:
                whq, wlq = block_q(wh, (Wr - wh) * 2.0 ** le, kind, prod)
            wlq = wlq / 2.0 ** le
            # all three terms in renormalised units, then undo the row exponent (the input exponent is inside W')
            y = (cv(x.hi, wh) + cv(x.hi_q(prod), wlq) + cv(x.lo(prod), whq)) / torch.exp2(row.view(1, O, 1, 1, 1))
        scale = (p["gamma"].astype(np.float64) * p["inv_std"].astype(np.float64)).astype(np.float32)
        shift = (p["beta"].astype(np.float64) - p["mean"].astype(np.float64) * scale.astype(np.float64)).astype(np.float32)
        y = y.to(torch.float32) * torch.from_numpy(scale).view(1, -1, 1, 1, 1) + torch.from_numpy(shift).view(1, -1, 1, 1, 1)
        y = (torch.relu(y) if act == "relu" else torch.sigmoid(y)).to(td)
        if raw_out:
            return y
        oe = np.zeros(O)
        if act == "relu":
            m = np.maximum(np.abs(p["gamma"].astype(np.float64)), np.abs(p["beta"].astype(np.float64)))
            oe = np.clip(-_ilogb(m), -60, 60).astype(np.float64)
            oe[~(m > 0)] = 0
        return T(y, oe, s6_tab.get(name, s6_out))       # s6_tab: premultiplier of a layer's OUTPUT tensor as its fp6 readers see it

    def up(x, name, f):
        k = P[name]["W"].shape[2]
        Wk = torch.from_numpy(np.ascontiguousarray(P[name]["W"])).to(td)
        B, C = x.shape[:2]
        z = torch.zeros((B, C, x.shape[2] * f, x.shape[3] * f, x.shape[4] * f), dtype=td)
        z[:, :, ::f, ::f, ::f] = x
        return F.conv3d(z.reshape(B * C, 1, *z.shape[2:]), Wk, padding=k // 2).reshape(B, C, *z.shape[2:])

    A = "m6" if full else "x3"          # arithmetic / storage of the 1x1x1 layers and side maps upstream of the concat buffer
    f = (lambda name: "m6") if full else (lambda name: fmt_of[name])
    x = T(torch.from_numpy(np.ascontiguousarray(X)).to(td), np.zeros(X.shape[1]), S_X0, 0)
    c11 = conv(x, "conv1_1", "conv3", "relu", f("conv1_1"))
    c12 = conv(c11, "conv1_2", "conv3", "relu", f("conv1_2"))
    c13 = conv(c12, "conv1_3", "conv3", "relu", f("conv1_3"))
    y13 = c13.r / torch.exp2(c13.oe)
    # conv1_3 / conv2_3: side conv and pool run on the unrounded registers; the pooled tensor is stored in the format its reader asks for
    s1 = conv(y13, "side_op1", "conv1", "sigmoid", "x3", raw_out=True)
    p1 = T(F.max_pool3d(y13, 2, 2), c13.oe.view(-1).numpy(), s6_tab.get("pool1", S_ACT))
    c21 = conv(p1, "conv2_1", "conv3", "relu", f("conv2_1"))
    c22 = conv(c21, "conv2_2", "conv3", "relu", f("conv2_2"))
    c23 = conv(c22, "conv2_3", "conv3", "relu", f("conv2_3"))
    y23 = c23.r / torch.exp2(c23.oe)
    s2 = T(conv(y23, "side_op2", "conv1", "sigmoid", "x3", raw_out=True), np.zeros(16))
    p2 = T(F.max_pool3d(y23, 2, 2), c23.oe.view(-1).numpy(), s6_tab.get("pool2", S_ACT))
    c31 = conv(p2, "conv3_1", "conv3", "relu", f("conv3_1"))
    c32 = conv(c31, "conv3_2", "conv3", "relu", f("conv3_2"))
    c33 = conv(c32, "conv3_3", "conv3", "relu", f("conv3_3"))
    s3 = conv(c33, "side_op3", "conv1", "sigmoid", A)
    c41 = conv(c33, "conv4_1", "dil3", "relu", f("conv4_1"))
    c42 = conv(c41, "conv4_2", "dil3", "relu", f("conv4_2"))
    c43 = conv(c42, "conv4_3", "dil3", "relu", f("conv4_3"))
    s4 = conv(c43, "side_op4", "dil1", "sigmoid", A)
    cat_v = torch.cat([s1, up(s2.v(A), "side_op2_deconv", 2), up(s3.v(A), "side_op3_deconv", 4), up(s4.v(A), "side_op4_deconv", 4)], dim=1)
    cat = T(cat_v, np.zeros(64), S_CAT)
    ma = conv(cat, "merge_conv_a", "conv3", "relu", f("merge_conv_a"))
    mb = conv(ma, "merge_conv_b", "conv3", "relu", f("merge_conv_b"), raw_out=True)                    # stays in fp32 registers
    p3 = P["merge_conv3"]
    w3 = torch.from_numpy(np.ascontiguousarray(p3["W"].astype(np.float64))).to(td)
    y = F.conv3d(mb, w3)
    scale = (p3["gamma"].astype(np.float64) * p3["inv_std"].astype(np.float64)).astype(np.float32)
    shift = (p3["beta"].astype(np.float64) - p3["mean"].astype(np.float64) * scale.astype(np.float64)).astype(np.float32)
    out = torch.sigmoid(y.to(torch.float32) * float(scale[0]) + float(shift[0])).to(td)
    unf = out.numpy().astype(np.float64)
    s = X.shape[-1]
    unfused = unf.reshape(-1, n_vp, s, s, s)
    return net_oracle.fuse(unfused, w, n_vp), unfused
