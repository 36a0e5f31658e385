"""The weight layouts of lssvc_amd/csrc/weight_prep.cpp restated with torch ops (the round-1..3 implementation of
lssvc_amd/weights.py, kept as the checker of tests/test_host_logic.py::test_weight_prep_*): every function documents one layout
of include/lssvc_hip.h; the product path calls lssvc_prepare_weights (C++) instead."""
import math
import torch
import torch.nn.functional as F

CK = 8


def _pad_to(n, m):
    return (n + m - 1) // m * m


def layout_conv(w, bias, splits, pixel_shuffle):
    """w: (Cout, Cin, KH, KW) cpu fp32 -> (w_prepared, bias_prepared, Cout, M_pad)."""
    cout, cin, kh, kw = w.shape
    assert sum(splits) == cin, (splits, cin)
    if pixel_shuffle:
        cps = cout // 4
        w = w.reshape(cps, 4, cin, kh, kw).permute(1, 0, 2, 3, 4).reshape(cout, cin, kh, kw)
        if bias is not None:
            bias = bias.reshape(cps, 4).t().reshape(cout)
    m_pad = _pad_to(cout, 16)
    segs, a = [], 0
    for c in splits:
        s = w[:, a:a + c]
        segs.append(F.pad(s, (0, 0, 0, 0, 0, _pad_to(c, CK) - c)))
        a += c
    wp = torch.cat(segs, dim=1)
    wp = F.pad(wp, (0, 0, 0, 0, 0, 0, 0, m_pad - cout))
    nchunk = wp.shape[1] // CK
    wp = wp.reshape(m_pad, nchunk, CK, kh, kw).permute(1, 3, 4, 0, 2).contiguous()
    bp = torch.zeros(m_pad, dtype=torch.float32)
    if bias is not None:
        bp[:cout] = bias
    return wp, bp, cout, m_pad


F16X3_WEIGHT_EXP = 12      # max|w * 2^e| in [2^11, 2^12): far from fp16's 65504, lo parts normal down to |w'| ~ 0.125


def layout_conv_f16x3(w, splits, pixel_shuffle):
    """fp16 hi/lo planes for the f16x3 conv mode: (2, chunk16, KH, KW, M_pad, 16) fp16 of w' = w * 2^e,
    hi = fp16(w'), lo = fp16(w' - hi); same concat-segment / pixel-shuffle / M padding rules as layout_conv,
    16-channel chunks. Returns (planes, 2^-e). e is picked per layer so that max|w'| lands in [2^11, 2^12]: the
    lo parts of typical weights (|w| ~ 1e-2) would otherwise be fp16 subnormals and lose up to 10 of their 11
    bits; the scaling is a power of two, so it is exact and the kernel undoes it exactly on the accumulators."""
    cout, cin, kh, kw = w.shape
    if pixel_shuffle:
        cps = cout // 4
        w = w.reshape(cps, 4, cin, kh, kw).permute(1, 0, 2, 3, 4).reshape(cout, cin, kh, kw)
    m_pad = _pad_to(cout, 16)
    segs, a = [], 0
    for c in splits:
        segs.append(F.pad(w[:, a:a + c], (0, 0, 0, 0, 0, _pad_to(c, 16) - c)))
        a += c
    wp = F.pad(torch.cat(segs, dim=1), (0, 0, 0, 0, 0, 0, 0, m_pad - cout))
    wp = wp.reshape(m_pad, wp.shape[1] // 16, 16, kh, kw).permute(1, 3, 4, 0, 2).contiguous()
    wmax = float(wp.abs().max())
    e = 0 if wmax == 0.0 or not math.isfinite(wmax) else max(-14, min(24, F16X3_WEIGHT_EXP - math.frexp(wmax)[1]))
    wp = wp * (2.0 ** e)
    hi = wp.half()
    lo = (wp - hi.float()).half()
    return torch.stack([hi, lo], 0).contiguous(), 2.0 ** -e


def _f16x3_planes(w):
    """(w * 2^e split into fp16 hi / lo, flattened and concatenated [hi | lo]; 2^-e) -- see layout_conv_f16x3."""
    wmax = float(w.abs().max())
    e = 0 if wmax == 0.0 or not math.isfinite(wmax) else max(-14, min(24, F16X3_WEIGHT_EXP - math.frexp(wmax)[1]))
    w = w * (2.0 ** e)
    hi = w.half()
    lo = (w - hi.float()).half()
    return torch.cat([hi.reshape(-1), lo.reshape(-1)]).contiguous(), 2.0 ** -e


def _chained_k(n_frag_pairs):
    """Channel index of K position (pair p, k = 8g + j) when a B operand is assembled from two accumulator
    fragments of the previous GEMM (csrc/ffn_f16x3.hip): fragment 2p + (j >> 2), row 4g + (j & 3)."""
    k = torch.arange(32)
    g, j = k // 8, k % 8
    p = torch.arange(n_frag_pairs)[:, None]
    return (2 * p + (j >> 2)[None, :]) * 16 + (4 * g + (j & 3))[None, :]            # (pairs, 32)


def layout_ffn_f16x3(w1, w2):
    """ConvFFN weights (hidden, C, 1, 1) / (C, hidden, 1, 1) -> the two LDS images lssvc_ffn_f16x3 stages:
    W1 [t][f][s][16][32] (hidden fragment 2t+f, K-step s over the C channels in chained order) and
    W2 [t][m][16][32] (output fragment m, K = the 32 hidden channels of pair t in chained order)."""
    hidden, c = w1.shape[0], w1.shape[1]
    assert c % 16 == 0 and hidden % 32 == 0 and tuple(w2.shape[:2]) == (c, hidden)
    cf, t = c // 16, hidden // 32
    s = (cf + 1) // 2
    w1p = F.pad(w1.reshape(hidden, c), (0, 32 * s - c))                              # (hidden, 32 s) zero-padded
    ic = _chained_k(s)                                                               # (s, 32)
    a = w1p[:, ic]                                                                   # (hidden, s, 32)
    a = a.reshape(t, 2, 16, s, 32).permute(0, 1, 3, 2, 4)                            # [t][f][s][i][k]
    hc = _chained_k(t)                                                               # (t, 32)
    b = w2.reshape(c, hidden)[:, hc]                                                 # (c, t, 32)
    b = b.reshape(cf, 16, t, 32).permute(2, 0, 1, 3)                                 # [t][m][i][k]
    return _f16x3_planes(a.contiguous()), _f16x3_planes(b.contiguous())


def layout_pw_natural_f16x3(w):
    """A 1x1 conv weight (Cout, Cin, 1, 1) as [m][s][16][32] fragments in natural K order (leading conv of
    lssvc_ffn_f16x3)."""
    cout, cin = w.shape[0], w.shape[1]
    assert cout % 16 == 0
    s = (cin + 31) // 32
    a = F.pad(w.reshape(cout, cin), (0, 32 * s - cin)).reshape(cout // 16, 16, s, 32).permute(0, 2, 1, 3)
    return _f16x3_planes(a.contiguous())


def conv_t_as_conv(w, bias, stride):
    """ConvTranspose2d(k=3, padding=1[, stride=2, output_padding=1]) weight (Cin, Cout, 3, 3) ->
    (equivalent conv weight OIHW, bias, KH, pad, pixel_shuffle)."""
    cin, cout = w.shape[0], w.shape[1]
    if stride == 1:
        return w.flip(2, 3).permute(1, 0, 2, 3).contiguous(), bias, 1, False
    # out[2i+a, 2j+b] = sum_{dy,dx in {0,1}} in[i+dy, j+dx] * w[:, :, ky(a,dy), kx(b,dx)]
    tap = {(0, 0): 1, (1, 0): 2, (1, 1): 0}            # (phase, delta) -> kernel index; (0,1) has none
    w2 = torch.zeros(4, cout, cin, 2, 2, dtype=w.dtype)
    for a in (0, 1):
        for b in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    if (a, dy) in tap and (b, dx) in tap:
                        w2[a * 2 + b, :, :, dy, dx] = w[:, :, tap[(a, dy)], tap[(b, dx)]].t()
    # rows are (q, co)-major == the kernel's pixel-shuffle order m = q*Cout + co (no further permutation)
    w2 = w2.reshape(4 * cout, cin, 2, 2)
    b2 = bias.repeat(4)
    return w2.contiguous(), b2.contiguous(), 0, True


_REPARAM_OFFSET = 2 ** -18
_PEDESTAL = _REPARAM_OFFSET ** 2
_BETA_BOUND = (1e-6 + _REPARAM_OFFSET ** 2) ** 0.5
