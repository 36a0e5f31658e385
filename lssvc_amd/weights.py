"""Checkpoint handling: strict validation against the reference state-dict layout and one-time
re-layout of every tensor into what the HIP kernels consume.

Layouts produced here (see include/lssvc_hip.h):
  conv     OIHW fp32  ->  [chunk][ky][kx][m][8]   (input channels of every concatenated input segment
                           zero-padded to a multiple of 8, output channels padded to 16; for a
                           sub-pixel conv the output axis is permuted to (dy,dx)-major so that the
                           kernel's PixelShuffle store is a plain float4 per lane)
  convT    (Cin,Cout,3,3) -> an equivalent plain conv: stride 1 = flipped 3x3; stride 2 (pad 1,
                           output_padding 1) = 2x2 conv producing the 4 output phases + pixel shuffle
  GDN      beta/gamma de-reparametrised once (gdn.py:31-33 / video_net_component.py:86-93) into a
           1x1 conv (weight gamma, bias beta) applied to x^2 with a fused x*rsqrt / x*sqrt / x/sqrt epilogue
  depthwise (C,1,3,3) -> [9][C]
  BitEstimator / EntropyBottleneck parameters -> [rows][C] tables with softplus / tanh pre-applied
"""
import math
import torch
import torch.nn.functional as F

from .synth import load_manifest

CK = 8


class CheckpointError(RuntimeError):
    pass


def strip_module_prefix(sd):
    """'module.' prefix removal (IntraSS.py:193-198, LSSVC_net.py:141-149) and optional 'state_dict' unwrap."""
    if "state_dict" in sd and not torch.is_tensor(sd["state_dict"]):
        sd = sd["state_dict"]
    return {(k[7:] if k[:7] == "module." else k): v for k, v in sd.items()}


def validate(sd, manifest_name, ignore=(), resizable=()):
    """Same contract as nn.Module.load_state_dict(strict=True): every expected key present with the
    expected shape, nothing unexpected. `resizable` keys (CDF buffers) may have any shape."""
    want = {e["key"]: tuple(e["shape"]) for e in load_manifest(manifest_name)}
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want and k not in ignore]
    bad = [k for k in want if k in sd and not any(k.endswith(r) for r in resizable) and tuple(sd[k].shape) != want[k]]
    if missing or unexpected or bad:
        raise CheckpointError("Error(s) in loading state_dict: missing %s; unexpected %s; size mismatch %s"
                              % (missing[:5], unexpected[:5], bad[:5]))


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


class WeightStore:
    """Lazily prepared, cached device copies of one model's tensors."""

    def __init__(self, sd, device):
        self.sd = {k: (v.detach().to("cpu", torch.float32) if v.is_floating_point() else v.detach().cpu())
                   for k, v in sd.items()}
        self.device = device
        self._cache = {}
        self.regions = {}               # device pointer -> (bytes, host tensor) of every prepared tensor (plan_compiler.py stores them)
        self.force_f32 = set()          # layers the fp16 range audit moved to the exact fp32 conv kernel (hip_ops.RangeAudit)

    def has(self, key):
        return key in self.sd

    def raw(self, key):
        return self.sd[key]

    def _dev(self, t):
        host = t.contiguous()
        dev = host.to(self.device)
        self.regions[dev.data_ptr()] = (dev.numel() * dev.element_size(), host)
        return dev

    def conv(self, name, splits, pixel_shuffle=False):
        key = ("conv", name, tuple(splits), pixel_shuffle)
        if key not in self._cache:
            w = self.sd[name + ".weight"]
            b = self.sd.get(name + ".bias")
            wp, bp, cout, m_pad = layout_conv(w, b, splits, pixel_shuffle)
            self._cache[key] = (self._dev(wp), self._dev(bp), cout, m_pad, w.shape[2], w.shape[3])
        return self._cache[key]

    def conv_f16x3(self, name, splits, pixel_shuffle=False):
        key = ("conv16", name, tuple(splits), pixel_shuffle)
        if key not in self._cache:
            planes, unscale = layout_conv_f16x3(self.sd[name + ".weight"], splits, pixel_shuffle)
            self._cache[key] = (self._dev(planes), unscale)
        return self._cache[key]

    def ffn_f16x3(self, ffn_prefix, pre_name=None):
        """Device blobs for lssvc_ffn_f16x3: ConvFFN `ffn_prefix`.conv.{0,2} and, optionally, the leading 1x1 conv."""
        key = ("ffn16", ffn_prefix, pre_name)
        if key not in self._cache:
            w1, w2 = self.sd[ffn_prefix + ".conv.0.weight"], self.sd[ffn_prefix + ".conv.2.weight"]
            (a, ua), (b, ub) = layout_ffn_f16x3(w1, w2)
            rec = {"w1": self._dev(a), "u1": ua, "w2": self._dev(b), "u2": ub, "hidden": w1.shape[0], "C": w1.shape[1],
                   "b1": self._dev(self.sd[ffn_prefix + ".conv.0.bias"]), "b2": self._dev(self.sd[ffn_prefix + ".conv.2.bias"])}
            if pre_name is not None:
                wp = self.sd[pre_name + ".weight"]
                blob, up = layout_pw_natural_f16x3(wp)
                rec.update({"wp": self._dev(blob), "up": up, "bp": self._dev(self.sd[pre_name + ".bias"]), "pre_cin": wp.shape[1]})
            self._cache[key] = rec
        return self._cache[key]

    def conv_t(self, name, stride):
        key = ("convT", name, stride)
        if key not in self._cache:
            w, b, pad, ps = conv_t_as_conv(self.sd[name + ".weight"], self.sd[name + ".bias"], stride)
            wp, bp, cout, m_pad = layout_conv(w, b, [w.shape[1]], False)  # rows already in shuffle order
            self._cache[key] = (self._dev(wp), self._dev(bp), cout, m_pad, w.shape[2], w.shape[3], pad, ps)
        return self._cache[key]

    def dwconv(self, name):
        key = ("dw", name)
        if key not in self._cache:
            w = self.sd[name + ".weight"]                      # (C,1,3,3)
            self._cache[key] = (self._dev(w.reshape(w.shape[0], 9).t()), self._dev(self.sd[name + ".bias"]))
        return self._cache[key]

    def gdn(self, name, flavour):
        """flavour 'intra' (gdn.py + others.py reparam buffers) | 'inter' (video_net_component.py constants)."""
        key = ("gdn", name, flavour)
        if key not in self._cache:
            beta, gamma = self.sd[name + ".beta"], self.sd[name + ".gamma"]
            if flavour == "intra":
                beta = torch.max(beta, self.sd[name + ".beta_reparam.lower_bound.bound"]) ** 2 \
                    - self.sd[name + ".beta_reparam.pedestal"]
                gamma = torch.max(gamma, self.sd[name + ".gamma_reparam.lower_bound.bound"]) ** 2 \
                    - self.sd[name + ".gamma_reparam.pedestal"]
            else:
                beta = torch.max(beta, torch.ones_like(beta) * _BETA_BOUND) ** 2 - _PEDESTAL
                gamma = torch.max(gamma, torch.ones_like(gamma) * _REPARAM_OFFSET) ** 2 - _PEDESTAL
            c = gamma.shape[0]
            wp, bp, cout, m_pad = layout_conv(gamma.reshape(c, c, 1, 1), beta, [c], False)
            planes, unscale = layout_conv_f16x3(gamma.reshape(c, c, 1, 1), [c], False)
            self._cache[key] = (self._dev(wp), self._dev(bp), cout, m_pad, 1, 1)
            self._cache[("gdn16", name, flavour)] = (self._dev(planes), unscale)
        return self._cache[key]

    def gdn_f16x3(self, name, flavour):
        self.gdn(name, flavour)
        return self._cache[("gdn16", name, flavour)]

    def vector(self, key):
        k = ("vec", key)
        if k not in self._cache:
            self._cache[k] = self._dev(self.sd[key].reshape(-1))
        return self._cache[k]

    def bit_estimator(self, name):
        """[11][C]: softplus(h_i), b_i, tanh(a_i) for i=1..3, softplus(h_4), b_4 (video_entropy_models.py:110-129)."""
        key = ("be", name)
        if key not in self._cache:
            rows = []
            for i in (1, 2, 3):
                rows += [F.softplus(self.sd["%s.f%d.h" % (name, i)]), self.sd["%s.f%d.b" % (name, i)],
                         torch.tanh(self.sd["%s.f%d.a" % (name, i)])]
            rows += [F.softplus(self.sd[name + ".f4.h"]), self.sd[name + ".f4.b"]]
            self._cache[key] = self._dev(torch.stack([r.reshape(-1) for r in rows], 0))
        return self._cache[key]

    def entropy_bottleneck(self, name):
        """[59][C]: softplus(matrices) 3+9+9+9+3, biases 3+3+3+3+1, tanh(factors) 3x4, median
        (img_entropy_models.py:483-502, 432-434)."""
        key = ("eb", name)
        if key not in self._cache:
            rows = []
            for i in range(5):
                m = F.softplus(self.sd["%s._matrices.%d" % (name, i)])      # (C, f_out, f_in)
                rows += [m[:, j, k] for j in range(m.shape[1]) for k in range(m.shape[2])]
            for i in range(5):
                b = self.sd["%s._biases.%d" % (name, i)]
                rows += [b[:, j, 0] for j in range(b.shape[1])]
            for i in range(4):
                f = torch.tanh(self.sd["%s._factors.%d" % (name, i)])
                rows += [f[:, j, 0] for j in range(f.shape[1])]
            rows.append(self.sd[name + ".quantiles"][:, 0, 1])
            assert len(rows) == 59
            self._cache[key] = self._dev(torch.stack(rows, 0))
        return self._cache[key]
