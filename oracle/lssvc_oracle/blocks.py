"""ORACLE (test infrastructure only) -- CPU fp32 restatement of LSSVC's NN building blocks.

Nothing in the product package (`lssvc_amd/`) may import this module; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg use it, as the checker.

Everything here is a pure function of (input tensors, state-dict slice).  Tensors are
NCHW fp32 on the CPU, exactly as the reference's PyTorch CPU path computes them, so the
functions below are bit-comparable with the reference when fed the same state dict.
Each function cites the reference file:line (relative to /root/reference/) it restates.
"""
import torch
import torch.nn.functional as F


class Params:
    """A prefix view over a flat state dict: Params(sd, 'g_a.')['conv1.weight']."""

    def __init__(self, sd, prefix=""):
        self.sd = sd
        self.prefix = prefix

    def __getitem__(self, key):
        return self.sd[self.prefix + key]

    def has(self, key):
        return (self.prefix + key) in self.sd

    def sub(self, name):
        return Params(self.sd, self.prefix + name + ".")


# ----------------------------------------------------------------------------- conv helpers
def conv(x, p, name, stride=1, pad=None):
    """nn.Conv2d with square kernel, padding k//2 unless given (layers.py:36-38,55-57)."""
    w = p[name + ".weight"]
    b = p[name + ".bias"] if p.has(name + ".bias") else None
    if pad is None:
        pad = w.shape[-1] // 2
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def conv_t(x, p, name, stride, out_pad):
    """nn.ConvTranspose2d(k=3, padding=1) as used by the BL hyper/MV decoders (dmc_net.py:198-221)."""
    return F.conv_transpose2d(x, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=1,
                              output_padding=out_pad)


def subpel(x, p, name, r=2):
    """conv -> PixelShuffle(r): subpel_conv3x3 / subpel_conv1x1 (layers.py:41-52). `name` is the
    nn.Sequential; the conv is its element 0."""
    return F.pixel_shuffle(conv(x, p, name + ".0"), r)


def lrelu(x, slope=0.01):
    return F.leaky_relu(x, slope)


# ----------------------------------------------------------------------------- GDN (two flavours)
def gdn_intra(x, p, name, inverse=False):
    """IntraModules GDN (gdn.py:8-44) with NonNegativeParametrizer (others.py:43-67):
    beta = max(beta, bound)^2 - pedestal, same for gamma; norm = conv1x1(x^2, gamma, beta);
    out = x * sqrt(norm) (inverse) or x * rsqrt(norm)."""
    q = p.sub(name)
    c = x.shape[1]
    beta = torch.max(q["beta"], q["beta_reparam.lower_bound.bound"]) ** 2 - q["beta_reparam.pedestal"]
    gamma = torch.max(q["gamma"], q["gamma_reparam.lower_bound.bound"]) ** 2 - q["gamma_reparam.pedestal"]
    norm = F.conv2d(x ** 2, gamma.reshape(c, c, 1, 1), beta)
    norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
    return x * norm


_REPARAM_OFFSET = 2 ** -18
_PEDESTAL = _REPARAM_OFFSET ** 2
_BETA_BOUND = (1e-6 + _REPARAM_OFFSET ** 2) ** 0.5
_GAMMA_BOUND = _REPARAM_OFFSET


def gdn_inter(x, p, name, inverse=False):
    """InterModules GDN (video_net_component.py:52-105): bounds are python floats, the norm is
    sqrt(conv1x1(x^2, gamma, beta)) and the forward form DIVIDES (x / norm), inverse multiplies."""
    q = p.sub(name)
    c = x.shape[1]
    beta = torch.max(q["beta"], torch.ones_like(q["beta"]) * _BETA_BOUND) ** 2 - _PEDESTAL
    gamma = torch.max(q["gamma"], torch.ones_like(q["gamma"]) * _GAMMA_BOUND) ** 2 - _PEDESTAL
    norm = torch.sqrt(F.conv2d(x ** 2, gamma.view(c, c, 1, 1), beta))
    return x * norm if inverse else x / norm


# ----------------------------------------------------------------------------- residual blocks
def res_block(x, p, name, slope=0.01, start_from_relu=True, end_with_relu=False):
    """ResBlock: x + [lrelu](conv2(lrelu(conv1([lrelu](x))))) (layers.py:229-255,
    video_net_component.py:170-188). Bottleneck width comes from the weights' own shapes."""
    q = p.sub(name)
    out = lrelu(x, slope) if start_from_relu else x
    out = conv(out, q, "conv1")
    out = lrelu(out, slope)
    out = conv(out, q, "conv2")
    if end_with_relu:
        out = lrelu(out, slope)
    return x + out


def residual_block(x, p, name, slope=0.01):
    """IntraNoAR ResidualBlock: lrelu(conv2(lrelu(conv1 x))) + x (layers.py:122-145)."""
    q = p.sub(name)
    out = lrelu(conv(x, q, "conv1"), slope)
    out = lrelu(conv(out, q, "conv2"), slope)
    return out + x


def residual_block_with_stride(x, p, name):
    """conv3x3 s2 -> lrelu -> conv3x3 -> GDN, plus 1x1 s2 skip (layers.py:60-91)."""
    q = p.sub(name)
    out = lrelu(conv(x, q, "conv1", stride=2))
    out = conv(out, q, "conv2")
    out = gdn_intra(out, q, "gdn")
    return out + conv(x, q, "downsample", stride=2, pad=0)


def residual_block_upsample(x, p, name):
    """subpel conv3x3 -> lrelu -> conv3x3 -> IGDN, plus subpel conv3x3 skip (layers.py:94-119)."""
    q = p.sub(name)
    out = lrelu(subpel(x, q, "subpel_conv"))
    out = conv(out, q, "conv")
    out = gdn_intra(out, q, "igdn", inverse=True)
    return out + subpel(x, q, "upsample")


def depth_conv_block(x, p, name):
    """DepthConvBlock = DepthConv + ConvFFN (lssvc_modules.py:15-72).
    DepthConv: 1x1 -> lrelu(.01) -> depthwise 3x3 -> 1x1, + (1x1 adaptor if Cin != Cout else x).
    ConvFFN:   x + lrelu(.1)(1x1(lrelu(.1)(1x1 x)))."""
    q = p.sub(name + ".block.0")
    ident = conv(x, q, "adaptor") if q.has("adaptor.weight") else x
    out = lrelu(conv(x, q, "conv1.0"), 0.01)
    wd = q["depth_conv.weight"]
    out = F.conv2d(out, wd, q["depth_conv.bias"], padding=1, groups=wd.shape[0])
    out = conv(out, q, "conv2") + ident
    f = p.sub(name + ".block.1")
    ffn = lrelu(conv(out, f, "conv.0"), 0.1)
    ffn = lrelu(conv(ffn, f, "conv.2"), 0.1)
    return out + ffn


# ----------------------------------------------------------------------------- resampling / warping
def bilinear(x, size):
    """F.interpolate(bilinear, align_corners=False) to an explicit size (layers.py:269,284;
    lssvc_modules.py:360,393,425)."""
    return F.interpolate(x, size=tuple(int(s) for s in size), mode="bilinear", align_corners=False)


def up2(x):
    """bilinearupsacling (video_net_component.py:355-360)."""
    return bilinear(x, (x.shape[2] * 2, x.shape[3] * 2))


def down2(x):
    """bilineardownsacling (video_net_component.py:363-368)."""
    return bilinear(x, (x.shape[2] // 2, x.shape[3] // 2))


def flow_warp(feature, flow):
    """torch_warp (video_net_component.py:329-347): grid = linspace(-1,1) + flow/((size-1)/2),
    grid_sample(bilinear, border, align_corners=True)."""
    n, _, h, w = flow.shape
    hor = torch.linspace(-1.0, 1.0, w, dtype=feature.dtype).view(1, 1, 1, w).expand(n, -1, h, -1)
    ver = torch.linspace(-1.0, 1.0, h, dtype=feature.dtype).view(1, 1, h, 1).expand(n, -1, -1, w)
    base = torch.cat([hor, ver], 1)
    nflow = torch.cat([flow[:, 0:1] / ((feature.size(3) - 1.0) / 2.0),
                       flow[:, 1:2] / ((feature.size(2) - 1.0) / 2.0)], 1)
    grid = base + nflow
    return F.grid_sample(feature, grid.permute(0, 2, 3, 1), mode="bilinear", padding_mode="border",
                         align_corners=True)


def spynet(im1, im2, p, name):
    """ME_Spynet / ME_Spynet_DCVC forward (video_net_component.py:213-248,292-326): 4-level
    avg-pool pyramid, per level flow = up2(flow)*2 + MEBasic(cat(im1, warp(im2, up), up))."""
    q = p.sub(name)
    levels = 4
    l1, l2 = [im1], [im2]
    for i in range(levels - 1):
        l1.append(F.avg_pool2d(l1[i], kernel_size=2, stride=2))
        l2.append(F.avg_pool2d(l2[i], kernel_size=2, stride=2))
    coarse = l2[levels - 1]
    flow = torch.zeros(im1.shape[0], 2, coarse.shape[2] // 2, coarse.shape[3] // 2, dtype=torch.float32)
    for lvl in range(levels):
        up = up2(flow) * 2.0
        a = l1[levels - 1 - lvl]
        b = flow_warp(l2[levels - 1 - lvl], up)
        t = torch.cat([a, b, up], 1)
        m = q.sub("moduleBasic.%d" % lvl)
        t = F.relu(conv(t, m, "conv1"))
        t = F.relu(conv(t, m, "conv2"))
        t = F.relu(conv(t, m, "conv3"))
        t = F.relu(conv(t, m, "conv4"))
        t = conv(t, m, "conv5")
        flow = up + t
    return flow
